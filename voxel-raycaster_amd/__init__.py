"""voxel-raycaster_amd -- MI355X-native drop-in for the reference's CLCaster path.

Host-side mirror (Python, for tests and bench) of the reference interface
``class CLCaster`` (include/CLCaster.h:93-329): same method names, argument
meaning and bool-returning error behaviour, layered over the C ABI of
``libvrc.so`` (include/vrc.h).  All compute happens in the hand-written gfx950
kernels inside that library; there is no Python or CPU fallback -- importing
this package without the built library raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VRC_LIB_PATH: load another build of the same library (A/B variants and the -DVRC_SCHED_STATS profiling build of tools/)
# instead of copying it over the product file
LIB_PATH = os.environ.get("VRC_LIB_PATH") or os.path.join(_HERE, "libvrc.so")


class VrcError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise VrcError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i8p = C.POINTER(C.c_int8)
_H = C.c_void_p


class BuildInfo(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in ("n_descriptors", "root_index", "n_bricks", "n_top_slots", "n_far_pointers_top")]
                + [(n, C.c_double) for n in ("seconds_total", "seconds_height", "seconds_count", "seconds_emit")]
                + [(n, C.c_uint64) for n in ("device_bytes_peak", "host_bytes", "validate_samples", "validate_mismatches")])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Memory(C.Structure):
    _fields_ = [("device", C.c_int32), ("rows", C.c_int32), ("viewport_bytes", C.c_uint64), ("image_bytes", C.c_uint64),
                ("hit_bytes", C.c_uint64), ("octree_bytes", C.c_uint64), ("octree_shared", C.c_int32), ("peer_access", C.c_int32),
                ("coarse_bytes", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Memory2(C.Structure):
    _fields_ = ([("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in ("device", "rows", "octree_shared", "peer_access", "tree_holders",
                                                                      "coarse_log2", "empty_boxes")]
                + [(n, C.c_uint64) for n in ("viewport_bytes", "image_bytes", "hit_bytes", "octree_bytes", "coarse_bytes", "box_bytes")]
                + [("box_build_seconds", C.c_double), ("note", C.c_char * 160), ("box_queries_cut", C.c_uint64),
                   ("box_records", C.c_uint64), ("box_levels", C.c_int32), ("reserved_", C.c_int32)])

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["note"] = d["note"].decode(errors="replace")
        return d


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "primary_rays", "shadow_rays", "descriptor_reads", "texel_reads", "map_reads", "steps",
        "unwritten_pixels", "watchdog_trips")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "watchdog_trips"}


# every symbol include/vrc.h declares, with its signature
SIGNATURES = {
    "vrc_create": (C.c_int, [C.c_int, C.POINTER(_H)]),
    "vrc_destroy": (C.c_int, [_H]),
    "vrc_last_error": (C.c_char_p, [_H]),
    "vrc_device_count": (C.c_int, [_i32p]),
    "vrc_assign_map": (C.c_int, [_H, _i8p, C.c_int32, C.c_int32, C.c_int32]),
    "vrc_release_map": (C.c_int, [_H]),
    "vrc_assign_octree": (C.c_int, [_H, _u64p, C.c_uint64, C.c_uint64]),
    "vrc_assign_octree_attachments": (C.c_int, [_H, C.POINTER(C.c_uint32), C.c_uint64, _u64p, C.c_uint64]),
    "vrc_release_octree": (C.c_int, [_H]),
    "vrc_create_viewport": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_float, C.c_float]),
    "vrc_create_viewport_table": (C.c_int, [_H, C.c_int32, C.c_int32, _f32p]),
    "vrc_release_viewport": (C.c_int, [_H]),
    "vrc_create_texture_atlas": (C.c_int, [_H, _u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "vrc_assign_camera": (C.c_int, [_H, _f32p, _f32p]),
    "vrc_assign_camera_trig": (C.c_int, [_H, _f32p]),
    "vrc_release_camera": (C.c_int, [_H]),
    "vrc_assign_lights": (C.c_int, [_H, _f32p, _i32p]),
    "vrc_setting_add": (C.c_int, [_H, C.c_char_p, C.c_char_p, C.c_int64]),
    "vrc_setting_set": (C.c_int, [_H, C.c_char_p, C.c_int64]),
    "vrc_setting_get": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_int64)]),
    "vrc_validate": (C.c_int, [_H]),
    "vrc_prepare": (C.c_int, [_H]),
    "vrc_compute": (C.c_int, [_H]),
    "vrc_compute_async": (C.c_int, [_H]),
    "vrc_sync": (C.c_int, [_H]),
    "vrc_set_row_tiling": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32]),
    "vrc_read_image_f32": (C.c_int, [_H, _f32p, C.c_size_t]),
    "vrc_read_image_rgba8": (C.c_int, [_H, _u8p, C.c_size_t]),
    "vrc_read_hits": (C.c_int, [_H, _i32p, C.c_size_t]),
    "vrc_device_image": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "vrc_get_counters": (C.c_int, [_H, C.POINTER(Counters)]),
    "vrc_counters_canonical": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "vrc_get_scheduler_stats": (C.c_int, [_H, _u64p]),
    "vrc_timing_reset": (C.c_int, [_H]),
    "vrc_timing_get": (C.c_int, [_H, _u64p, C.POINTER(C.c_double)]),
    "vrc_octree_generate": (C.c_int, [_i8p, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(_u64p), _u64p, _u64p]),
    "vrc_scene_shell_terrain": (C.c_int, [C.c_uint32, C.c_uint64, C.c_int32, C.c_int, C.POINTER(_u64p), _u64p, _u64p, _i32p]),
    "vrc_octree_attachments_from_grid": (C.c_int, [_i8p, C.c_uint32, _u64p, C.c_uint64, C.c_uint64,
                                                    C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(_u64p), _u64p]),
    "vrc_scene_shell_terrain_attachments": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint32, _u64p, C.c_uint64, C.c_uint64,
                                                       C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(_u64p), _u64p]),
    "vrc_scene_shell_terrain_dense": (C.c_int, [C.c_uint32, C.c_uint64, C.c_int32, _i8p]),
    "vrc_scene_atlas": (C.c_int, [C.c_int32, C.c_int32, _u8p]),
    "vrc_octree_get_voxel": (C.c_int, [_u64p, C.c_uint64, C.c_uint32, _i32p, _i32p, _i32p, _i32p]),
    "vrc_octree_save": (C.c_int, [C.c_char_p, C.c_uint32, _u64p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), _u64p, C.c_uint64]),
    "vrc_octree_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(_u64p), _u64p, _u64p,
                                  C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(_u64p), _u64p]),
    "vrc_assign_octree_file": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_uint32)]),
    "vrc_scene_diamond_square": (C.c_int, [C.c_uint32, C.c_double, _u8p, _i8p]),
    "vrc_scene_diamond_square_f64": (C.c_int, [C.c_uint32, C.c_double, C.POINTER(C.c_double)]),
    "vrc_build_heightfield": (C.c_int, [_H, C.c_uint32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.c_uint32, C.c_uint64,
                                       C.POINTER(BuildInfo)]),
    "vrc_octree_from_columns": (C.c_int, [C.c_uint32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.c_uint32, C.POINTER(_u64p), _u64p, _u64p]),
    "vrc_build_dense_grid": (C.c_int, [_H, C.c_uint32, _i8p, C.c_uint32, C.c_uint64, C.POINTER(BuildInfo)]),
    "vrc_octree_generate_ex": (C.c_int, [_i8p, C.c_uint32, C.c_uint32, C.POINTER(_u64p), _u64p, _u64p]),
    "vrc_free": (None, [C.c_void_p]),
    "vrc_set_row_slice": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32]),
    "vrc_create_group": (C.c_int, [_i32p, C.c_int32, C.c_int32, C.POINTER(_H)]),
    "vrc_create_group_ex": (C.c_int, [_i32p, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(_H)]),
    "vrc_group_size": (C.c_int, [_H, _i32p]),
    "vrc_pin_host_buffer": (C.c_int, [C.c_void_p, C.c_size_t]),
    "vrc_unpin_host_buffer": (C.c_int, [C.c_void_p]),
    "vrc_memory_usage": (C.c_int, [_H, C.c_int32, C.POINTER(Memory)]),
    "vrc_empty_boxes_check": (C.c_int, [_H, C.c_uint64, C.c_uint64, _u64p, _u64p, C.POINTER(C.c_double)]),
    "vrc_assign_octree_from": (C.c_int, [_H, _H]),
    "vrc_read_empty_boxes": (C.c_int, [_H, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]),
    "vrc_memory_usage2": (C.c_int, [_H, C.c_int32, C.POINTER(Memory2)]),
    "vrc_scene_shell_terrain_ex": (C.c_int, [C.c_uint32, C.c_uint64, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(_u64p), _u64p, _u64p, _i32p]),
    "vrc_scene_shell_column": (C.c_int, [C.c_uint32, C.c_uint64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _i32p, _i32p]),
    "vrc_build_shell_terrain": (C.c_int, [_H, C.c_uint32, C.c_uint64, C.c_int32, C.c_int32, C.c_uint32, C.c_uint64, _i32p, C.c_uint32,
                                          _i32p, C.POINTER(BuildInfo)]),
    "vrc_read_descriptors": (C.c_int, [_H, C.c_uint64, C.c_uint64, _u64p]),
    "vrc_octree_size": (C.c_int, [_H, _u64p, _u64p]),
}
LAYOUT_STRICT_REFERENCE, LAYOUT_NO_PAGE_HEADERS, BUILD_COUNT_ONLY = 1, 2, 1
BUILD_ATTACHMENTS = 2
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

STATUS = {0: "VRC_OK", 1: "VRC_ERR_INVALID_ARGUMENT", 2: "VRC_ERR_NOT_READY", 3: "VRC_ERR_DEVICE",
          4: "VRC_ERR_OUT_OF_MEMORY", 5: "VRC_ERR_NOT_FOUND", 6: "VRC_ERR_LIMIT"}


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(ty)


# ---------------------------------------------------------------------------
# scene-side helpers: Octree / Map (src/map/Octree.cpp, src/map/Map.cpp)
# ---------------------------------------------------------------------------
class Octree:
    """Descriptor array + root index (include/map/Octree.h:26-124)."""

    buffer_size = 100000  # Octree.h:29

    def __init__(self, descriptors: np.ndarray, root_index: int, dim: int):
        self.descriptor_buffer = np.ascontiguousarray(descriptors, dtype=np.uint64)
        self.root_index = int(root_index)
        self.dim = int(dim)
        self.attachment_lookup = None      # uint32[n_descriptors]  (Octree.h:46)
        self.attachment_buffer = None      # uint64[]               (Octree.h:49)

    @classmethod
    def Generate(cls, data: np.ndarray, dim: int, buffer_size: int = 0, strict_reference: bool = True,
                 layout: Optional[int] = None) -> "Octree":
        """Octree::Generate (src/map/Octree.cpp:13-43).  data: int8[dim^3], x + dim*(y + dim*z).  layout (VRC_LAYOUT_* flags,
        e.g. 2 = no page headers, what the device builders emit): vrc_octree_generate_ex, the array sized exactly."""
        data = np.ascontiguousarray(data, dtype=np.int8).reshape(-1)
        if data.size != dim ** 3:
            raise VrcError("grid size does not match dim^3")
        out = _u64p()
        n = C.c_uint64()
        root = C.c_uint64()
        if layout is not None:
            rc = lib.vrc_octree_generate_ex(_ptr(data, _i8p), dim, layout, C.byref(out), C.byref(n), C.byref(root))
        else:
            rc = lib.vrc_octree_generate(_ptr(data, _i8p), dim, buffer_size, int(strict_reference),
                                         C.byref(out), C.byref(n), C.byref(root))
        if rc != 0:
            raise VrcError(f"vrc_octree_generate: {STATUS.get(rc, rc)}")
        arr = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
        lib.vrc_free(out)
        return cls(arr, root.value, dim)

    def _take_attachments(self, lookup_p, attach_p, n_attach):
        self.attachment_lookup = np.ctypeslib.as_array(lookup_p, shape=(self.descriptor_buffer.size,)).copy()
        self.attachment_buffer = np.ctypeslib.as_array(attach_p, shape=(n_attach,)).copy()
        lib.vrc_free(lookup_p)
        lib.vrc_free(attach_p)

    def attach_materials_from_grid(self, data: np.ndarray) -> "Octree":
        """Fill attachment_lookup / attachment_buffer (Octree.h:46-50) with the per-voxel materials of `data`."""
        data = np.ascontiguousarray(data, dtype=np.int8).reshape(-1)
        lp, ap, n = C.POINTER(C.c_uint32)(), _u64p(), C.c_uint64()
        rc = lib.vrc_octree_attachments_from_grid(_ptr(data, _i8p), self.dim, _ptr(self.descriptor_buffer, _u64p),
                                                  self.descriptor_buffer.size, self.root_index, C.byref(lp), C.byref(ap), C.byref(n))
        if rc != 0:
            raise VrcError(f"vrc_octree_attachments_from_grid: {STATUS.get(rc, rc)}")
        self._take_attachments(lp, ap, n.value)
        return self

    def attach_materials_procedural(self, depth: int, seed: int = 1, mirror_period: int = 64) -> "Octree":
        lp, ap, n = C.POINTER(C.c_uint32)(), _u64p(), C.c_uint64()
        rc = lib.vrc_scene_shell_terrain_attachments(depth, seed, mirror_period, _ptr(self.descriptor_buffer, _u64p),
                                                     self.descriptor_buffer.size, self.root_index, C.byref(lp), C.byref(ap), C.byref(n))
        if rc != 0:
            raise VrcError(f"vrc_scene_shell_terrain_attachments: {STATUS.get(rc, rc)}")
        self._take_attachments(lp, ap, n.value)
        return self

    def Save(self, path: str) -> None:
        has = self.attachment_lookup is not None and self.attachment_buffer is not None
        rc = lib.vrc_octree_save(path.encode(), self.dim, _ptr(self.descriptor_buffer, _u64p), self.descriptor_buffer.size,
                                 self.root_index, _ptr(self.attachment_lookup, C.POINTER(C.c_uint32)) if has else None,
                                 _ptr(self.attachment_buffer, _u64p) if has else None,
                                 self.attachment_buffer.size if has else 0)
        if rc != 0:
            raise VrcError(f"vrc_octree_save: {STATUS.get(rc, rc)}")

    @classmethod
    def Load(cls, octree_file_name: str) -> "Octree":
        """Octree::Load (include/map/Octree.h:38)."""
        dim, n, root, na = C.c_uint32(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        dp, lp, ap = _u64p(), C.POINTER(C.c_uint32)(), _u64p()
        rc = lib.vrc_octree_load(octree_file_name.encode(), C.byref(dim), C.byref(dp), C.byref(n), C.byref(root),
                                 C.byref(lp), C.byref(ap), C.byref(na))
        if rc != 0:
            raise VrcError(f"vrc_octree_load: {STATUS.get(rc, rc)}")
        o = cls(np.ctypeslib.as_array(dp, shape=(n.value,)).copy(), root.value, dim.value)
        lib.vrc_free(dp)
        if na.value:
            o._take_attachments(lp, ap, na.value)
        return o

    def GetVoxel(self, position):
        """Octree::GetVoxel (src/map/Octree.cpp:45-158): (found, resolution, sub_oct_pos)."""
        pos = (C.c_int32 * 3)(*[int(v) for v in position])
        found = C.c_int32()
        res = C.c_int32()
        sub = (C.c_int32 * 3)()
        rc = lib.vrc_octree_get_voxel(_ptr(self.descriptor_buffer, _u64p), self.root_index, self.dim, pos,
                                      C.byref(found), C.byref(res), sub)
        if rc != 0:
            raise VrcError(f"vrc_octree_get_voxel: {STATUS.get(rc, rc)}")
        return bool(found.value), int(res.value), tuple(int(v) for v in sub)


def shell_terrain(depth: int, seed: int = 1, thickness: int = 2, strict_reference: bool = False,
                  want_height: bool = True):
    """Procedural sparse scene of SURVEY 8(d); returns (Octree, height[dim,dim] or None)."""
    dim = 1 << depth
    out = _u64p()
    n = C.c_uint64()
    root = C.c_uint64()
    height = np.zeros((dim, dim), dtype=np.int32) if want_height else None
    rc = lib.vrc_scene_shell_terrain(depth, seed, thickness, int(strict_reference), C.byref(out), C.byref(n),
                                     C.byref(root), _ptr(height, _i32p) if want_height else None)
    if rc != 0:
        raise VrcError(f"vrc_scene_shell_terrain: {STATUS.get(rc, rc)}")
    arr = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    lib.vrc_free(out)
    return Octree(arr, root.value, dim), height


def shell_terrain_ex(depth: int, seed: int = 1, thickness: int = 2, octave_floor: int = 2, layout: int = 0,
                     want_height: bool = False):
    """vrc_scene_shell_terrain_ex: the scene with its knobs exposed (layout = LAYOUT_* flags)."""
    dim = 1 << depth
    out, n, root = _u64p(), C.c_uint64(), C.c_uint64()
    height = np.zeros((dim, dim), dtype=np.int32) if want_height else None
    rc = lib.vrc_scene_shell_terrain_ex(depth, seed, thickness, octave_floor, layout, C.byref(out), C.byref(n), C.byref(root),
                                        _ptr(height, _i32p) if want_height else None)
    if rc != 0:
        raise VrcError(f"vrc_scene_shell_terrain_ex: {STATUS.get(rc, rc)}")
    arr = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    lib.vrc_free(out)
    return Octree(arr, root.value, dim), height


def shell_column(depth: int, x: int, y: int, seed: int = 1, thickness: int = 2, octave_floor: int = 2):
    """(lo, hi) of one column of the scene, evaluated procedurally: solid for lo <= z <= hi."""
    lo, hi = C.c_int32(), C.c_int32()
    rc = lib.vrc_scene_shell_column(depth, seed, thickness, octave_floor, int(x), int(y), C.byref(lo), C.byref(hi))
    if rc != 0:
        raise VrcError(f"vrc_scene_shell_column: {STATUS.get(rc, rc)}")
    return int(lo.value), int(hi.value)


def shell_terrain_dense(depth: int, seed: int = 1, thickness: int = 2) -> np.ndarray:
    dim = 1 << depth
    grid = np.zeros(dim ** 3, dtype=np.int8)
    rc = lib.vrc_scene_shell_terrain_dense(depth, seed, thickness, _ptr(grid, _i8p))
    if rc != 0:
        raise VrcError(f"vrc_scene_shell_terrain_dense: {STATUS.get(rc, rc)}")
    return grid


def diamond_square(dim: int, corner_seed: float = 58.0, want_grid: bool = True):
    """Map::GenerateHeightBitmap (src/map/Map.cpp:144-262) + the voxel fill ApplyHeightmap never got.
    Returns (height uint8[dim, dim], grid int8[dim^3] or None)."""
    height = np.zeros((dim, dim), dtype=np.uint8)
    grid = np.zeros(dim ** 3, dtype=np.int8) if want_grid else None
    rc = lib.vrc_scene_diamond_square(dim, corner_seed, _ptr(height, _u8p), _ptr(grid, _i8p) if want_grid else None)
    if rc != 0:
        raise VrcError(f"vrc_scene_diamond_square: {STATUS.get(rc, rc)}")
    return height, grid


def diamond_square_field(dim: int, corner_seed: float = 58.0) -> np.ndarray:
    """The double field Map::GenerateHeightBitmap holds before Map.cpp:248 quantises it, float64[dim, dim] indexed [y, x]."""
    field = np.zeros((dim, dim), dtype=np.float64)
    rc = lib.vrc_scene_diamond_square_f64(dim, corner_seed, field.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise VrcError(f"vrc_scene_diamond_square_f64: {STATUS.get(rc, rc)}")
    return field


def octree_from_columns(depth: int, hi: np.ndarray, lo: Optional[np.ndarray] = None, layout: int = 2) -> "Octree":
    """vrc_octree_from_columns: the host twin of CLCaster.build_heightfield (layout 2 = VRC_LAYOUT_NO_PAGE_HEADERS, the
    layout the device builder emits)."""
    dim = 1 << depth
    hi = np.ascontiguousarray(hi, dtype=np.uint16).reshape(dim, dim)
    lo = None if lo is None else np.ascontiguousarray(lo, dtype=np.uint16).reshape(dim, dim)
    out, n, root = _u64p(), C.c_uint64(), C.c_uint64()
    u16 = C.POINTER(C.c_uint16)
    rc = lib.vrc_octree_from_columns(depth, _ptr(hi, u16), _ptr(lo, u16) if lo is not None else None, layout,
                                     C.byref(out), C.byref(n), C.byref(root))
    if rc != 0:
        raise VrcError(f"vrc_octree_from_columns: {STATUS.get(rc, rc)}")
    arr = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    lib.vrc_free(out)
    return Octree(arr, root.value, dim)


def pin_host_buffer(a: np.ndarray) -> None:
    """vrc_pin_host_buffer: page-lock a caller-owned frame buffer so that read_image*(out=a) copies by DMA without a
    staging pass (the reference shares a GL texture instead, src/CLCaster.cpp:278-296)."""
    rc = lib.vrc_pin_host_buffer(C.c_void_p(a.ctypes.data), a.nbytes)
    if rc != 0:
        raise VrcError(f"vrc_pin_host_buffer: {STATUS.get(rc, rc)}")


def unpin_host_buffer(a: np.ndarray) -> None:
    rc = lib.vrc_unpin_host_buffer(C.c_void_p(a.ctypes.data))
    if rc != 0:
        raise VrcError(f"vrc_unpin_host_buffer: {STATUS.get(rc, rc)}")


def synthetic_atlas(width: int = 256, height: int = 256) -> np.ndarray:
    a = np.zeros((height, width, 4), dtype=np.uint8)
    rc = lib.vrc_scene_atlas(width, height, _ptr(a, _u8p))
    if rc != 0:
        raise VrcError(f"vrc_scene_atlas: {STATUS.get(rc, rc)}")
    return a


class Map:
    """Map (src/map/Map.cpp:5-19): dense ArrayMap + the Octree generated from it."""

    def __init__(self, dimensions: int, data: Optional[np.ndarray] = None, buffer_size: int = 0,
                 strict_reference: bool = True):
        self.dimensions = int(dimensions)
        if data is None:
            data = np.full(self.dimensions ** 3, 5, dtype=np.int8)  # ArrayMap ctor fills with 5 (ArrayMap.cpp:17-23)
        self.array_map = np.ascontiguousarray(data, dtype=np.int8).reshape(-1)
        self.octree = Octree.Generate(self.array_map, self.dimensions, buffer_size, strict_reference)


# ---------------------------------------------------------------------------
# CLCaster (include/CLCaster.h:93-329)
# ---------------------------------------------------------------------------
class CLCaster:
    """Same surface as the reference class; every method returns bool like there."""

    def __init__(self):
        self._h = _H()
        self._keep = {}
        self.last_status = 0

    # -- helpers
    def _ok(self, rc: int) -> bool:
        self.last_status = rc
        return rc == 0

    def last_error(self) -> str:
        if not self._h:
            return "not initialised"
        return lib.vrc_last_error(self._h).decode()

    def __del__(self):
        try:
            if self._h:
                lib.vrc_destroy(self._h)
                self._h = _H()
        except Exception:
            pass

    # -- CLCaster::init (CLCaster.cpp:14-74)
    def init(self, device_ordinal: int = 0) -> bool:
        return self._ok(lib.vrc_create(device_ordinal, C.byref(self._h)))

    def init_group(self, device_ordinals, band_rows: int = 8, own_copies: Optional[bool] = None) -> bool:
        """One host thread, several GPUs (vrc_create_group): this object becomes rank 0 of a row-sliced group.
        own_copies (VRC_GROUP_OWN_COPIES): ranks on rank 0's GPU take the copy path of a rank on another GPU; None (the
        default) leaves the choice to vrc_create_group, which honours VRC_GROUP_OWN_COPIES=1 in the environment like a C host."""
        d = np.ascontiguousarray(device_ordinals, dtype=np.int32)
        if own_copies is None:
            return self._ok(lib.vrc_create_group(_ptr(d, _i32p), d.size, band_rows, C.byref(self._h)))
        return self._ok(lib.vrc_create_group_ex(_ptr(d, _i32p), d.size, band_rows, 1 if own_copies else 0, C.byref(self._h)))

    def group_size(self) -> int:
        n = C.c_int32()
        lib.vrc_group_size(self._h, C.byref(n))
        return int(n.value)

    def memory_usage(self, rank: int = 0) -> dict:
        m = Memory()
        if not self._ok(lib.vrc_memory_usage(self._h, rank, C.byref(m))):
            raise VrcError(self.last_error())
        return m.as_dict()

    def read_empty_boxes(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        """vrc_read_empty_boxes: uint32[count, 8] box words of the descriptors first .. first + count (all of them by default)."""
        if count is None:
            n = C.c_uint64()
            if not self._ok(lib.vrc_octree_size(self._h, C.byref(n), None)):
                raise VrcError(self.last_error())
            count = int(n.value) - first
        out = np.zeros((count, 8), dtype=np.uint32)
        if not self._ok(lib.vrc_read_empty_boxes(self._h, first, count, _ptr(out, C.POINTER(C.c_uint32)))):
            raise VrcError(self.last_error())
        return out

    def memory_usage2(self, rank: int = 0) -> dict:
        """vrc_memory_usage2: the size-versioned form, with what is held per TREE (coarse table, empty boxes, holders)."""
        m = Memory2()
        m.struct_size = C.sizeof(Memory2)
        if not self._ok(lib.vrc_memory_usage2(self._h, rank, C.byref(m))):
            raise VrcError(self.last_error())
        return m.as_dict()

    def used_empty_boxes(self) -> bool:
        """True when the last frame was rendered with the tree's empty boxes: the descriptor-read counts (counter and hit-record
        field 7) are then those of the box traversal, not SURVEY 8d's canonical ones (setting empty_boxes = 0 gives those)."""
        return bool(self.memory_usage2()["empty_boxes"])

    def assign_octree_from(self, src: "CLCaster") -> bool:
        """vrc_assign_octree_from: adopt the tree (arrays, coarse table, empty boxes) another caster on the same GPU holds."""
        return self._ok(lib.vrc_assign_octree_from(self._h, src._h))

    def empty_boxes_check(self, samples: int = 1 << 20, seed: int = 1) -> dict:
        """vrc_empty_boxes_check: sampled voxels of the empty boxes the last frame used, looked up in the tree."""
        n, bad, sec = C.c_uint64(), C.c_uint64(), C.c_double()
        if not self._ok(lib.vrc_empty_boxes_check(self._h, samples, seed, C.byref(n), C.byref(bad), C.byref(sec))):
            raise VrcError(self.last_error())
        return {"boxes_sampled": int(n.value), "solid_voxels": int(bad.value), "build_seconds": float(sec.value)}

    def build_shell_terrain(self, depth: int, seed: int = 1, thickness: int = 2, octave_floor: int = 2, count_only: bool = False,
                            validate_samples: int = 0, probe_xy: Optional[np.ndarray] = None):
        """vrc_build_shell_terrain: the scene's SVO built in HBM.  Returns (info dict, probe lo/hi array or None);
        raises on failure."""
        info = BuildInfo()
        pxy = plh = None
        if probe_xy is not None:
            pxy = np.ascontiguousarray(probe_xy, dtype=np.int32).reshape(-1, 2)
            plh = np.zeros_like(pxy)
        rc = lib.vrc_build_shell_terrain(self._h, depth, seed, thickness, octave_floor, BUILD_COUNT_ONLY if count_only else 0,
                                         validate_samples, _ptr(pxy, _i32p) if pxy is not None else None,
                                         0 if pxy is None else pxy.shape[0], _ptr(plh, _i32p) if plh is not None else None,
                                         C.byref(info))
        if not self._ok(rc):
            raise VrcError(self.last_error())
        return info.as_dict(), plh

    def build_heightfield(self, depth: int, hi: np.ndarray, lo: Optional[np.ndarray] = None, count_only: bool = False,
                          validate_samples: int = 0) -> dict:
        """vrc_build_heightfield: the SVO of a column scene (solid for lo <= z <= hi, uint16[dim, dim] each; lo None:
        from z = 0 up) built in HBM from the 2-D field alone.  Returns the build info; raises on failure."""
        dim = 1 << depth
        hi = np.ascontiguousarray(hi, dtype=np.uint16).reshape(dim, dim)
        lo = None if lo is None else np.ascontiguousarray(lo, dtype=np.uint16).reshape(dim, dim)
        info = BuildInfo()
        u16 = C.POINTER(C.c_uint16)
        rc = lib.vrc_build_heightfield(self._h, depth, _ptr(hi, u16), _ptr(lo, u16) if lo is not None else None,
                                       BUILD_COUNT_ONLY if count_only else 0, validate_samples, C.byref(info))
        if not self._ok(rc):
            raise VrcError(self.last_error())
        return info.as_dict()

    def build_dense_grid(self, depth: int, grid: Optional[np.ndarray] = None, count_only: bool = False, validate_samples: int = 0,
                         attachments: bool = False) -> dict:
        """vrc_build_dense_grid: Octree::Generate's input (int8[dim^3], x + dim*(y + dim*z), non-zero = solid) built into the
        SVO in HBM and installed as the octree; grid None: from the map assign_map has already uploaded.  Returns the build
        info; raises on failure."""
        dim = 1 << depth
        if grid is not None:
            grid = np.ascontiguousarray(grid, dtype=np.int8).reshape(-1)
            if grid.size != dim ** 3:
                raise VrcError("grid size does not match dim^3")
        info = BuildInfo()
        rc = lib.vrc_build_dense_grid(self._h, depth, _ptr(grid, _i8p) if grid is not None else None,
                                      (BUILD_COUNT_ONLY if count_only else 0) | (BUILD_ATTACHMENTS if attachments else 0),
                                      validate_samples, C.byref(info))
        if not self._ok(rc):
            raise VrcError(self.last_error())
        return info.as_dict()

    def octree_size(self):
        n, root = C.c_uint64(), C.c_uint64()
        if not self._ok(lib.vrc_octree_size(self._h, C.byref(n), C.byref(root))):
            raise VrcError(self.last_error())
        return int(n.value), int(root.value)

    def read_descriptors(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        if count is None:
            count = self.octree_size()[0] - first
        out = np.empty(count, dtype=np.uint64)
        if not self._ok(lib.vrc_read_descriptors(self._h, first, count, _ptr(out, _u64p))):
            raise VrcError(self.last_error())
        return out

    # -- scene
    def assign_map(self, map_: "Map | np.ndarray", dims=None) -> bool:
        if isinstance(map_, Map):
            data, d = map_.array_map, (map_.dimensions,) * 3
        else:
            data, d = np.ascontiguousarray(map_, dtype=np.int8).reshape(-1), tuple(dims)
        return self._ok(lib.vrc_assign_map(self._h, _ptr(data, _i8p), d[0], d[1], d[2]))

    def release_map(self) -> bool:
        return self._ok(lib.vrc_release_map(self._h))

    def assign_octree(self, map_: "Map | Octree") -> bool:
        oct_ = map_.octree if isinstance(map_, Map) else map_
        buf = oct_.descriptor_buffer
        if not self._ok(lib.vrc_assign_octree(self._h, _ptr(buf, _u64p), buf.size, oct_.root_index)):
            return False
        if oct_.attachment_lookup is not None and oct_.attachment_buffer is not None:
            return self._ok(lib.vrc_assign_octree_attachments(
                self._h, _ptr(oct_.attachment_lookup, C.POINTER(C.c_uint32)), oct_.attachment_lookup.size,
                _ptr(oct_.attachment_buffer, _u64p), oct_.attachment_buffer.size))
        return True

    def assign_octree_attachments(self, octree: "Octree") -> bool:
        """The two attachment arrays of `octree` (material per leaf voxel) for the tree that is already resident -- e.g. one
        built on the device (build_dense_grid / build_heightfield) and read back once to compute them."""
        return self._ok(lib.vrc_assign_octree_attachments(
            self._h, _ptr(octree.attachment_lookup, C.POINTER(C.c_uint32)), octree.attachment_lookup.size,
            _ptr(octree.attachment_buffer, _u64p), octree.attachment_buffer.size))

    def assign_octree_file(self, path: str) -> int:
        """Extension: stream a tree saved by Octree.Save straight into device memory; returns the map
        dimension stored in the file, 0 on failure."""
        dim = C.c_uint32()
        return int(dim.value) if self._ok(lib.vrc_assign_octree_file(self._h, str(path).encode(), C.byref(dim))) else 0

    def release_octree(self) -> bool:
        return self._ok(lib.vrc_release_octree(self._h))

    def assign_camera(self, direction: np.ndarray, position: np.ndarray) -> bool:
        """direction: float32[2] (inclination, azimuth); position: float32[3].  The arrays are
        retained and re-read at every compute(), like the reference's USE_HOST_PTR buffers."""
        assert direction.dtype == np.float32 and position.dtype == np.float32
        self._keep["cam"] = (direction, position)
        return self._ok(lib.vrc_assign_camera(self._h, _ptr(direction, _f32p), _ptr(position, _f32p)))

    def assign_camera_trig(self, trig: Optional[np.ndarray]) -> bool:
        """trig: float32[4] = sin / cos of direction[0], sin / cos of direction[1] as the host evaluates them (the reference
        kernel does it per pixel with the OpenCL library's, ray_caster_kernel.cl:280-291); retained and re-read at every
        compute() like the camera.  None: the library's own sinf / cosf of the live direction (the default)."""
        if trig is None:
            self._keep.pop("trig", None)
            return self._ok(lib.vrc_assign_camera_trig(self._h, None))
        assert trig.dtype == np.float32 and trig.size == 4 and trig.flags["C_CONTIGUOUS"]
        self._keep["trig"] = trig
        return self._ok(lib.vrc_assign_camera_trig(self._h, _ptr(trig, _f32p)))

    def release_camera(self) -> bool:
        self._keep.pop("cam", None)
        self._keep.pop("trig", None)
        return self._ok(lib.vrc_release_camera(self._h))

    def assign_lights(self, packed: np.ndarray, light_count: Optional[np.ndarray] = None) -> bool:
        """packed: float32[n,10] = rgbi[4], position[3], direction[3] (LightController.h:63-73)."""
        assert packed.dtype == np.float32
        if light_count is None:
            light_count = np.array([packed.reshape(-1, 10).shape[0]], dtype=np.int32)
        self._keep["lights"] = (packed, light_count)
        return self._ok(lib.vrc_assign_lights(self._h, _ptr(packed, _f32p), _ptr(light_count, _i32p)))

    def create_viewport(self, width: int, height: int, v_fov: float = 0.0, h_fov: float = 0.0) -> bool:
        self.viewport_size = (int(width), int(height))
        return self._ok(lib.vrc_create_viewport(self._h, width, height, v_fov, h_fov))

    def create_viewport_table(self, table: np.ndarray) -> bool:
        """Extension: host-supplied float4 ray table [h, w, 4]."""
        t = np.ascontiguousarray(table, dtype=np.float32)
        self.viewport_size = (int(t.shape[1]), int(t.shape[0]))
        return self._ok(lib.vrc_create_viewport_table(self._h, t.shape[1], t.shape[0], _ptr(t, _f32p)))

    def release_viewport(self) -> bool:
        return self._ok(lib.vrc_release_viewport(self._h))

    def device_image(self):
        """(device pointer, bytes) of the float4 frame in HBM, for consumers on the same GPU (vrc_device_image)."""
        ptr, n = C.c_void_p(), C.c_size_t()
        if not self._ok(lib.vrc_device_image(self._h, C.byref(ptr), C.byref(n))):
            raise VrcError(self.last_error())
        return ptr.value, n.value

    def create_texture_atlas(self, rgba8: np.ndarray, tile_dim) -> bool:
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        hgt, wid = a.shape[0], a.shape[1]
        return self._ok(lib.vrc_create_texture_atlas(self._h, _ptr(a, _u8p), wid, hgt, int(tile_dim[0]), int(tile_dim[1])))

    # -- settings (CLCaster.cpp:1029-1109)
    def add_to_settings_buffer(self, setting_name: str, define_accessor_name: str, value: int) -> bool:
        return self._ok(lib.vrc_setting_add(self._h, setting_name.encode(), define_accessor_name.encode(), int(value)))

    def overwrite_setting(self, setting_name: str, value: int) -> bool:
        return self._ok(lib.vrc_setting_set(self._h, setting_name.encode(), int(value)))

    def get_setting(self, setting_name: str) -> Optional[int]:
        v = C.c_int64()
        if not self._ok(lib.vrc_setting_get(self._h, setting_name.encode(), C.byref(v))):
            return None
        return int(v.value)

    # -- validate / compute (CLCaster.cpp:157-228)
    def validate(self) -> bool:
        return self._ok(lib.vrc_validate(self._h))

    def prepare(self) -> bool:
        """Builds what the SVO kernels derive from the tree (coarse table, empty boxes) for the current settings; validate()
        ends with it, so this is for settings changed afterwards."""
        return self._ok(lib.vrc_prepare(self._h))

    def compute(self) -> bool:
        return self._ok(lib.vrc_compute(self._h))

    def compute_async(self) -> bool:
        return self._ok(lib.vrc_compute_async(self._h))

    def sync(self) -> bool:
        return self._ok(lib.vrc_sync(self._h))

    def set_row_tiling(self, rank: int, world: int, band_rows: int = 8) -> bool:
        return self._ok(lib.vrc_set_row_tiling(self._h, rank, world, band_rows))

    def set_row_slice(self, rank: int, world: int, band_rows: int = 8) -> bool:
        """Like set_row_tiling, but the buffers hold only this rank's rows (call before create_viewport)."""
        return self._ok(lib.vrc_set_row_slice(self._h, rank, world, band_rows))

    # -- output (replaces CLCaster::draw, CLCaster.cpp:330-332)
    @staticmethod
    def _check_out(out: np.ndarray, dtype, shape, what: str) -> None:
        """A caller-supplied destination is written through a raw pointer: it must be exactly what the library assumes."""
        if not isinstance(out, np.ndarray) or out.dtype != np.dtype(dtype) or out.shape != shape or not out.flags.c_contiguous \
                or not out.flags.writeable:
            raise VrcError(f"{what}: `out` must be a writable C-contiguous {np.dtype(dtype).name} array of shape {shape}, got "
                           f"{getattr(out, 'dtype', type(out))} {getattr(out, 'shape', '')}")

    def read_image(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        """The float4 frame.  A row-sliced handle fills only its own rows of `out` (pass the same array to every rank)."""
        w, h = self.viewport_size
        if out is None:
            out = np.empty((h, w, 4), dtype=np.float32)
        self._check_out(out, np.float32, (h, w, 4), "read_image")
        if not self._ok(lib.vrc_read_image_f32(self._h, _ptr(out, _f32p), out.size)):
            raise VrcError(self.last_error())
        return out

    def read_image_rgba8(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        w, h = self.viewport_size
        if out is None:
            out = np.empty((h, w, 4), dtype=np.uint8)
        self._check_out(out, np.uint8, (h, w, 4), "read_image_rgba8")
        if not self._ok(lib.vrc_read_image_rgba8(self._h, _ptr(out, _u8p), out.size)):
            raise VrcError(self.last_error())
        return out

    def read_hits(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        w, h = self.viewport_size
        if out is None:
            out = np.empty((h, w, 8), dtype=np.int32)
        self._check_out(out, np.int32, (h, w, 8), "read_hits")
        if not self._ok(lib.vrc_read_hits(self._h, _ptr(out, _i32p), out.size)):
            raise VrcError(self.last_error())
        return out

    def counters(self) -> dict:
        c = Counters()
        if not self._ok(lib.vrc_get_counters(self._h, C.byref(c))):
            raise VrcError(self.last_error())
        d = c.as_dict()
        # descriptor_reads (and field 7 of the hit records) is SURVEY 8d's canonical count unless the frame used the empty boxes
        canonical = C.c_int32(-1)
        if lib.vrc_counters_canonical(self._h, C.byref(canonical)) != 0:
            raise RuntimeError(self.last_error())
        d["canonical_reads"] = bool(canonical.value)
        return d

    def scheduler_stats(self) -> dict:
        out = (C.c_uint64 * 8)()
        if not self._ok(lib.vrc_get_scheduler_stats(self._h, out)):
            raise VrcError(self.last_error())
        names = ("wave_step_iterations", "bursts", "event_passes", "event_lanes", "shade_passes", "shade_lanes", "jump_attempts", "jump_successes")
        return {n: int(out[i]) for i, n in enumerate(names)}

    def timing_reset(self) -> bool:
        return self._ok(lib.vrc_timing_reset(self._h))

    def timing(self):
        n = C.c_uint64()
        ms = C.c_double()
        if not self._ok(lib.vrc_timing_get(self._h, C.byref(n), C.byref(ms))):
            raise VrcError(self.last_error())
        return int(n.value), float(ms.value)
