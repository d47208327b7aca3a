"""ctypes binding of the CPU oracle (oracle/libvrc_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg as the checker -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvrc_oracle.so")
MAX_DEPTH = 32


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vrc_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libvrc_oracle.so"])
    return LIB_PATH


build()
lib = C.CDLL(LIB_PATH)

_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i8p = C.POINTER(C.c_int8)


class TraversalState(C.Structure):
    _fields_ = [
        ("sub_oct_pos", C.c_int32 * 3), ("parent_stack_position", C.c_int32),
        ("parent_stack", C.c_uint64 * MAX_DEPTH), ("parent_stack_index", C.c_uint64 * MAX_DEPTH),
        ("scale", C.c_uint8), ("idx_stack", C.c_uint8 * MAX_DEPTH),
        ("current_descriptor", C.c_uint64), ("current_descriptor_index", C.c_uint64),
        ("oct_pos", C.c_int32 * 3), ("resolution", C.c_int32), ("found", C.c_int8), ("reads", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [
        ("map", _i8p), ("map_dim", C.c_int32 * 3), ("resolution", C.c_int32 * 2),
        ("viewport_matrix", _f32p), ("cam_dir", C.c_float * 2), ("cam_pos", C.c_float * 3),
        ("lights", _f32p), ("light_count", C.c_int32),
        ("atlas_rgba8", _u8p), ("atlas_dim", C.c_int32 * 2), ("tile_dim", C.c_int32 * 2),
        ("descriptors", _u64p), ("n_descriptors", C.c_uint64),
        ("attachment_lookup", C.POINTER(C.c_uint32)), ("attachments", _u64p),
        ("octree_dimensions", C.c_int64), ("using_octree", C.c_int64), ("octree_root_index", C.c_int64),
        ("max_distance", C.c_int32), ("shadow_rays", C.c_int32), ("no_bias", C.c_int32), ("active_lights", C.c_int32),
        ("cam_trig", C.c_float * 4), ("stepping_mode", C.c_int32), ("coarse_log2", C.c_int32),
        ("desc_pages", C.POINTER(C.c_void_p)), ("desc_page_fetch", C.c_void_p), ("desc_page_user", C.c_void_p)]


PAGE_SHIFT = 12
PAGE_SIZE = 1 << PAGE_SHIFT
PAGE_FETCH = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)


class PagedDescriptors:
    """Descriptor source for trees that exist only in GPU memory: read(first, count) -> uint64 array is called once per
    page of PAGE_SIZE descriptors the oracle touches; pages are kept for the lifetime of this object."""

    def __init__(self, n_descriptors: int, read):
        self.n = int(n_descriptors)
        self.read = read
        self.n_pages = (self.n + PAGE_SIZE - 1) >> PAGE_SHIFT
        self.table = np.zeros(self.n_pages, dtype=np.uint64)       # the oracle's page table (pointers)
        self.pages = {}

        def fetch(_user, page):
            first = int(page) << PAGE_SHIFT
            buf = np.zeros(PAGE_SIZE, dtype=np.uint64)
            cnt = min(PAGE_SIZE, self.n - first)
            buf[:cnt] = self.read(first, cnt)
            self.pages[int(page)] = buf
            return buf.ctypes.data

        self.callback = PAGE_FETCH(fetch)

    @property
    def bytes_fetched(self) -> int:
        return len(self.pages) * PAGE_SIZE * 8


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("primary_rays", "shadow_rays", "n_desc", "n_tex", "n_map", "n_steps", "unwritten")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


lib.orc_octree_generate.restype = C.c_int
lib.orc_octree_generate.argtypes = [_i8p, C.c_int, _u64p, C.c_uint64, _u64p, _u64p]
lib.orc_get_oct_vox.restype = None
lib.orc_get_oct_vox.argtypes = [_i32p, _u64p, C.c_uint64, C.c_int32, C.POINTER(TraversalState)]
lib.orc_octree_validate.restype = C.c_int64
lib.orc_octree_validate.argtypes = [_i8p, C.c_int, _u64p, C.c_uint64]
lib.orc_create_viewport.restype = None
lib.orc_create_viewport.argtypes = [C.c_int32, C.c_int32, _f32p]
lib.orc_camera_trig.restype = None
lib.orc_camera_trig.argtypes = [_f32p, _f32p]
lib.orc_view_light.restype = None
lib.orc_view_light.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _i32p]
lib.orc_raycast.restype = None
lib.orc_raycast.argtypes = [C.POINTER(Scene), C.c_int32, C.c_int32, _f32p, _i32p, C.POINTER(Counters), C.c_int]
lib.orc_clear_image.restype = None
lib.orc_clear_image.argtypes = [_f32p, C.c_int64]
lib.orc_image_to_rgba8.restype = None
lib.orc_image_to_rgba8.argtypes = [_f32p, _u8p, C.c_int64]
lib.orc_ray_cast.restype = C.c_uint32
lib.orc_ray_cast.argtypes = [_i8p, _i32p, _f32p, _f32p, C.c_int, _i32p]


def _p(a, ty):
    return a.ctypes.data_as(ty)


def octree_generate(grid: np.ndarray, dim: int, buffer_size: int = 100000):
    """Octree::Generate into a fixed buffer filled from the end (Octree.h:29).  Returns
    (buffer uint64[buffer_size], root_index) or raises on underflow."""
    grid = np.ascontiguousarray(grid, dtype=np.int8).reshape(-1)
    buf = np.zeros(buffer_size, dtype=np.uint64)
    root = C.c_uint64()
    low = C.c_uint64()
    rc = lib.orc_octree_generate(_p(grid, _i8p), dim, _p(buf, _u64p), buffer_size, C.byref(root), C.byref(low))
    if rc != 0:
        raise OverflowError("descriptor buffer too small (the reference would corrupt memory here)")
    return buf, int(root.value)


def get_oct_vox(position, descriptors: np.ndarray, root_index: int, dim: int) -> TraversalState:
    ts = TraversalState()
    pos = (C.c_int32 * 3)(*[int(v) for v in position])
    lib.orc_get_oct_vox(pos, _p(descriptors, _u64p), root_index, dim, C.byref(ts))
    return ts


def view_light(in_color, light, light_color, view, mask) -> np.ndarray:
    """view_light (ray_caster_kernel.cl:78-99) for one case."""
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (in_color, light, light_color, view)]
    m = np.ascontiguousarray(mask, dtype=np.int32)
    out = np.zeros(4, dtype=np.float32)
    lib.orc_view_light(_p(out, _f32p), _p(a[0], _f32p), _p(a[1], _f32p), _p(a[2], _f32p), _p(a[3], _f32p), _p(m, _i32p))
    return out


def octree_validate(grid, dim, descriptors, root_index) -> int:
    grid = np.ascontiguousarray(grid, dtype=np.int8).reshape(-1)
    return int(lib.orc_octree_validate(_p(grid, _i8p), dim, _p(descriptors, _u64p), root_index))


def create_viewport(w: int, h: int) -> np.ndarray:
    t = np.zeros((h, w, 4), dtype=np.float32)
    lib.orc_create_viewport(w, h, _p(t, _f32p))
    return t


def camera_trig(cam_dir) -> np.ndarray:
    d = np.asarray(cam_dir, dtype=np.float32)
    t = np.zeros(4, dtype=np.float32)
    lib.orc_camera_trig(_p(d, _f32p), _p(t, _f32p))
    return t


def raycast(*, width, height, cam_dir, cam_pos, lights, atlas, tile_dim, descriptors, root_index, octree_dim,
            using_octree, grid=None, map_dim=None, max_distance=20, shadow_rays=1, viewport=None, trig=None,
            rows=None, threads=1, want_hits=True, attachment_lookup=None, attachments=None, active_lights=1, no_bias=0,
            stepping_mode=0, coarse_log2=-1):
    """Render with the oracle.  Returns (image[h,w,4] f32, hits[h,w,8] i32 or None, counters dict)."""
    keep = []
    s = Scene()
    if grid is not None:
        g = np.ascontiguousarray(grid, dtype=np.int8).reshape(-1)
        keep.append(g)
        s.map = _p(g, _i8p)
    md = map_dim if map_dim is not None else (octree_dim,) * 3
    s.map_dim = (C.c_int32 * 3)(*md)
    s.resolution = (C.c_int32 * 2)(width, height)
    vp = viewport if viewport is not None else create_viewport(width, height)
    vp = np.ascontiguousarray(vp, dtype=np.float32)
    keep.append(vp)
    s.viewport_matrix = _p(vp, _f32p)
    s.cam_dir = (C.c_float * 2)(*[float(v) for v in cam_dir])
    s.cam_pos = (C.c_float * 3)(*[float(v) for v in cam_pos])
    li = np.ascontiguousarray(lights, dtype=np.float32).reshape(-1)
    keep.append(li)
    s.lights = _p(li, _f32p)
    s.light_count = li.size // 10
    at = np.ascontiguousarray(atlas, dtype=np.uint8)
    keep.append(at)
    s.atlas_rgba8 = _p(at, _u8p)
    s.atlas_dim = (C.c_int32 * 2)(at.shape[1], at.shape[0])
    s.tile_dim = (C.c_int32 * 2)(*tile_dim)
    if isinstance(descriptors, PagedDescriptors):
        s.n_descriptors = descriptors.n
        s.desc_pages = descriptors.table.ctypes.data_as(C.POINTER(C.c_void_p))
        s.desc_page_fetch = C.cast(descriptors.callback, C.c_void_p)
    else:
        de = np.ascontiguousarray(descriptors, dtype=np.uint64)
        keep.append(de)
        s.descriptors = _p(de, _u64p)
        s.n_descriptors = de.size
    if attachment_lookup is not None and attachments is not None:
        al = np.ascontiguousarray(attachment_lookup, dtype=np.uint32)
        ab = np.ascontiguousarray(attachments, dtype=np.uint64)
        keep += [al, ab]
        s.attachment_lookup = _p(al, C.POINTER(C.c_uint32))
        s.attachments = _p(ab, _u64p)
    s.octree_dimensions = octree_dim
    s.using_octree = using_octree
    s.octree_root_index = root_index
    s.max_distance = max_distance
    s.shadow_rays = shadow_rays
    s.active_lights = active_lights
    s.no_bias = no_bias
    s.stepping_mode = stepping_mode
    s.coarse_log2 = coarse_log2
    tr = np.asarray(trig, dtype=np.float32) if trig is not None else camera_trig(np.asarray(cam_dir, dtype=np.float32))
    s.cam_trig = (C.c_float * 4)(*[float(v) for v in tr])
    image = np.zeros((height, width, 4), dtype=np.float32)
    lib.orc_clear_image(_p(image, _f32p), width * height)
    hits = np.zeros((height, width, 8), dtype=np.int32) if want_hits else None
    if hits is not None:
        hits[..., 0:3] = -1
    ctr = Counters()
    y0, y1 = rows if rows is not None else (0, height)
    lib.orc_raycast(C.byref(s), y0, y1, _p(image, _f32p), _p(hits, _i32p) if want_hits else None, C.byref(ctr), threads)
    return image, hits, ctr.as_dict()


def image_to_rgba8(image: np.ndarray) -> np.ndarray:
    image = np.ascontiguousarray(image, dtype=np.float32)
    out = np.zeros(image.shape, dtype=np.uint8)
    lib.orc_image_to_rgba8(_p(image, _f32p), _p(out, _u8p), image.size // 4)
    return out


def ray_cast(grid, dim, origin, direction, as_written=True):
    g = np.ascontiguousarray(grid, dtype=np.int8).reshape(-1) if grid is not None else np.zeros(1, np.int8)
    d = (C.c_int32 * 3)(*dim)
    o = (C.c_float * 3)(*[float(v) for v in origin])
    r = (C.c_float * 3)(*[float(v) for v in direction])
    steps = C.c_int32()
    col = lib.orc_ray_cast(_p(g, _i8p), d, o, r, int(as_written), C.byref(steps))
    return (col & 255, (col >> 8) & 255, (col >> 16) & 255, (col >> 24) & 255), int(steps.value)


lib.orc_ray_cast_frame.restype = C.c_int64
lib.orc_ray_cast_frame.argtypes = [_i8p, _i32p, C.c_int32, C.c_int32, _f32p, _f32p, _f32p, C.POINTER(C.c_uint32), C.c_int]


def ray_cast_frame(grid, dim, width, height, cam_dir, cam_pos, threads=1):
    """Ray::Cast (restored) for every pixel of a frame; returns (rgba uint32[h,w], total DDA steps)."""
    g = np.ascontiguousarray(grid, dtype=np.int8).reshape(-1)
    d = (C.c_int32 * 3)(*dim)
    vp = create_viewport(width, height)
    trig = camera_trig(np.asarray(cam_dir, dtype=np.float32))
    pos = np.asarray(cam_pos, dtype=np.float32)
    out = np.zeros((height, width), dtype=np.uint32)
    steps = lib.orc_ray_cast_frame(_p(g, _i8p), d, width, height, _p(vp, _f32p), _p(trig, _f32p), _p(pos, _f32p),
                                   _p(out, C.POINTER(C.c_uint32)), threads)
    return out, int(steps)
