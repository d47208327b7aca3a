"""GPU suite (-m gpu): the hand-written gfx950 path, called through the C ABI (libvrc.so via
voxel_raycaster_amd.CLCaster), against the CPU oracle on the same seeded inputs.

Bar: bit-exact on hit voxel / material / face / flags / step counts and canonical descriptor-read
counts (integer work); RGB floats are required bit-exact as well, which is stronger than the
1e-5 relative tolerance BASELINE.json states (both sides evaluate the same IEEE expression tree).
"""
import glob
import os

import numpy as np
import pytest

import scenes
import voxel_raycaster_amd as vrc
from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "orc_*.npz")))
RTOL = 1e-5  # BASELINE.json north_star tolerance for RGB floats


def make_caster(octree, dim, using_octree, cam_dir, cam_pos, lights, atlas, w, h, max_distance, grid=None,
                shadow_rays=1, light_count=None, empty_boxes=None, tree_from=None):
    c = vrc.CLCaster()
    assert c.init(0), "vrc_create failed: is this a GPU box?"
    assert c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim)          # Application.cpp:35
    assert c.add_to_settings_buffer("using_octree", "OCTENABLED", using_octree)  # Application.cpp:38-39
    assert c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", max_distance)
    assert c.add_to_settings_buffer("shadow_rays", "SHADOW_RAYS", shadow_rays)
    if light_count is not None:                                                  # multi-light extension (8f-1)
        assert c.add_to_settings_buffer("light_count", "LIGHT_COUNT", light_count)
    if empty_boxes is not None:                                                  # None: the library's rule (on where the tree has a coarse table)
        assert c.add_to_settings_buffer("empty_boxes", "EMPTY_BOXES", empty_boxes)
    # tree_from: a caster on this GPU that already holds this very tree -- adopted (vrc_assign_octree_from: one array, one coarse
    # table, one set of boxes), not uploaded and annotated again
    assert c.assign_octree_from(tree_from) if tree_from is not None else c.assign_octree(octree), c.last_error()
    if grid is not None:
        assert c.assign_map(grid, (dim, dim, dim))
    cd, cp = np.array(cam_dir, dtype=np.float32), np.array(cam_pos, dtype=np.float32)
    assert c.assign_camera(cd, cp)
    assert c.create_viewport(w, h, 0.0, 0.0), c.last_error()
    li = np.zeros((8, 10), dtype=np.float32)
    li[: np.asarray(lights).reshape(-1, 10).shape[0]] = np.asarray(lights).reshape(-1, 10)
    assert c.assign_lights(li)
    assert c.create_texture_atlas(atlas, (16, 16))
    assert c.validate(), c.last_error()
    c._li = li
    return c


def hits_match(c, hits, ohits):
    """Hit records against the oracle's: every field.  Field 7 (descriptor reads) is SURVEY 8d's canonical count only when the
    frame was rendered by the canonical traversal; with the tree's empty boxes (the default where they exist, setting
    empty_boxes) the kernel counts the reads IT makes, and the field is left out of the comparison."""
    canonical = c if isinstance(c, bool) else not c.used_empty_boxes()
    return np.array_equal(hits, ohits) if canonical else np.array_equal(hits[..., :7], ohits[..., :7])


def assert_same(img, hits, ctr, oimg, ohits, octr):
    canonical = ctr.get("canonical_reads", True)
    assert hits_match(canonical, hits, ohits), f"{int((hits[..., :7] != ohits[..., :7]).any(-1).sum())} pixels differ in hit records"
    err = np.abs(img - oimg) / np.maximum(np.abs(oimg), 1e-6)
    assert np.nanmax(err) <= RTOL
    assert np.array_equal(img.view(np.uint32), oimg.view(np.uint32)), "floats within tolerance but not bit-exact"
    assert ctr["primary_rays"] == octr["primary_rays"] and ctr["shadow_rays"] == octr["shadow_rays"]
    assert (ctr["descriptor_reads"] == octr["n_desc"] or not canonical) and ctr["texel_reads"] == octr["n_tex"]
    assert ctr["map_reads"] == octr["n_map"] and ctr["steps"] == octr["n_steps"]
    assert ctr["unwritten_pixels"] == octr["unwritten"]


@pytest.mark.parametrize("branch", ["array", "svo", "svo-canonical"])
@pytest.mark.parametrize("make", scenes.ALL, ids=[f.__name__ for f in scenes.ALL])
@pytest.mark.parametrize("res", [(160, 120), (97, 61)], ids=["160x120", "ragged97x61"])
def test_hip_equals_oracle(make, branch, res, atlas):
    """svo: the product's default -- empty boxes where the tree has a coarse table (32^3 and up); everything but the
    descriptor-read count is compared.  svo-canonical: empty_boxes = 0, the canonical traversal, read counts included."""
    using_octree = 1 if branch == "array" else 0
    s = make()
    dim, (w, h) = s["dim"], res
    m = vrc.Map(dim, s["grid"], buffer_size=100000)
    md = 20 if dim <= 16 else 3 * dim
    c = make_caster(m.octree, dim, using_octree, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md, grid=s["grid"],
                    empty_boxes=0 if branch == "svo-canonical" else None)
    assert c.compute(), c.last_error()
    if branch == "svo-canonical":
        assert not c.used_empty_boxes()
    elif branch == "svo" and dim >= 32:
        assert c.used_empty_boxes() and c.empty_boxes_check(1 << 16)["solid_voxels"] == 0
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li,
                                    atlas=atlas, tile_dim=(16, 16), descriptors=m.octree.descriptor_buffer,
                                    root_index=m.octree.root_index, octree_dim=dim, using_octree=using_octree,
                                    grid=s["grid"], max_distance=md)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    # RGBA8 read-back == UNORM8 quantisation of the float frame
    assert np.array_equal(c.read_image_rgba8(), orc.image_to_rgba8(oimg))


@pytest.mark.parametrize("boxes", [None, 0], ids=["default", "canonical"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_reproduces_committed_vectors(path, boxes):
    import golden_io
    g = golden_io.load(path)
    dim, w, h = int(g["dim"]), int(g["width"]), int(g["height"])
    o = vrc.Octree(g["descriptors"], int(g["root_index"]), dim)
    c = make_caster(o, dim, int(g["using_octree"]), g["cam_dir"], g["cam_pos"], g["lights"], g["atlas"], w, h,
                    int(g["max_distance"]), grid=g["grid"], light_count=int(g.get("active_lights", 1)), empty_boxes=boxes)
    assert c.compute(), c.last_error()
    assert hits_match(c, c.read_hits(), g["hits"])
    assert np.array_equal(c.read_image().view(np.uint32), g["image"].view(np.uint32))


@pytest.mark.parametrize("make", [scenes.mirror_wall, scenes.floor_pillars, scenes.random_sparse],
                         ids=["mirror_wall", "floor_pillars", "random_sparse"])
def test_svo_with_attachments_equals_array_and_oracle(make, atlas):
    """Per-voxel materials through the attachment buffers: the SVO kernel must equal the oracle
    bit for bit and render what the array kernel renders (mirrors bounce, other materials pass through)."""
    from test_oracle_cpu import _with_pass_through
    s = _with_pass_through(make())
    dim, w, h = s["dim"], 160, 120
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000).attach_materials_from_grid(s["grid"])
    md = 3 * dim
    svo = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
    arr = make_caster(o, dim, 1, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md, grid=s["grid"])
    assert svo.compute() and arr.compute()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=svo._li,
                                    atlas=atlas, tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                    octree_dim=dim, using_octree=0, max_distance=md, attachment_lookup=o.attachment_lookup,
                                    attachments=o.attachment_buffer)
    assert_same(svo.read_image(), svo.read_hits(), svo.counters(), oimg, ohits, octr)
    assert np.array_equal(svo.read_image().view(np.uint32), arr.read_image().view(np.uint32))
    assert np.array_equal(svo.read_hits()[..., :7], arr.read_hits()[..., :7])
    if make is scenes.mirror_wall:
        assert (ohits[..., 3] == 6).sum() > 0


@pytest.mark.parametrize("make", [scenes.floor_pillars, scenes.open_sky, scenes.random_sparse, scenes.mirror_wall],
                         ids=["floor_pillars", "open_sky", "random_sparse", "mirror_wall"])
def test_exact_jump_kernel_equals_oracle(make, atlas):
    """Opt-in closed-form jumps (csrc/exact_jump.hpp, setting jump_min_run): same bits as the stepping kernel."""
    s = make()
    dim, w, h = s["dim"], 160, 120
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000).attach_materials_from_grid(s["grid"])
    md = 3 * dim
    c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
    assert c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 2)
    assert c.compute()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li,
                                    atlas=atlas, tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                    octree_dim=dim, using_octree=0, max_distance=md, attachment_lookup=o.attachment_lookup,
                                    attachments=o.attachment_buffer)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)


def test_exact_jump_kernel_headline_frame():
    """The jump kernel on the full depth-12 1080p frame must reproduce the stepping kernel bit for bit."""
    sc = _bench_scene(12)
    w, h, dim = 1920, 1080, sc["dim"]
    c = make_caster(sc["octree"], dim, 0, sc["cam_dir"], sc["cam_pos"], sc["lights"], sc["atlas"], w, h, 3 * dim)
    assert c.compute()
    img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
    assert c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 8)
    assert c.compute()
    assert np.array_equal(c.read_hits(), hits)
    assert np.array_equal(c.read_image().view(np.uint32), img.view(np.uint32))
    assert c.counters() == ctr


EDGE_CASES = [
    # name, dim, camera position, camera direction, max_distance
    ("camera_outside_map", 16, (-3.4, 5.3, 6.2), (1.8, 0.3), 40),
    ("camera_above_map", 16, (7.3, 8.6, 19.7), (2.6, 1.1), 40),
    ("camera_on_integer_coordinates", 16, (8.0, 3.0, 9.0), (2.0, 1.5708), 40),
    ("zero_step_cap", 16, (8.4, 3.3, 9.2), (2.0, 1.5708), 0),
    ("one_step_cap", 16, (8.4, 3.3, 9.2), (2.0, 1.5708), 1),
    ("tiny_map_dim2", 2, (0.6, 0.4, 1.3), (2.2, 0.9), 8),
    ("tiny_map_dim4", 4, (1.6, 0.4, 3.3), (2.2, 0.9), 12),
    ("camera_inside_solid", 32, (16.5, 16.5, 1.5), (1.2, 2.0), 96),
    # t values below 2^-100: the step loop's arithmetic face mask must hand over to the compare/select loop
    ("camera_at_tiny_offsets", 16, (1e-36, 3e-37, 9.0 + 2.0 ** -20), (2.0, 1.5708), 40),
    ("camera_at_denormal_offsets", 16, (1e-41, 5e-42, 1e-39), (0.9, 0.7), 40),
]


@pytest.mark.parametrize("using_octree", [1, 0], ids=["array", "svo"])
@pytest.mark.parametrize("case", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_edge_cases(case, using_octree, atlas):
    name, dim, cam_pos, cam_dir, md = case
    rng = np.random.default_rng(dim)
    g = (rng.random((dim, dim, dim)) < 0.15).astype(np.int8) * 5
    g[0:max(1, dim // 8)] = 5
    g = g.reshape(-1)
    o = vrc.Octree.Generate(g, dim, buffer_size=100000)
    lights = np.array([[0.01, 0.01, 0.01, 0.2, dim * 0.7, dim * 0.3, dim * 0.9, -1, -1, -1.5]], dtype=np.float32)
    w, h = 72, 40
    c = make_caster(o, dim, using_octree, cam_dir, cam_pos, lights, atlas, w, h, md, grid=g)
    assert c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                    octree_dim=dim, using_octree=using_octree, grid=g, max_distance=md)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)


@pytest.mark.parametrize("make", [scenes.floor_pillars, scenes.random_sparse, scenes.app_default],
                         ids=["floor_pillars", "random_sparse", "app_default"])
def test_step_loop_variants_agree(make, atlas):
    """Settings arith_mask / safe_run pick how the SVO step loop forms face_mask (v_cmp/v_cndmask or subtract +
    clamped fma) and whether lanes deep inside a node step without countdowns (csrc/safe_run.hpp): speed knobs,
    same frame every way."""
    s = make()
    dim, w, h = s["dim"], 160, 120
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    md = 20 if dim <= 16 else 3 * dim
    frames = []
    for arith, safe, safe_steps, single in ((0, 0, 64, 1), (1, 0, 64, 1), (1, 1, 64, 1), (1, 1, 256, 0), (1, 1, 16, 1), (1, 1, 8, 1),
                                            (1, 1, 2, 1), (1, 1, 64, 0), (1, 0, 64, 0)):
        c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
        assert c.add_to_settings_buffer("single_step", "SINGLE_STEP", single)
        assert c.add_to_settings_buffer("arith_mask", "ARITH_MASK", arith)
        assert c.add_to_settings_buffer("safe_run", "SAFE_RUN", safe)
        assert c.add_to_settings_buffer("safe_steps", "SAFE_STEPS", safe_steps) and c.compute(), c.last_error()
        frames.append((c.read_image(), c.read_hits(), c.counters()))
    for f in frames[1:]:
        assert np.array_equal(frames[0][0].view(np.uint32), f[0].view(np.uint32))
        assert np.array_equal(frames[0][1], f[1]) and frames[0][2] == f[2]


KNOBS = [dict(widen_nodes=0), dict(xcd_mode=0), dict(xcd_mode=2), dict(burst_steps=1, safe_run=0), dict(burst_steps=7, arith_mask=0),
         dict(shade_threshold=1), dict(shade_threshold=16), dict(exact_steps=1), dict(exact_steps=2, single_step=0),
         dict(exact_steps=64), dict(lds_pad_bytes=8192), dict(safe_steps=24), dict(safe_steps=100, exact_steps=3),
         dict(widen_nodes=0, safe_steps=16, shade_threshold=8, xcd_mode=0)]


@pytest.mark.parametrize("make", [scenes.floor_pillars, scenes.random_sparse], ids=["floor_pillars", "random_sparse"])
def test_scheduling_knobs_never_change_the_frame(make, atlas):
    """widen_nodes, xcd_mode, burst_steps, shade_threshold, exact_steps, safe_steps, lds_pad_bytes only move
    work around inside the kernel -- frame, hit records and counters stay bit-identical (and equal to the oracle's)."""
    s = make()
    dim, w, h = s["dim"], 200, 136
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000).attach_materials_from_grid(s["grid"])
    md = 3 * dim
    base = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
    assert base.compute()
    ref = (base.read_image(), base.read_hits(), base.counters())
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=base._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                    octree_dim=dim, using_octree=0, max_distance=md, attachment_lookup=o.attachment_lookup,
                                    attachments=o.attachment_buffer)
    assert_same(*ref, oimg, ohits, octr)
    for knobs in KNOBS:
        c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
        for k, v in knobs.items():
            assert c.add_to_settings_buffer(k, k.upper(), v)
        assert c.compute(), (knobs, c.last_error())
        assert np.array_equal(c.read_image().view(np.uint32), ref[0].view(np.uint32)), knobs
        assert np.array_equal(c.read_hits(), ref[1]) and c.counters() == ref[2], knobs


@pytest.mark.parametrize("using_octree", [1, 0], ids=["array", "svo"])
def test_diamond_square_terrain_scene(using_octree, atlas):
    """The reference's own terrain generator (Map::GenerateHeightBitmap) as a scene: 128^3, camera over
    the hills, light high above -- both kernels bit-exact vs the oracle."""
    dim, w, h, md = 128, 192, 128, 3 * 128
    height, grid = vrc.diamond_square(dim)
    o = vrc.Octree.Generate(grid, dim)
    cam_pos = (dim * 0.5 + 0.3, dim * 0.1 + 0.2, 40.3)
    cam_dir = (1.67, 1.5708)
    lights = np.array([[0.01, 0.01, 0.01, 0.2, dim * 0.3, dim * 0.6, dim * 0.95, -1, -1, -1.5]], dtype=np.float32)
    c = make_caster(o, dim, using_octree, cam_dir, cam_pos, lights, atlas, w, h, md, grid=grid)
    assert c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                    octree_dim=dim, using_octree=using_octree, grid=grid, max_distance=md)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    assert (ohits[..., 3] == 5).mean() > 0.3 and octr["shadow_rays"] > 0.3 * w * h


@pytest.mark.parametrize("using_octree", [1, 0], ids=["array", "svo"])
def test_octree_bias_can_be_switched_off(using_octree, atlas):
    """The reference adds (sub_oct_pos - voxel) * resolution / 2 to intersection_t (:353-354), which shears the
    picture whenever the camera is in an empty node whose corner is not the camera voxel.  Setting octree_bias = 0
    (extension) drops it; both ways equal the oracle, and they differ from each other in this scene."""
    s = scenes.open_sky()
    dim, w, h, md = s["dim"], 128, 96, 3 * s["dim"]
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    frames = []
    for bias in (1, 0):
        c = make_caster(o, dim, using_octree, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md, grid=s["grid"])
        assert c.add_to_settings_buffer("octree_bias", "OCTREE_BIAS", bias) and c.compute(), c.last_error()
        oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas,
                                        tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                        octree_dim=dim, using_octree=using_octree, grid=s["grid"], max_distance=md, no_bias=1 - bias)
        assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
        frames.append(c.read_hits())
    assert not np.array_equal(frames[0], frames[1])


def test_lifecycle_release_and_device_image(atlas):
    """release_* put the handle back into 'not ready' (validate/compute fail, nothing crashes), re-assigning makes it
    whole again with the same frame; vrc_device_image hands out the float4 frame in HBM for same-GPU consumers."""
    import torch
    s = scenes.floor_pillars()
    dim, w, h, md = s["dim"], 128, 96, 3 * s["dim"]
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    c = make_caster(o, dim, 1, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md, grid=s["grid"])
    assert c.compute()
    ref = c.read_image().copy()
    ptr, nbytes = c.device_image()
    assert nbytes == w * h * 16
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    host = np.zeros((h, w, 4), dtype=np.float32)
    assert hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) == 0   # device -> host
    assert np.array_equal(host.view(np.uint32), ref.view(np.uint32))
    for release, restore in ((c.release_camera, lambda: c.assign_camera(np.array(s["cam_dir"], np.float32), np.array(s["cam_pos"], np.float32))),
                             (c.release_viewport, lambda: c.create_viewport(w, h)),
                             (c.release_octree, lambda: c.assign_octree(o)),
                             (c.release_map, lambda: c.assign_map(s["grid"], (dim, dim, dim)))):
        assert release()
        assert not c.validate() and not c.compute()
        keep = restore()                                                        # arrays must outlive the handle's use
        assert keep and c.validate(), c.last_error()
        assert c.compute() and np.array_equal(c.read_image().view(np.uint32), ref.view(np.uint32))


def test_async_frames_timing_and_settings_queries(atlas):
    """vrc_compute_async / vrc_sync queue frames on the handle's stream (live camera pointers are read at enqueue time),
    vrc_timing_* reports one hipEvent-timed launch per frame, settings can be read back, and the profiling counters of
    the product build are zero (they exist only in the -DVRC_SCHED_STATS build)."""
    s = scenes.random_sparse()
    dim, w, h, md = s["dim"], 160, 120, 3 * s["dim"]
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
    assert c.get_setting("max_distance") == md and c.get_setting("using_octree") == 0
    assert c.get_setting("octree_root_index") == o.root_index and c.get_setting("no_such_setting") is None
    assert c.timing_reset()
    for _ in range(5):
        assert c.compute_async()
    assert c.sync()
    n, ms = c.timing()
    assert n == 5 and 0.0 < ms < 1000.0
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                    octree_dim=dim, using_octree=0, max_distance=md)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    assert all(v == 0 for v in c.scheduler_stats().values())


def test_round_watchdog_reports_instead_of_hanging(atlas):
    """A wave that would run more rounds than any legal frame needs is stopped; compute() itself then fails (a truncated
    frame never looks like success) and so does vrc_get_counters (setting watchdog_rounds only exists to provoke this);
    a normal frame never trips it."""
    s = scenes.floor_pillars()
    dim, w, h, md = s["dim"], 128, 96, 3 * s["dim"]
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
    assert c.compute()
    good = c.counters()
    assert good["primary_rays"] == w * h
    assert c.add_to_settings_buffer("watchdog_rounds", "WATCHDOG_ROUNDS", 2)
    assert c.compute() is False and c.last_status == 3 and "watchdog" in c.last_error()
    with pytest.raises(vrc.VrcError):
        c.counters()
    assert "watchdog" in c.last_error()
    assert c.overwrite_setting("watchdog_rounds", 1 << 30) and c.compute()
    assert c.counters() == good


def test_unnormalised_ray_table(atlas):
    """A host-supplied ray table need not be normalised (vrc_create_viewport_table): delta_t = |1/dir| then falls
    below 1 and the safe run must stand aside (csrc/safe_run.hpp); longer-than-unit and shorter-than-unit rays, SVO."""
    s = scenes.random_sparse()
    dim, w, h, md = s["dim"], 128, 96, 3 * s["dim"]
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    table = orc.create_viewport(w, h).reshape(h, w, 4).copy()
    table[: h // 3] *= np.float32(2.5)
    table[h // 3: 2 * h // 3] *= np.float32(0.4)
    c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
    assert c.create_viewport_table(table) and c.validate() and c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                    octree_dim=dim, using_octree=0, max_distance=md, viewport=table)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)


def test_streamed_upload_of_a_saved_tree(tmp_path, atlas):
    """vrc_assign_octree_file: a tree saved with its attachments and streamed from the file into
    device memory renders the frame the in-memory tree renders; truncated / foreign files are refused."""
    from test_oracle_cpu import _with_pass_through
    s = _with_pass_through(scenes.mirror_wall())
    dim, w, h, md = s["dim"], 160, 120, 3 * s["dim"]
    o = vrc.Octree.Generate(s["grid"], dim).attach_materials_from_grid(s["grid"])
    path = str(tmp_path / "scene.svo")
    o.Save(path)
    a = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
    assert a.compute()
    b = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
    assert b.release_octree()
    assert b.assign_octree_file(path) == dim and b.validate() and b.compute(), b.last_error()
    assert np.array_equal(a.read_image().view(np.uint32), b.read_image().view(np.uint32))
    assert np.array_equal(a.read_hits(), b.read_hits()) and a.counters() == b.counters()
    assert (a.read_hits()[..., 3] == 6).sum() > 0                  # materials came through the file
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[: len(raw) // 2])
    assert b.assign_octree_file(path) == 0 and "truncated" in b.last_error()
    assert not b.validate()                                         # the half-loaded tree is gone
    open(path, "wb").write(b"not a tree" * 10)
    assert b.assign_octree_file(path) == 0 and "VRCSVO01" in b.last_error()
    assert b.assign_octree_file(str(tmp_path / "missing.svo")) == 0


def test_non_cubic_dense_map(atlas):
    """The array branch takes any dx,dy,dz (kernel index x + dx*(y + dz*z), ray_caster_kernel.cl:569)."""
    dx, dy, dz = 24, 16, 16                       # dy == dz keeps the reference's dim.z-as-y-stride quirk in bounds
    g = np.zeros((dz, dy, dx), dtype=np.int8)
    g[0:2] = 5
    g[2:9, 8, 12] = 6
    g = g.reshape(-1)
    o = vrc.Octree.Generate(np.zeros(16 ** 3, dtype=np.int8), 16, buffer_size=100000)   # bias source only
    lights = np.array([[0.01, 0.01, 0.01, 0.2, 20.0, 3.0, 14.0, -1, -1, -1.5]], dtype=np.float32)
    w, h, md = 80, 48, 60
    c = vrc.CLCaster()
    assert c.init(0)
    assert c.add_to_settings_buffer("octree_dimensions", "OCTDIM", 16) and c.add_to_settings_buffer("using_octree", "OCTENABLED", 1)
    assert c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", md)
    assert c.assign_octree(o) and c.assign_map(g, (dx, dy, dz))
    cd, cp = np.array([1.9, 1.2], np.float32), np.array([3.3, 2.4, 9.6], np.float32)
    li = np.zeros((8, 10), np.float32); li[0] = lights[0]
    assert c.assign_camera(cd, cp) and c.create_viewport(w, h) and c.assign_lights(li) and c.create_texture_atlas(atlas, (16, 16))
    assert c.validate(), c.last_error()
    assert c.compute()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=cd, cam_pos=cp, lights=li, atlas=atlas, tile_dim=(16, 16),
                                    descriptors=o.descriptor_buffer, root_index=o.root_index, octree_dim=16, using_octree=1,
                                    grid=g, map_dim=(dx, dy, dz), max_distance=md)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)


def test_primary_only_and_live_camera(atlas):
    s = scenes.floor_pillars()
    dim = s["dim"]
    m = vrc.Map(dim, s["grid"])
    c = make_caster(m.octree, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, 128, 96, 3 * dim, shadow_rays=0)
    assert c.compute()
    oimg, ohits, octr = orc.raycast(width=128, height=96, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li,
                                    atlas=atlas, tile_dim=(16, 16), descriptors=m.octree.descriptor_buffer,
                                    root_index=m.octree.root_index, octree_dim=dim, using_octree=0, max_distance=3 * dim,
                                    shadow_rays=0)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    # the camera arrays are live (CL_MEM_USE_HOST_PTR semantics): mutate in place, recompute
    cd, cp = c._keep["cam"]
    cp[0] += 1.25
    cd[0] -= 0.2
    assert c.overwrite_setting("shadow_rays", 1)
    assert c.compute()
    oimg2, ohits2, octr2 = orc.raycast(width=128, height=96, cam_dir=cd, cam_pos=cp, lights=c._li, atlas=atlas,
                                       tile_dim=(16, 16), descriptors=m.octree.descriptor_buffer,
                                       root_index=m.octree.root_index, octree_dim=dim, using_octree=0,
                                       max_distance=3 * dim)
    # pixels the second frame does not write keep the first frame's contents
    expect = np.where((ohits2[..., 5:6] & 1) != 0, oimg2, oimg)
    assert np.array_equal(c.read_image().view(np.uint32), expect.view(np.uint32))
    assert hits_match(c, c.read_hits(), ohits2)


def test_error_behaviour_matches_the_boundary_contract(atlas):
    c = vrc.CLCaster()
    assert c.init(0)
    assert c.validate() is False and c.last_status == 2            # nothing assigned: NOT_READY, no abort
    assert "camera" in c.last_error()
    assert c.compute() is False
    assert c.overwrite_setting("no_such_setting", 1) is False and c.last_status == 5
    assert c.release_map() is False                                 # release before assign
    for i in range(64):
        assert c.add_to_settings_buffer(f"s{i}", f"S{i}", i)
    assert c.add_to_settings_buffer("one_too_many", "X", 0) is False and c.last_status == 6   # 64 slots
    assert c.set_row_tiling(2, 2, 8) is False and c.set_row_tiling(0, 1, 12) is False


def test_row_tiling_union_equals_full_frame(atlas):
    s = scenes.random_sparse()
    dim = s["dim"]
    m = vrc.Map(dim, s["grid"])
    w, h = 160, 120
    full = make_caster(m.octree, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, 3 * dim)
    assert full.compute()
    ref_img, ref_hits, ref_ctr = full.read_image(), full.read_hits(), full.counters()
    from voxel_raycaster_amd import tiling
    for world, band in [(2, 8), (3, 16), (8, 8)]:
        frames, hits, rays = [], [], 0
        for r in range(world):
            c = make_caster(m.octree, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, 3 * dim)
            assert c.set_row_tiling(r, world, band)
            assert c.compute()
            frames.append(c.read_image()); hits.append(c.read_hits())
            rays += c.counters()["primary_rays"]
            rows = tiling.rows_of_rank(h, r, world, band)
            other = np.setdiff1d(np.arange(h), rows)
            assert np.allclose(frames[-1][other], [1, 1, 1, 100 / 255])   # rows of other ranks untouched
        assert rays == ref_ctr["primary_rays"]
        assert np.array_equal(tiling.merge_tiles(frames, h, world, band).view(np.uint32), ref_img.view(np.uint32))
        assert np.array_equal(tiling.merge_tiles(hits, h, world, band), ref_hits)


def test_cpp_host_mirror(tmp_path, atlas):
    """csrc/app_init_example.cpp replays Application::init_clcaster (src/Application.cpp:27-88) through the C++
    CLCaster mirror (csrc/clcaster.hpp): its frames must equal the oracle's for the application's default scene."""
    import subprocess
    exe = os.path.join(ROOT, "voxel-raycaster_amd", "app_init_example")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    w, h = 96, 64
    atlas_path = tmp_path / "atlas.rgba"
    atlas.tofile(atlas_path)
    out = subprocess.run([exe, str(w), str(h), str(atlas_path), str(tmp_path / "frame")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    s = scenes.app_default()
    m = vrc.Map(16, buffer_size=100000)
    li = np.zeros((8, 10), dtype=np.float32)
    li[0] = s["lights"][0]
    for suffix, cam_pos in ((".image.f32", (2.34, 2.5, 7.17)), (".image2.f32", (2.34, 2.5, np.float32(7.17) + np.float32(0.5)))):
        img = np.fromfile(str(tmp_path / "frame") + suffix, dtype=np.float32).reshape(h, w, 4)
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=(2.424, 3.141), cam_pos=cam_pos, lights=li, atlas=atlas,
                                     tile_dim=(16, 16), descriptors=m.octree.descriptor_buffer,
                                     root_index=m.octree.root_index, octree_dim=16, using_octree=0, max_distance=20)
        assert np.array_equal(img.view(np.uint32), oimg.view(np.uint32))
    hits = np.fromfile(str(tmp_path / "frame") + ".hits.i32", dtype=np.int32).reshape(h, w, 8)
    oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=(2.424, 3.141), cam_pos=(2.34, 2.5, 7.17), lights=li, atlas=atlas,
                                 tile_dim=(16, 16), descriptors=m.octree.descriptor_buffer, root_index=m.octree.root_index,
                                 octree_dim=16, using_octree=0, max_distance=20)
    assert np.array_equal(hits, ohits)


import functools


@functools.lru_cache(maxsize=3)
def _bench_scene(depth):
    import bench
    return bench.build_scene(depth)


@pytest.mark.parametrize("depth,w,h", [(8, 640, 480), (10, 1920, 1080)], ids=["C1-d8-640x480", "C2-d10-1080p"])
def test_baseline_configs_sampled_rows(depth, w, h):
    """BASELINE configs[0]/[1] geometry (primary rays only): every 16th row bit-exact vs the oracle."""
    sc = _bench_scene(depth)
    c = make_caster(sc["octree"], sc["dim"], 0, sc["cam_dir"], sc["cam_pos"], sc["lights"], sc["atlas"], w, h,
                    3 * sc["dim"], shadow_rays=0)
    assert c.compute(), c.last_error()
    img, hits = c.read_image(), c.read_hits()
    for y0 in range(0, h, 16):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"], lights=c._li,
                                     atlas=sc["atlas"], tile_dim=(16, 16), descriptors=sc["octree"].descriptor_buffer,
                                     root_index=sc["octree"].root_index, octree_dim=sc["dim"], using_octree=0,
                                     max_distance=3 * sc["dim"], shadow_rays=0, rows=(y0, y0 + 1), threads=8)
        assert hits_match(c, hits[y0], ohits[y0])
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))


def test_8k_frame_sampled_rows():
    """BASELINE configs[4]'s frame size on the depth-12 scene: 7680x4320 (33 M pixels, 129 600 blocks, 1.6 GB of frame +
    hit records): sampled rows bit-exact vs the oracle."""
    sc = _bench_scene(12)
    dim, w, h = sc["dim"], 7680, 4320
    c = make_caster(sc["octree"], dim, 0, sc["cam_dir"], sc["cam_pos"], sc["lights"], sc["atlas"], w, h, 3 * dim)
    assert c.compute(), c.last_error()
    ctr = c.counters()
    assert ctr["primary_rays"] == w * h
    img, hits = c.read_image(), c.read_hits()
    for y0 in (3, 2161, 4316):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"], lights=c._li,
                                     atlas=sc["atlas"], tile_dim=(16, 16), descriptors=sc["octree"].descriptor_buffer,
                                     root_index=sc["octree"].root_index, octree_dim=dim, using_octree=0,
                                     max_distance=3 * dim, rows=(y0, y0 + 1), threads=16)
        assert hits_match(c, hits[y0], ohits[y0])
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))


@pytest.mark.parametrize("w,h,n", [(3840, 2160, 2), (1920, 1080, 4)], ids=["C4-4K-2lights", "d12-1080p-4lights"])
def test_multi_light_baseline_geometry_sampled_rows(w, h, n):
    """BASELINE configs[3] geometry on one GPU (depth-12 SVO, 3840x2160, 2 lights) and configs[4]'s light count on
    the headline scene: sampled rows bit-exact vs the oracle, shadow-ray counter = casts of every light."""
    sc = _bench_scene(12)
    dim = sc["dim"]
    c = make_caster(sc["octree"], dim, 0, sc["cam_dir"], sc["cam_pos"], sc["lights"], sc["atlas"], w, h, 3 * dim,
                    light_count=n)
    assert c.compute(), c.last_error()
    img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
    cast = (hits[..., 5] & 2) != 0
    assert cast.sum() < ctr["shadow_rays"] <= n * cast.sum() and ctr["primary_rays"] == w * h
    for y0 in range(29, h, 211):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"], lights=c._li,
                                     atlas=sc["atlas"], tile_dim=(16, 16), descriptors=sc["octree"].descriptor_buffer,
                                     root_index=sc["octree"].root_index, octree_dim=dim, using_octree=0,
                                     max_distance=3 * dim, rows=(y0, y0 + 1), threads=8, active_lights=n)
        assert hits_match(c, hits[y0], ohits[y0])
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))


POSES = [  # (cam_dir, camera offset from the bench pose in voxels)
    ((1.5708, 0.0), (0.0, 0.0, 0.0)),            # straight along an axis: sin(yaw) == 0 column is never written, many exact ties
    ((3.05, 0.7854), (100.5, -200.25, 1500.0)),  # from high above, looking down: t up to ~2^13
    ((0.35, 2.3), (0.0, 0.0, -60.0)),            # low over the terrain, looking up into the sky
    ((2.0, 4.0), (-1900.0, 3400.0, 700.0)),      # from near the far corner, back across the map
    ((2.2, 1.5708), (0.0, -4000.0, 200.0)),      # camera outside the map (negative y), looking in
]


@pytest.mark.parametrize("pose", range(len(POSES)))
def test_headline_scene_other_cameras(pose):
    """The depth-12 bench scene from other camera poses (the step-loop variants of DESIGN 4 depend on the ray mix:
    long safe runs, rays entering from outside, grazing rays): a fifth of the frame (whole 8-row tile bands) bit-exact
    vs the oracle."""
    sc = _bench_scene(12)
    dim, w, h = sc["dim"], 640, 360
    cam_dir, off = POSES[pose]
    cam_pos = tuple(float(a + b) for a, b in zip(sc["cam_pos"], off))
    c = make_caster(sc["octree"], dim, 0, cam_dir, cam_pos, sc["lights"], sc["atlas"], w, h, 3 * dim)
    assert c.compute(), c.last_error()
    img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
    assert ctr["descriptor_reads"] == int(hits[..., 7].sum())
    for y0 in range(0, h, 40):                        # bands of 8 rows (whole wave tiles), one band in five
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=c._li, atlas=sc["atlas"],
                                     tile_dim=(16, 16), descriptors=sc["octree"].descriptor_buffer,
                                     root_index=sc["octree"].root_index, octree_dim=dim, using_octree=0,
                                     max_distance=3 * dim, rows=(y0, y0 + 8), threads=16)
        assert hits_match(c, hits[y0:y0 + 8], ohits[y0:y0 + 8])
        assert np.array_equal(img[y0:y0 + 8].view(np.uint32), oimg[y0:y0 + 8].view(np.uint32))


def test_depth13_scene_sampled_rows():
    """Beyond the headline: 8192^3 (depth 13), 84 M descriptors (674 MB), thousands of far pointers and page
    headers, 12 stack levels in LDS -- sampled rows bit-exact vs the oracle, counters consistent."""
    sc = _bench_scene(13)
    w, h, dim = 1920, 1080, sc["dim"]
    d = sc["octree"].descriptor_buffer
    far = ((d >> np.uint64(15)) & np.uint64(1)).astype(bool) & (d != np.uint64(0xFFFFFFFFFFFFFFFF))
    assert far.sum() > 1000 and (d == np.uint64(0xFFFFFFFFFFFFFFFF)).sum() > 1000
    c = make_caster(sc["octree"], dim, 0, sc["cam_dir"], sc["cam_pos"], sc["lights"], sc["atlas"], w, h, 3 * dim)
    assert c.compute(), c.last_error()
    img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
    assert ctr["descriptor_reads"] == int(hits[..., 7].sum()) and ctr["primary_rays"] == w * h
    for y0 in range(37, h, 131):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"], lights=c._li,
                                     atlas=sc["atlas"], tile_dim=(16, 16), descriptors=d, root_index=sc["octree"].root_index,
                                     octree_dim=dim, using_octree=0, max_distance=3 * dim, rows=(y0, y0 + 1), threads=8)
        assert hits_match(c, hits[y0], ohits[y0])
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))


def test_depth14_scene_sampled_rows():
    """16384^3 (depth 14): 345 M descriptors (2.8 GB), 13 stack levels in LDS, t values up to 2^15 -- sampled rows
    bit-exact vs the oracle."""
    sc = _bench_scene(14)
    w, h, dim = 1920, 1080, sc["dim"]
    d = sc["octree"].descriptor_buffer
    assert d.size > 300_000_000
    c = make_caster(sc["octree"], dim, 0, sc["cam_dir"], sc["cam_pos"], sc["lights"], sc["atlas"], w, h, 3 * dim)
    assert c.compute(), c.last_error()
    img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
    assert ctr["descriptor_reads"] == int(hits[..., 7].sum()) and ctr["primary_rays"] == w * h
    for y0 in (61, 533, 1002):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"], lights=c._li,
                                     atlas=sc["atlas"], tile_dim=(16, 16), descriptors=d, root_index=sc["octree"].root_index,
                                     octree_dim=dim, using_octree=0, max_distance=3 * dim, rows=(y0, y0 + 1), threads=8)
        assert hits_match(c, hits[y0], ohits[y0])
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))


def test_headline_config_full_size_properties():
    """BASELINE configs[2] at full size (depth-12 SVO, 1920x1080, primary + shadow + shading):
    sampled rows bit-exact vs the oracle, idempotence, tiling invariance, counter identities."""
    sc = _bench_scene(12)
    w, h, dim = 1920, 1080, sc["dim"]
    c = make_caster(sc["octree"], dim, 0, sc["cam_dir"], sc["cam_pos"], sc["lights"], sc["atlas"], w, h, 3 * dim)
    assert c.compute(), c.last_error()
    img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
    # counter identities
    assert ctr["primary_rays"] + ctr["unwritten_pixels"] >= w * h
    assert ctr["shadow_rays"] == int(((hits[..., 5] & 2) != 0).sum())
    assert ctr["steps"] >= int(hits[..., 6].sum())
    assert ctr["descriptor_reads"] == int(hits[..., 7].sum())
    assert ctr["texel_reads"] >= ctr["shadow_rays"]
    # every primary hit is a solid voxel of the scene
    hit = hits[..., 3] == 5
    hv = hits[hit][:, :3]
    hh = sc["height"][hv[:, 1], hv[:, 0]]
    assert (hv[:, 2] <= hh).all()
    # idempotence
    assert c.compute()
    assert np.array_equal(c.read_image().view(np.uint32), img.view(np.uint32)) and np.array_equal(c.read_hits(), hits)
    # sampled rows vs the oracle (bit-exact)
    for y0 in range(4, h, 67):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"], lights=c._li,
                                     atlas=sc["atlas"], tile_dim=(16, 16), descriptors=sc["octree"].descriptor_buffer,
                                     root_index=sc["octree"].root_index, octree_dim=dim, using_octree=0,
                                     max_distance=3 * dim, rows=(y0, y0 + 1), threads=8)
        assert hits_match(c, hits[y0], ohits[y0])
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))
    # tiling invariance at full size
    from voxel_raycaster_amd import tiling
    frames = []
    for r in range(2):
        assert c.set_row_tiling(r, 2, 8) and c.compute()
        frames.append(c.read_image())
    assert np.array_equal(frames[1].view(np.uint32), img.view(np.uint32))   # same buffer: both halves rewritten


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_random_scenes(seed, atlas):
    """Seeded random scenes (tests/fuzz_scenes.py): dims 4..32, materials {0,1,5,6}, cameras inside/outside/on the grid,
    random lights, ragged resolutions, random step caps; array kernel, SVO kernel (+attachments), the exact jumps and mode B
    vs the oracle.  tests/soak_fuzz_gpu.py runs the same cases for as long as one likes."""
    import fuzz_scenes
    fuzz_scenes.run_case(1000 + seed, atlas)


@pytest.mark.parametrize("mode", ["array", "svo", "svo_jump", "svo_primary_only"])
@pytest.mark.parametrize("n", [2, 4, 8])
@pytest.mark.parametrize("make", [scenes.floor_pillars, scenes.random_sparse, scenes.open_sky, scenes.mirror_wall,
                                  scenes.app_default], ids=lambda f: f.__name__)
def test_multi_light_equals_oracle(make, n, mode, atlas):
    """Multi-light extension (setting light_count): every finished shadow ray goes back to the first
    strike for the next light; array kernel, SVO kernel (lane mode kRelight), with jumps and without shadow rays."""
    s = scenes.with_lights(make(), n)
    dim, w, h = s["dim"], 160, 120
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    md = 20 if dim <= 16 else 3 * dim
    using_octree = 1 if mode == "array" else 0
    sr = 0 if mode == "svo_primary_only" else 1
    c = make_caster(o, dim, using_octree, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md, grid=s["grid"],
                    shadow_rays=sr, light_count=n)
    if mode == "svo_jump":
        assert c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 2)
    assert c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                    octree_dim=dim, using_octree=using_octree, grid=s["grid"], max_distance=md,
                                    shadow_rays=sr, active_lights=n)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    if sr and make is not scenes.mirror_wall and make is not scenes.app_default:
        assert octr["shadow_rays"] > octr["primary_rays"] * 0.5


def test_multi_light_is_opt_in_and_follows_the_live_count(atlas):
    """Default: the reference's behaviour -- light_count is bound but only light 0 shades (ray_caster_kernel.cl:264).
    With the setting, the lights used are min(setting, *light_count), re-read on every compute."""
    s = scenes.with_lights(scenes.floor_pillars(), 4)
    dim, w, h, md = s["dim"], 96, 64, 96
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)

    def oracle(n):
        return orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas,
                           tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index, octree_dim=dim,
                           using_octree=0, max_distance=md, active_lights=n)

    c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
    assert c.compute()
    assert_same(c.read_image(), c.read_hits(), c.counters(), *oracle(1))
    assert c.add_to_settings_buffer("light_count", "LIGHT_COUNT", 8)
    count = np.array([3], dtype=np.int32)
    assert c.assign_lights(c._li, count) and c.validate(), c.last_error()
    assert c.compute(), c.last_error()
    assert_same(c.read_image(), c.read_hits(), c.counters(), *oracle(3))
    count[0] = 2                                      # live buffer: no re-assign
    assert c.compute()
    assert_same(c.read_image(), c.read_hits(), c.counters(), *oracle(2))
    count[0] = 0                                      # never fewer than light 0
    assert c.compute()
    assert_same(c.read_image(), c.read_hits(), c.counters(), *oracle(1))


def test_bench_two_rank_rehearsal():
    """bench.py's N>1 path (supersampled ray table, interleaved row tiling, SUM/MAX reductions, one JSON line)
    with two ranks sharing GPU 0 over gloo -- the driver's real multi-GPU launch uses RCCL on 2/4/8 GPUs."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, VRC_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--depth", "8", "--width", "320", "--height", "240"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["rays_per_step"] > 2 * 320 * 240          # both ranks' rays are counted
    assert rec["roofline"]["bound"] == "hbm" and "cpu_baseline" not in rec
