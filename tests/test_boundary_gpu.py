"""The drop-in boundary (-m gpu) -- SURVEY 8 row b (class CLCaster, include/CLCaster.h:93-329): live settings re-checked by
every compute (overwrite_setting, CLCaster.cpp:1087-1109), the optional hit records, caller buffers, viewports and atlases of
shapes other than the application's (CLCaster.cpp:208-222,233-299).  tests/test_prepare_gpu.py covers validate / prepare and
threads, tests/test_parity_gpu.py the call sequence of Application::init_clcaster."""
import functools
import os
import resource
import time

import numpy as np
import pytest

import scenes
import voxel_raycaster_amd as vrc
from oracle import orc
from gpu_helpers import _but_reads, _frame, _peak_rss_kb, _reset_peak_rss, _rss_now_kb, bench_scene, configure, lights4, survey_camera
from test_parity_gpu import assert_same, hits_match, make_caster

pytestmark = pytest.mark.gpu

def test_live_settings_are_rechecked_by_every_compute(atlas):
    """Settings stay live after validate(); a structural setting changed to nonsense makes compute() return an error
    code, never a device fault."""
    s = scenes.floor_pillars()
    dim, w, h = s["dim"], 96, 64
    m = vrc.Map(dim, s["grid"])
    li = np.zeros((8, 10), dtype=np.float32)
    li[:1] = s["lights"]
    c = vrc.CLCaster()
    assert c.init(0) and c.assign_octree(m.octree)
    configure(c, dim, atlas, s["cam_dir"], s["cam_pos"], li, w, h)
    assert c.validate() and c.compute()
    good = c.read_image()
    for name, value, word in [("octree_dimensions", 3, "power of two"), ("octree_dimensions", 1 << 30, "power of two"),
                              ("octree_root_index", 10 ** 9, "out of range"), ("using_octree", 1, "dense map"),
                              ("stepping_mode", 7, "stepping_mode"), ("max_distance", 1 << 31, "max_distance")]:
        old = c.get_setting(name)
        if old is None:
            assert c.add_to_settings_buffer(name, name.upper(), value)
            old = 0
        else:
            assert c.overwrite_setting(name, value)
        assert c.compute() is False and c.last_status in (1, 2) and word in c.last_error(), (name, c.last_error())
        assert c.overwrite_setting(name, old) and c.compute()
    assert np.array_equal(c.read_image().view(np.uint32), good.view(np.uint32))


def test_hit_records_can_be_switched_off(atlas):
    s = scenes.floor_pillars()
    dim, w, h = s["dim"], 128, 96
    m = vrc.Map(dim, s["grid"])
    li = np.zeros((8, 10), dtype=np.float32)
    li[:1] = s["lights"]
    c = vrc.CLCaster()
    assert c.init(0) and c.assign_octree(m.octree)
    configure(c, dim, atlas, s["cam_dir"], s["cam_pos"], li, w, h)
    assert c.add_to_settings_buffer("hit_records", "HIT_RECORDS", 0) and c.validate() and c.compute()
    assert c.memory_usage()["hit_bytes"] == 0
    with pytest.raises(vrc.VrcError):
        c.read_hits()
    img = c.read_image()
    assert c.overwrite_setting("hit_records", 1) and c.compute()
    assert np.array_equal(c.read_image().view(np.uint32), img.view(np.uint32)) and c.read_hits()[..., 3].max() == 5


def test_hit_records_of_an_older_frame_are_not_handed_out(atlas):
    """hit_records switched 1 -> 0: the buffer still holds the previous frame's records; read_hits must say NOT_READY
    instead of returning them beside the new image (advisor finding, round 2)."""
    s = scenes.floor_pillars()
    dim, w, h = s["dim"], 128, 96
    m = vrc.Map(dim, s["grid"])
    li = np.zeros((8, 10), dtype=np.float32)
    li[:1] = s["lights"]
    c = vrc.CLCaster()
    assert c.init(0) and c.assign_octree(m.octree)
    configure(c, dim, atlas, s["cam_dir"], s["cam_pos"], li, w, h)
    assert c.validate() and c.compute()
    hits = c.read_hits()
    c._keep["cam"][1][0] += 1.5                                      # the next frame differs
    assert c.add_to_settings_buffer("hit_records", "HIT_RECORDS", 0) and c.compute()
    with pytest.raises(vrc.VrcError):
        c.read_hits()
    assert c.last_status == 2 and "hit_records is 0" in c.last_error()
    assert c.overwrite_setting("hit_records", 1) and c.compute()
    assert not np.array_equal(c.read_hits(), hits)


def test_caller_supplied_destinations_are_checked(atlas):
    """read_*(out=...) writes through a raw pointer: a wrong dtype, shape or a strided view must be refused."""
    s = scenes.floor_pillars()
    dim, w, h = s["dim"], 64, 48
    m = vrc.Map(dim, s["grid"])
    li = np.zeros((8, 10), dtype=np.float32)
    li[:1] = s["lights"]
    c = vrc.CLCaster()
    assert c.init(0) and c.assign_octree(m.octree)
    configure(c, dim, atlas, s["cam_dir"], s["cam_pos"], li, w, h)
    assert c.validate() and c.compute()
    for bad in (np.zeros((h, w, 4), np.uint8), np.zeros((h, w, 3), np.float32), np.zeros((h, 2 * w, 4), np.float32)[:, ::2],
                np.zeros((w, h, 4), np.float32)):
        with pytest.raises(vrc.VrcError):
            c.read_image(out=bad)
    with pytest.raises(vrc.VrcError):
        c.read_image_rgba8(out=np.zeros((h, w, 4), np.float32))
    with pytest.raises(vrc.VrcError):
        c.read_hits(out=np.zeros((h, w, 8), np.int64))
    ok = np.zeros((h, w, 4), np.float32)
    assert c.read_image(out=ok) is ok and ok.any()


TINY_VIEWPORTS = [(1, 1), (1, 9), (9, 1), (3, 5), (8, 8), (65, 9)]


@pytest.mark.parametrize("path", ["array", "svo_exact", "svo_mode_b"])
@pytest.mark.parametrize("res", TINY_VIEWPORTS, ids=[f"{w}x{h}" for w, h in TINY_VIEWPORTS])
def test_tiny_and_ragged_viewports(res, path, atlas):
    """Viewports smaller than one 8x8 wave tile / one 32x8 block and not multiples of either: every pixel of the frame
    equals the oracle in all three stepping paths, and the same frame comes out of a 3-rank group of row slices."""
    from test_parity_gpu import make_caster, assert_same
    s = scenes.floor_pillars()
    dim, (w, h) = s["dim"], res
    m = vrc.Map(dim, s["grid"])
    using_octree = 1 if path == "array" else 0
    mode = 1 if path == "svo_mode_b" else 0
    c = make_caster(m.octree, dim, using_octree, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, 3 * dim, grid=s["grid"])
    assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", mode) and c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=m.octree.descriptor_buffer, root_index=m.octree.root_index,
                                    octree_dim=dim, using_octree=using_octree, grid=s["grid"], max_distance=3 * dim,
                                    stepping_mode=mode)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    assert np.array_equal(c.read_image_rgba8(), orc.image_to_rgba8(oimg))
    g = vrc.CLCaster()
    assert g.init_group([0, 0, 0], band_rows=8)
    assert g.assign_octree(m.octree) and g.assign_map(s["grid"], (dim, dim, dim))
    li = np.zeros((8, 10), dtype=np.float32)
    li[:1] = s["lights"]
    configure(g, dim, atlas, s["cam_dir"], s["cam_pos"], li, w, h)
    assert g.overwrite_setting("using_octree", using_octree) and g.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", mode)
    assert g.validate() and g.compute(), g.last_error()
    assert hits_match(g, g.read_hits(), ohits) and np.array_equal(g.read_image().view(np.uint32), oimg.view(np.uint32))


ATLAS_SHAPES = [(192, 128, (24, 8)), (64, 64, (16, 16)), (300, 200, (7, 9)), (16, 16, (16, 16)), (8, 8, (16, 16)), (1, 1, (1, 1))]


@pytest.mark.parametrize("using_octree", [1, 0], ids=["array", "svo"])
@pytest.mark.parametrize("make", [scenes.floor_pillars, scenes.mirror_wall], ids=["floor_pillars", "mirror_wall"])
@pytest.mark.parametrize("shape", ATLAS_SHAPES, ids=[f"{w}x{h}-tile{t[0]}x{t[1]}" for w, h, t in ATLAS_SHAPES])
def test_atlas_and_tile_shapes_other_than_the_apps(shape, make, using_octree):
    """create_texture_atlas (src/CLCaster.cpp:208-222) takes any texture and tile size; the kernel's texel arithmetic
    (:652-656, :684-688: uv * (atlas_dim / tile_dim) + tile * (atlas_dim / tile_dim), integer division) only lands
    inside tile (5,0) / (3,4) for the app's 256 / 16.  Whatever it lands on -- other tiles, the clamp at the atlas edge
    -- must be the same texel in the HIP path and the oracle: the material-5 tile colours the frame
    (floor_pillars), the mirror tile is fetched twice per pixel and only counted (mirror_wall: a reflected ray starts
    inside the mirror voxel and strikes it again, :700-702, so the reference's mirrors come out black)."""
    from test_parity_gpu import assert_same
    aw, ah, tile = shape
    atlas = vrc.synthetic_atlas(aw, ah)
    s = make()
    dim, w, h = s["dim"], 160, 120
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000).attach_materials_from_grid(s["grid"])
    c = vrc.CLCaster()
    assert c.init(0) and c.assign_octree(o) and c.assign_map(s["grid"], (dim, dim, dim))
    li = np.zeros((8, 10), dtype=np.float32)
    li[:1] = s["lights"]
    assert c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim) and c.add_to_settings_buffer("using_octree", "OCTENABLED", using_octree)
    assert c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", 3 * dim)
    cd, cp = np.array(s["cam_dir"], dtype=np.float32), np.array(s["cam_pos"], dtype=np.float32)
    assert c.assign_camera(cd, cp) and c.create_viewport(w, h) and c.assign_lights(li)
    assert c.create_texture_atlas(atlas, tile)
    if tile[0] > aw or tile[1] > ah:                     # atlas_dim / tile_dim == 0: refused by validate(), not rendered
        assert not c.validate() and "tile larger than atlas" in c.last_error()
        return
    assert c.validate() and c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=li, atlas=atlas,
                                    tile_dim=tile, descriptors=o.descriptor_buffer, root_index=o.root_index,
                                    octree_dim=dim, using_octree=using_octree, grid=s["grid"], max_distance=3 * dim,
                                    attachment_lookup=o.attachment_lookup, attachments=o.attachment_buffer)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    assert octr["n_tex"] > 0 and (ohits[..., 3] == (6 if make is scenes.mirror_wall else 5)).any()
