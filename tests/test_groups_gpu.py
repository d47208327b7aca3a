"""Multi-GPU (-m gpu) -- SURVEY 8 row e: row slices per rank, the single-process group handle (one synchronous compute() over n
ranks, CLCaster.cpp:224-228,946-987), the cross-device copy path rehearsed on one GPU (VRC_GROUP_OWN_COPIES), one tree shared by
the handles that render it (vrc_tree, vrc_assign_octree_from).  No collective anywhere: the SVO is replicated, tiles are copied out
by the GPU that rendered them."""
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

def test_row_slices_hold_only_their_rows_and_assemble_the_frame(atlas):
    s = scenes.random_sparse()
    dim, w, h = s["dim"], 200, 123                                  # ragged: last band partial, last tile row partial
    m = vrc.Map(dim, s["grid"])
    li = np.zeros((8, 10), dtype=np.float32)
    li[:1] = s["lights"]
    full = vrc.CLCaster()
    assert full.init(0) and full.assign_octree(m.octree)
    configure(full, dim, atlas, s["cam_dir"], s["cam_pos"], li, w, h)
    assert full.validate() and full.compute()
    ref_img, ref_hits, ref_rgba, ref_ctr = full.read_image(), full.read_hits(), full.read_image_rgba8(), full.counters()
    for world, band in [(2, 8), (3, 16), (4, 8)]:
        img = np.full((h, w, 4), -7.0, dtype=np.float32)
        hits = np.full((h, w, 8), -7, dtype=np.int32)
        rgba = np.full((h, w, 4), 77, dtype=np.uint8)
        rays, rows_seen = 0, 0
        for r in range(world):
            c = vrc.CLCaster()
            assert c.init(0) and c.set_row_slice(r, world, band) and c.assign_octree(m.octree)
            configure(c, dim, atlas, s["cam_dir"], s["cam_pos"], li, w, h)
            assert c.set_row_slice(r, world, band) is False          # the buffers are already sized
            assert c.validate() and c.compute(), c.last_error()
            mem = c.memory_usage()
            from voxel_raycaster_amd import tiling
            mine = tiling.rows_of_rank(h, r, world, band)
            assert mem["rows"] == len(mine) and mem["viewport_bytes"] == 16 * w * len(mine) == mem["image_bytes"]
            assert mem["hit_bytes"] == 32 * w * len(mine)
            c.read_image(img); c.read_hits(hits); c.read_image_rgba8(rgba)
            rays += c.counters()["primary_rays"]
            rows_seen += len(mine)
        assert rows_seen == h and rays == ref_ctr["primary_rays"]
        assert np.array_equal(img.view(np.uint32), ref_img.view(np.uint32)) and np.array_equal(hits, ref_hits)
        assert np.array_equal(rgba, ref_rgba)


@pytest.mark.parametrize("ranks", [2, 4])
def test_group_handle_is_one_synchronous_compute_over_all_ranks(ranks, atlas):
    """vrc_create_group with every rank on GPU 0 (all this box has): one compute() renders all row slices, the
    read-back gathers them, counters are summed, each rank holds 1/n of the frame buffers and shares rank 0's tree."""
    sc = bench_scene(10)
    dim, w, h = sc["dim"], 1280, 720
    one = vrc.CLCaster()
    assert one.init(0) and one.assign_octree(sc["octree"])
    configure(one, dim, sc["atlas"], sc["cam_dir"], sc["cam_pos"], sc["lights"], w, h)
    assert one.validate() and one.compute()
    g = vrc.CLCaster()
    assert g.init_group([0] * ranks, band_rows=8) and g.group_size() == ranks
    assert g.assign_octree(sc["octree"])
    configure(g, dim, sc["atlas"], sc["cam_dir"], sc["cam_pos"], sc["lights"], w, h)
    assert g.set_row_tiling(0, 1, 8) is False                       # a group's tiling is fixed
    assert g.validate() and g.compute(), g.last_error()
    assert np.array_equal(g.read_image().view(np.uint32), one.read_image().view(np.uint32))
    assert np.array_equal(g.read_hits(), one.read_hits()) and np.array_equal(g.read_image_rgba8(), one.read_image_rgba8())
    assert g.counters() == one.counters()
    rows = [g.memory_usage(r) for r in range(ranks)]
    assert sum(m["rows"] for m in rows) == h and all(m["image_bytes"] == 16 * w * m["rows"] for m in rows)
    assert rows[0]["octree_shared"] == 0 and all(m["octree_shared"] == 1 for m in rows[1:])
    # live settings and live camera reach every rank
    cam = g._keep["cam"][1]
    cam[2] += 3.0
    one._keep["cam"][1][2] += 3.0
    assert g.overwrite_setting("shadow_rays", 0) and one.overwrite_setting("shadow_rays", 0)
    assert g.compute() and one.compute()
    assert np.array_equal(g.read_image().view(np.uint32), one.read_image().view(np.uint32))
    # a device-built tree fans out too
    info, _ = g.build_shell_terrain(10, 1, 2, 2)
    assert g.validate() and g.compute(), g.last_error()
    assert g.octree_size()[0] == info["n_descriptors"] and g.counters()["primary_rays"] == w * h


def test_group_with_more_ranks_than_row_bands(atlas):
    """A 20-row frame on 4 ranks in bands of 8: ranks 0-2 own 8, 8 and 4 rows, rank 3 owns none -- still one frame."""
    s = scenes.floor_pillars()
    dim, w, h = s["dim"], 96, 20
    m = vrc.Map(dim, s["grid"])
    li = np.zeros((8, 10), dtype=np.float32)
    li[:1] = s["lights"]
    one = vrc.CLCaster()
    assert one.init(0) and one.assign_octree(m.octree)
    configure(one, dim, atlas, s["cam_dir"], s["cam_pos"], li, w, h)
    assert one.validate() and one.compute()
    g = vrc.CLCaster()
    assert g.init_group([0, 0, 0, 0]) and g.assign_octree(m.octree)
    configure(g, dim, atlas, s["cam_dir"], s["cam_pos"], li, w, h)
    assert g.validate() and g.compute(), g.last_error()
    assert [g.memory_usage(r)["rows"] for r in range(4)] == [8, 8, 4, 0]
    assert np.array_equal(g.read_image().view(np.uint32), one.read_image().view(np.uint32))
    assert np.array_equal(g.read_hits(), one.read_hits()) and g.counters() == one.counters()


def test_group_with_own_copies_takes_the_cross_device_path(atlas):
    """VRC_GROUP_OWN_COPIES: every rank sits on GPU 0 (all this box has) but takes the path of a rank on another GPU --
    own allocation + hipMemcpyPeerAsync of the tree, attachment re-copy, release -- on the depth-12 tree with
    attachments, 4 ranks.  Frame, hit records, RGBA8 and counters equal the single handle's; no rank shares rank 0's
    arrays.  (A same-GPU rehearsal: no 8-GPU hardware run exists, and no scaling number is claimed.)"""
    sc = bench_scene(12)
    dim, w, h = sc["dim"], 1280, 720
    # (a fresh Octree object over the cached scene's arrays: the materials must not follow the scene into other tests)
    tree = vrc.Octree(sc["octree"].descriptor_buffer, sc["octree"].root_index, dim).attach_materials_procedural(12, seed=1, mirror_period=64)
    one = vrc.CLCaster()
    assert one.init(0) and one.assign_octree(tree)
    configure(one, dim, sc["atlas"], sc["cam_dir"], sc["cam_pos"], sc["lights"], w, h)
    assert one.validate() and one.compute()
    ref_img, ref_hits, ref_rgba, ref_ctr = one.read_image(), one.read_hits(), one.read_image_rgba8(), one.counters()
    assert (ref_hits[..., 3] == 6).sum() > 100                       # the mirrors of the attachments are in the picture
    g = vrc.CLCaster()
    assert g.init_group([0] * 4, band_rows=8, own_copies=True) and g.group_size() == 4
    assert g.last_error() == ""                                      # no peer-access fallback to report on one GPU
    assert g.assign_octree(tree)                                     # tree, then attachments: fan-out + attachment re-fan-out
    configure(g, dim, sc["atlas"], sc["cam_dir"], sc["cam_pos"], sc["lights"], w, h)
    assert g.validate() and g.compute(), g.last_error()
    mem = [g.memory_usage(r) for r in range(4)]
    n_desc = tree.descriptor_buffer.size
    per_rank = n_desc * 8 + n_desc * 4 + tree.attachment_buffer.size * 8
    assert all(m["octree_shared"] == 0 and m["octree_bytes"] == per_rank for m in mem), mem
    assert all(m["peer_access"] == -1 for m in mem)
    # pageable destination: every rank stages its tile through its own pinned buffer; pinned destination: direct
    img = g.read_image()
    assert np.array_equal(img.view(np.uint32), ref_img.view(np.uint32))
    pinned = np.zeros_like(ref_img)
    vrc.pin_host_buffer(pinned)
    try:
        g.read_image(out=pinned)
        assert np.array_equal(pinned.view(np.uint32), ref_img.view(np.uint32))
    finally:
        vrc.unpin_host_buffer(pinned)
    assert np.array_equal(g.read_hits(), ref_hits) and np.array_equal(g.read_image_rgba8(), ref_rgba) and g.counters() == ref_ctr
    # a rejected attachment call leaves EVERY rank as it was (the check comes before anything is released)
    import ctypes as C
    bad = tree.attachment_lookup.copy()
    bad[7] = tree.attachment_buffer.size + 1
    rc = vrc.lib.vrc_assign_octree_attachments(g._h, bad.ctypes.data_as(C.POINTER(C.c_uint32)), bad.size,
                                               tree.attachment_buffer.ctypes.data_as(C.POINTER(C.c_uint64)), tree.attachment_buffer.size)
    assert rc == 1 and "past the attachment buffer" in g.last_error()
    assert [g.memory_usage(r)["octree_bytes"] for r in range(4)] == [per_rank] * 4
    assert g.validate() and g.compute() and np.array_equal(g.read_hits(), ref_hits)
    # dropping the attachments reaches every rank too: all of them render material 5 only
    rc = vrc.lib.vrc_assign_octree_attachments(g._h, None, 0, None, 0)
    assert rc == 0 and [g.memory_usage(r)["octree_bytes"] for r in range(4)] == [n_desc * 8] * 4
    assert g.validate() and g.compute()
    mats = g.read_hits()[..., 3]
    assert set(np.unique(mats).tolist()) <= {0, 5}
    # a new tree releases the ranks' own copies and fans out again (device-built this time)
    info, _ = g.build_shell_terrain(10, 1, 2, 2)
    assert all(g.memory_usage(r)["octree_bytes"] == info["n_descriptors"] * 8 for r in range(4))


def test_trees_are_shared_not_copied():
    """the coarse table and the boxes are functions of the TREE.  A second caster that adopts the first one's
    tree (vrc_assign_octree_from) and the ranks of a same-GPU group hold ONE descriptor array, ONE table, ONE set of boxes;
    frames are those of a caster with its own upload; the arrays outlive the handle that uploaded them."""
    import bench
    sc = bench.build_scene(10)
    w, h = 640, 360
    a = bench.make_caster(sc, w, h, 0)
    ref = _frame(a)
    ma = a.memory_usage2()
    assert ma["tree_holders"] == 1 and ma["coarse_bytes"] > 0 and ma["box_bytes"] > 0 and ma["octree_shared"] == 0
    b = vrc.CLCaster()
    assert b.init(0)
    for name, v in (("octree_dimensions", sc["dim"]), ("using_octree", 0), ("max_distance", 3 * sc["dim"])):
        assert b.add_to_settings_buffer(name, name.upper(), v)
    assert b.assign_octree_from(a), b.last_error()
    assert (b.assign_camera(sc["cam_dir"], sc["cam_pos"]) and b.create_viewport(w, h) and b.assign_lights(sc["lights"])
            and b.create_texture_atlas(sc["atlas"], (16, 16)) and b.validate()), b.last_error()
    got = _frame(b)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and got[2] == ref[2]
    mb = b.memory_usage2()
    assert mb["tree_holders"] == 2 and mb["octree_shared"] == 1 and mb["coarse_bytes"] == ma["coarse_bytes"] and mb["box_bytes"] == ma["box_bytes"]
    assert mb["box_build_seconds"] == ma["box_build_seconds"]        # not built a second time
    del a                                                           # the uploader goes away: the tree stays with its last holder
    import gc
    gc.collect()
    got = _frame(b)
    assert np.array_equal(got[0], ref[0]) and b.memory_usage2()["tree_holders"] == 1
    # an 8-rank group on one GPU: one tree between the ranks
    g = vrc.CLCaster()
    assert g.init_group([0] * 8, band_rows=8) and g.assign_octree(sc["octree"])
    for name, v in (("octree_dimensions", sc["dim"]), ("using_octree", 0), ("max_distance", 3 * sc["dim"])):
        assert g.add_to_settings_buffer(name, name.upper(), v)
    assert (g.assign_camera(sc["cam_dir"], sc["cam_pos"]) and g.create_viewport(w, h) and g.assign_lights(sc["lights"])
            and g.create_texture_atlas(sc["atlas"], (16, 16)) and g.validate()), g.last_error()
    got = _frame(g)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and got[2] == ref[2]
    mem = [g.memory_usage2(r) for r in range(8)]
    assert all(m["tree_holders"] == 8 for m in mem) and [m["octree_shared"] for m in mem] == [0] + [1] * 7
    assert len({m["box_build_seconds"] for m in mem}) == 1


def test_shared_tree_materials_follow_either_handle(atlas):
    """Materials belong to the TREE (vrc_assign_octree_from's contract): assigned through one holder they are rendered by the
    other from its next frame on -- mirrors and pass-through voxels appear -- and both frames equal the oracle's."""
    from test_oracle_cpu import _with_pass_through
    s = _with_pass_through(scenes.mirror_wall())
    dim, w, h = s["dim"], 128, 96
    plain = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    with_mat = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000).attach_materials_from_grid(s["grid"])
    a = make_caster(plain, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, 3 * dim)
    b = make_caster(plain, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, 3 * dim, tree_from=a)
    assert b.memory_usage2()["tree_holders"] == 2

    def oracle(tree):
        return orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=a._li, atlas=atlas, tile_dim=(16, 16),
                           descriptors=tree.descriptor_buffer, root_index=tree.root_index, octree_dim=dim, using_octree=0, max_distance=3 * dim,
                           attachment_lookup=tree.attachment_lookup, attachments=tree.attachment_buffer)

    assert b.compute(), b.last_error()
    assert_same(b.read_image(), b.read_hits(), b.counters(), *oracle(plain))
    assert a.assign_octree_attachments(with_mat) and a.validate() and b.validate()
    for c in (b, a):
        assert c.compute(), c.last_error()
        oimg, ohits, octr = oracle(with_mat)
        assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    assert (ohits[..., 3] == 6).sum() > 0
    # ... and taken away again through the OTHER holder
    assert vrc.lib.vrc_assign_octree_attachments(b._h, None, 0, None, 0) == 0 and a.validate() and b.validate()
    assert a.compute(), a.last_error()
    assert_same(a.read_image(), a.read_hits(), a.counters(), *oracle(plain))


def test_read_back_into_a_pinned_caller_buffer(atlas):
    """vrc_pin_host_buffer: the draw() replacement may page-lock its frame buffer once; the read-back calls then copy
    straight into it -- same bytes as the pageable path (tools/readback_rate.py times the two)."""
    s = scenes.floor_pillars()
    dim, w, h = s["dim"], 160, 120
    m = vrc.Map(dim, s["grid"])
    li = np.zeros((8, 10), dtype=np.float32)
    li[:1] = s["lights"]
    c = vrc.CLCaster()
    assert c.init(0) and c.assign_octree(m.octree)
    configure(c, dim, atlas, s["cam_dir"], s["cam_pos"], li, w, h)
    assert c.validate() and c.compute()
    img, rgba = c.read_image(), c.read_image_rgba8()
    pin_img, pin_rgba = np.zeros_like(img), np.zeros_like(rgba)
    vrc.pin_host_buffer(pin_img)
    vrc.pin_host_buffer(pin_rgba)
    try:
        c.read_image(out=pin_img)
        c.read_image_rgba8(out=pin_rgba)
        assert np.array_equal(pin_img.view(np.uint32), img.view(np.uint32)) and np.array_equal(pin_rgba, rgba)
        assert np.array_equal(pin_rgba, orc.image_to_rgba8(img))
    finally:
        vrc.unpin_host_buffer(pin_img)
        vrc.unpin_host_buffer(pin_rgba)
    with pytest.raises(vrc.VrcError):
        vrc.unpin_host_buffer(pin_img)               # not pinned any more
