"""The empty boxes of the exact kernel (-m gpu; csrc/empty_boxes.hip) -- SURVEY 8 row a5: a derived structure like the coarse
table.  Occupancy still comes from the descriptor array alone (Octree.h:89-94); inside a box the step loop of
ray_caster_kernel.cl:555-570 reads no occupancy, so the float recurrence (:559), the iteration count (:714) and with them every
pixel and every hit record are what they were.  Only the descriptor-read COUNT is the box traversal's own (field 7 of the hit
records, counter descriptor_reads)."""
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

@pytest.mark.parametrize("depth,lights,w,h", [(8, 1, 640, 360), (10, 2, 640, 360), (11, 1, 960, 540), (12, 1, 1920, 1080), (12, 4, 960, 540)],
                         ids=["d8", "d10-2lights", "d11", "d12-headline", "d12-4lights"])
def test_empty_boxes_never_change_the_frame(depth, lights, w, h):
    """Boxes on / off, closed-form jumps on / off / forced from 2 iterations, Euclid tables in LDS or global: one image, one set
    of hit records (the read count aside), the same counters; the boxes' self-check finds no solid voxel inside a box."""
    import bench
    sc = bench.build_scene(depth)
    c = bench.make_caster(sc, w, h, 0, light_count=lights)
    assert c.add_to_settings_buffer("empty_boxes", "EMPTY_BOXES", 0) and c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 1 << 24)
    assert c.add_to_settings_buffer("jump_tables_lds", "JUMP_TABLES_LDS", 2)
    ref = _frame(c)
    assert not c.used_empty_boxes() and ref[2]["canonical_reads"]
    for boxes in (1, -1):
        for jmr, lds in ((1 << 24, 2), (16, 2), (2, 0)):
            assert c.overwrite_setting("empty_boxes", boxes) and c.overwrite_setting("jump_min_run", jmr) and c.overwrite_setting("jump_tables_lds", lds)
            img, hits, ctr = _frame(c)
            tag = f"empty_boxes={boxes} jump_min_run={jmr} jump_tables_lds={lds}"
            assert c.used_empty_boxes() and not ctr["canonical_reads"], tag
            assert np.array_equal(img, ref[0]), f"{tag}: {int((img != ref[0]).any(-1).sum())} pixels differ"
            assert np.array_equal(hits[..., :7], ref[1][..., :7]), f"{tag}: {int((hits[..., :7] != ref[1][..., :7]).any(-1).sum())} hit records differ"
            assert _but_reads(ctr) == _but_reads(ref[2]), tag
            assert ctr["descriptor_reads"] == int(hits[..., 7].astype(np.int64).sum())
    chk = c.empty_boxes_check(1 << 22, seed=depth)
    assert chk["boxes_sampled"] > 100000 and chk["solid_voxels"] == 0
    m = c.memory_usage2()
    assert m["empty_boxes"] == 1 and m["box_bytes"] > 32 * sc["octree"].descriptor_buffer.size and m["note"] == ""
    # a word per descriptor: every level has records; no region query of the build ran into its budget (the counter exists)
    assert m["box_levels"] == depth and m["box_records"] == sc["octree"].descriptor_buffer.size and m["box_queries_cut"] == 0
    print(f"\ndepth {depth}: boxes built in {chk['build_seconds'] * 1e3:.0f} ms, {m['box_bytes'] / 1e6:.0f} MB; descriptor reads "
          f"{ref[2]['descriptor_reads'] / 1e6:.2f} M canonical, {ctr['descriptor_reads'] / 1e6:.2f} M with the boxes")
    # back to the canonical traversal on the same handle: the canonical count again
    assert c.overwrite_setting("empty_boxes", 0)
    img, hits, ctr = _frame(c)
    assert np.array_equal(hits, ref[1]) and {k: v for k, v in ctr.items()} == ref[2]


@pytest.mark.parametrize("make", [scenes.floor_pillars, scenes.random_sparse, scenes.mirror_wall, scenes.open_sky],
                         ids=["floor_pillars", "random_sparse", "mirror_wall", "open_sky"])
def test_empty_boxes_on_dense_grid_trees_with_materials(make, atlas):
    """Trees of Octree::Generate's own layout (100000-entry buffer, Octree.cpp:13-43) with attachments -- mirrors and
    pass-through materials, rays inside solid, rays restarted by the hit block -- rendered with the boxes against the oracle."""
    from test_oracle_cpu import _with_pass_through
    s = _with_pass_through(make())
    dim, w, h = s["dim"], 160, 120
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000).attach_materials_from_grid(s["grid"])
    c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, 3 * dim, empty_boxes=1)
    assert c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 2)
    assert c.compute(), c.last_error()
    assert c.used_empty_boxes() == (dim >= 32)
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index, octree_dim=dim,
                                    using_octree=0, max_distance=3 * dim, attachment_lookup=o.attachment_lookup,
                                    attachments=o.attachment_buffer)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    if c.used_empty_boxes():
        assert c.empty_boxes_check(1 << 18)["solid_voxels"] == 0


def test_empty_boxes_exhaustive_check_on_a_small_tree(atlas):
    """Every voxel of every box of a 64^3 tree against the dense grid (the device self-check samples; this one enumerates):
    the boxes are read back through the frame they render -- a ray from every empty voxel along every axis direction steps
    exactly as far as the oracle's ray does -- and through the sampled self-check with more samples than the tree has boxes."""
    s = scenes.random_sparse()
    dim = s["dim"]
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, 64, 48, 3 * dim, empty_boxes=1)
    assert c.compute(), c.last_error()
    n_boxes = 8 * o.descriptor_buffer.size
    chk = c.empty_boxes_check(64 * n_boxes, seed=3)               # ~64 samples per (descriptor, child) pair
    assert chk["solid_voxels"] == 0 and chk["boxes_sampled"] > n_boxes
    # cameras all over the map, looking along and across the axes: frames equal the oracle's
    rng = np.random.default_rng(11)
    empty = np.argwhere(np.asarray(s["grid"]).reshape(dim, dim, dim) == 0)      # [z, y, x]
    for k in range(12):
        z, y, x = empty[rng.integers(len(empty))]
        cam_pos = np.array([x + 0.31, y + 0.47, z + 0.59], dtype=np.float32)
        cam_dir = np.array([rng.uniform(0.3, 2.8), rng.uniform(0.0, 6.2)], dtype=np.float32)
        cc = make_caster(o, dim, 0, cam_dir, cam_pos, s["lights"], atlas, 64, 48, 3 * dim, empty_boxes=1)
        assert cc.compute(), cc.last_error()
        oimg, ohits, octr = orc.raycast(width=64, height=48, cam_dir=cam_dir, cam_pos=cam_pos, lights=cc._li, atlas=atlas, tile_dim=(16, 16),
                                        descriptors=o.descriptor_buffer, root_index=o.root_index, octree_dim=dim, using_octree=0,
                                        max_distance=3 * dim)
        assert_same(cc.read_image(), cc.read_hits(), cc.counters(), oimg, ohits, octr)


def test_optional_structures_fail_soft(atlas):
    """the table and the boxes are accelerations, not requirements.  A tree in a map too large for the default
    table gets a coarser one; with the table switched off there are no boxes and the frame is the same."""
    s = scenes.floor_pillars(32)
    dim = s["dim"]
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, 96, 64, 3 * dim)
    ref = _frame(c)
    assert c.used_empty_boxes()
    assert c.add_to_settings_buffer("coarse_log2", "COARSE_LOG2", 0)
    got = _frame(c)
    assert not c.used_empty_boxes() and c.memory_usage2()["coarse_bytes"] == 0
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1][..., :7], ref[1][..., :7])
    assert c.overwrite_setting("coarse_log2", -1)
    got = _frame(c)
    assert c.used_empty_boxes() and np.array_equal(got[1], ref[1]) and got[2] == ref[2]


@pytest.mark.parametrize("make", [scenes.random_sparse, scenes.floor_pillars, scenes.open_sky, scenes.terrain256],
                         ids=["random_sparse64", "floor_pillars32", "open_sky32", "terrain256"])
def test_every_voxel_of_every_box_is_empty_on_the_host(make, atlas):
    """The box words read back (vrc_read_empty_boxes) and checked EXHAUSTIVELY on the host, with nothing of the device's own
    machinery: an independent walk of the descriptor array gives every empty child slot its cube, the word gives the box, and
    the dense grid the tree was generated from must hold no solid voxel anywhere inside it.  Also: every box contains its own
    node (extents are non-negative by construction) and boxes do grow (the structure is not vacuous)."""
    import treetools
    s = make()
    dim = s["dim"]
    grid = np.asarray(s["grid"]).reshape(dim, dim, dim) != 0          # [z, y, x]
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=0 if dim > 64 else 100000)
    c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, 64, 48, 3 * dim, empty_boxes=1)
    assert c.compute(), c.last_error()
    assert c.used_empty_boxes()
    words = c.read_empty_boxes()
    assert words.shape == (o.descriptor_buffer.size, 8)
    # solid voxels in any box in O(1): a summed-volume table of the dense grid, S[z, y, x] = solids in [0, z) x [0, y) x [0, x)
    S = np.zeros((dim + 1, dim + 1, dim + 1), dtype=np.int64)
    S[1:, 1:, 1:] = grid.astype(np.int64).cumsum(0).cumsum(1).cumsum(2)

    def solids(x0, y0, z0, x1, y1, z1):
        return int(S[z1, y1, x1] - S[z0, y1, x1] - S[z1, y0, x1] - S[z1, y1, x0] + S[z0, y0, x1] + S[z0, y1, x0] + S[z1, y0, x0] - S[z0, y0, x0])

    assert solids(0, 0, 0, dim, dim, dim) == int(grid.sum()) > 0
    n_boxes = grown = 0
    volume = node_volume = 0
    for index, k, (x, y, z), size in treetools.empty_children(o.descriptor_buffer, o.root_index, dim):
        w = int(words[index, k])
        ext = [treetools.box_decode((w >> (5 * j)) & 31) * size for j in range(6)]      # -x -y -z +x +y +z
        x0, y0, z0 = max(x - ext[0], 0), max(y - ext[1], 0), max(z - ext[2], 0)
        x1, y1, z1 = min(x + size + ext[3], dim), min(y + size + ext[4], dim), min(z + size + ext[5], dim)
        assert solids(x0, y0, z0, x1, y1, z1) == 0, f"descriptor {index} slot {k}: the box {(x0, y0, z0)}..{(x1, y1, z1)} of the {size}^3 node at {(x, y, z)} holds a solid voxel"
        n_boxes += 1
        grown += any(ext)
        volume += (x1 - x0) * (y1 - y0) * (z1 - z0)
        node_volume += size ** 3
    assert n_boxes > 20 and grown > n_boxes // 4
    print(f"\n{s['name']}: {n_boxes} empty child slots, {grown} of them widened; mean box volume {volume / n_boxes:.0f} voxels "
          f"({volume / node_volume:.1f} x the nodes')")


@pytest.mark.parametrize("depth,lights,w,h", [(8, 1, 640, 360), (10, 2, 640, 360), (12, 1, 1280, 720), (13, 1, 960, 540)], ids=["d8", "d10-2lights", "d12", "d13"])
def test_boxes_for_the_upper_levels_only_never_change_the_frame(depth, lights, w, h):
    """Round 6 (the limit lifted is include/map/Octree.h:29 -- the reference stops at 100 000 descriptors, round 5's
    boxes at 2^28): box records for the descriptors of the tree's upper levels only (setting empty_boxes = 2), numbered breadth-first,
    the record id carried down the traversal stack, nodes below the record levels widened over their empty siblings.  Whatever the number
    of levels -- one (the root alone), a few, all -- with and without the closed-form jumps: the image and the hit records of the
    canonical traversal (read count aside), the same counters; the self-check finds no solid voxel in a sampled box."""
    import bench
    sc = bench.build_scene(depth)
    c = bench.make_caster(sc, w, h, 0, light_count=lights)
    assert c.add_to_settings_buffer("empty_boxes", "EMPTY_BOXES", 0)
    ref = _frame(c)
    assert ref[2]["canonical_reads"]
    full = None
    assert c.add_to_settings_buffer("empty_box_levels", "EMPTY_BOX_LEVELS", 0)
    for levels in (1, 2, depth - 4, depth - 2, 0):
        assert c.overwrite_setting("empty_boxes", 2) and c.overwrite_setting("empty_box_levels", levels)
        for jmr in ((1 << 24, 16) if depth >= 11 else (1 << 24,)):
            assert c.overwrite_setting("jump_min_run", jmr) or c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", jmr)
            img, hits, ctr = _frame(c)
            m = c.memory_usage2()
            tag = f"empty_box_levels={levels} jump_min_run={jmr}: {m['box_levels']} levels, {m['box_records']} records"
            assert c.used_empty_boxes() and not ctr["canonical_reads"] and m["box_records"] > 0, tag
            assert (m["box_levels"] == levels) if levels else (1 <= m["box_levels"] <= depth), tag
            assert np.array_equal(img, ref[0]), f"{tag}: {int((img != ref[0]).any(-1).sum())} pixels differ"
            assert np.array_equal(hits[..., :7], ref[1][..., :7]), f"{tag}: {int((hits[..., :7] != ref[1][..., :7]).any(-1).sum())} hit records differ"
            assert _but_reads(ctr) == _but_reads(ref[2]), tag
        chk = c.empty_boxes_check(1 << 20, seed=depth + levels)
        assert chk["solid_voxels"] == 0 and chk["boxes_sampled"] > 0, (levels, chk)
        if levels == 0:
            full = (m["box_records"], m["box_levels"], ctr["descriptor_reads"])
    # the records with all levels: as many reads as ... at most the canonical traversal's, and fewer records than descriptors x 1
    assert full[0] <= sc["octree"].descriptor_buffer.size + 8 and full[2] <= ref[2]["descriptor_reads"]
    with pytest.raises(Exception):
        c.read_empty_boxes(0, 1)                                    # (records of the upper levels are not indexed by descriptor)
    # back to a word per descriptor on the same tree
    assert c.overwrite_setting("empty_boxes", 1) and c.overwrite_setting("jump_min_run", 1 << 24 if depth < 11 else 16)
    img, hits, ctr = _frame(c)
    assert np.array_equal(img, ref[0]) and np.array_equal(hits[..., :7], ref[1][..., :7]) and c.memory_usage2()["box_levels"] == depth
    print(f"\ndepth {depth}: reads canonical {ref[2]['descriptor_reads'] / 1e6:.2f} M, upper-level records (all levels) {full[2] / 1e6:.2f} M "
          f"({full[0]} records, {full[1]} levels), a word per descriptor {ctr['descriptor_reads'] / 1e6:.2f} M")
