"""Closed-form jumps and the coarse top table of the exact kernel (-m gpu) -- SURVEY 8 rows a5 / a8 (the DDA recurrence of
ray_caster_kernel.cl:558-560 and the node-sized steps :525 was heading for): jumps on / off / forced, Euclid tables in LDS or in
global memory, table levels: one image, one set of hit records, the same counters.  (csrc/exact_jump.hpp against the plain loop on
the host: tests/test_exact_jump.py; the long differential run: tests/soak_jumps_gpu.py.)"""
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

@pytest.mark.parametrize("depth,lights,k", [(10, 1, 8), (12, 1, 96), (12, 4, 32), (13, 2, 96)],
                         ids=["d10-K8", "d12-default", "d12-4lights-K32", "d13-2lights"])
def test_exact_jumps_leave_the_frame_bit_identical(depth, lights, k):
    """exact_jump.hpp inside the step kernel: the frame -- image, hit records, every counter -- with jumps is the frame
    without them, on whole 1080p frames (primary + shadow rays, multi-light relighting, mirrors via attachments at depth
    10), whatever the threshold."""
    import bench
    sc = bench.build_scene(depth)
    tree = sc["octree"]
    if depth == 10 and tree.attachment_lookup is None:
        tree.attach_materials_procedural(depth, seed=1, mirror_period=64)
    c = bench.make_caster(sc, 1920, 1080, 0, light_count=lights)
    assert c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 1 << 24)
    times = {}
    frames = {}
    for setting in (1 << 24, k):
        assert c.overwrite_setting("jump_min_run", setting)
        for _ in range(2):
            assert c.compute(), c.last_error()
        c.timing_reset()
        for _ in range(4):
            assert c.compute(), c.last_error()
        n, ms = c.timing()
        times[setting] = ms / n
        frames[setting] = (c.read_image(), c.read_hits(), c.counters())
    a, b = frames[1 << 24], frames[k]
    assert a[2] == b[2]
    assert np.array_equal(a[1], b[1]), f"{int((a[1] != b[1]).any(-1).sum())} pixels differ in hit records"
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    print(f"\ndepth {depth}, {lights} light(s): {times[1 << 24]:.3f} ms stepping, {times[k]:.3f} ms with jumps (jump_min_run {k})")


def test_three_casters_in_flight_with_jumps_keep_their_frames():
    """Three handles (three HIP streams, three jump-table buffers) rendering different sizes of the depth-12 scene at the
    same time, frame after frame without a host sync in between, closed-form jumps on: every caster's frame stays the
    frame it renders alone without jumps -- the kernels of different handles share CUs, L2s and XCDs, not tables."""
    import bench
    sc = bench.build_scene(12)
    sizes = [(640, 360), (1920, 1080), (200, 136)]
    casters, refs = [], []
    for w, h in sizes:
        c = bench.make_caster(sc, w, h, 0)
        assert c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 1 << 24) and c.compute()
        refs.append((c.read_image(), c.read_hits()))
        assert c.overwrite_setting("jump_min_run", 96)
        casters.append(c)
    for _ in range(6):
        for _ in range(5):
            for c in casters:
                assert c.compute_async(), c.last_error()
        for c, (img, hits) in zip(casters, refs):
            assert c.sync(), c.last_error()
            assert np.array_equal(c.read_hits(), hits) and np.array_equal(c.read_image().view(np.uint32), img.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("depth,lights", [(8, 1), (10, 2), (12, 1)], ids=["d8", "d10-2lights", "d12"])
def test_coarse_table_and_lds_tables_never_change_the_frame(depth, lights):
    """Round 4's two memory-side changes of the exact kernel are invisible in its results: with the dense table of the tree's
    top (setting coarse_log2: by depth, coarser, none) and with the Euclid tables of the closed-form jumps in LDS or in global
    memory, the image, the hit records -- the canonical descriptor-read count of every pixel included, which the table path
    gets by arithmetic -- and every counter are the same, jumps on (threshold 16 so that small trees jump too) and off."""
    import bench
    sc = bench.build_scene(depth)
    w, h = (640, 360) if depth < 12 else (1920, 1080)
    c = bench.make_caster(sc, w, h, 0, light_count=lights)
    # (empty_boxes = 0: this test is about the CANONICAL read count, which the table path keeps; the boxes -- round 5, on by
    # default wherever the table is -- count their own reads: tests/test_boxes_gpu.py)
    for name, v in (("coarse_log2", 0), ("jump_tables_lds", 0), ("jump_min_run", 1 << 24), ("empty_boxes", 0)):
        assert c.add_to_settings_buffer(name, name.upper(), v)
    ref = _frame(c)                                        # no table, no jumps: the plain traversal
    assert c.memory_usage()["coarse_bytes"] == 0
    for coarse in (-1, max(depth - 5, 1), 0):
        for jmr, lds in ((1 << 24, 2), (16, 2), (16, 0), (16, 1)):
            assert c.overwrite_setting("coarse_log2", coarse) and c.overwrite_setting("jump_min_run", jmr) and c.overwrite_setting("jump_tables_lds", lds)
            img, hits, ctr = _frame(c)
            tag = f"coarse_log2={coarse} jump_min_run={jmr} jump_tables_lds={lds}"
            assert ctr == ref[2], tag
            assert np.array_equal(hits, ref[1]), f"{tag}: {int((hits != ref[1]).any(-1).sum())} pixels differ in hit records"
            assert np.array_equal(img, ref[0]), tag
        want = 0 if coarse == 0 else 8 << (3 * (min(depth - 2, 9) if coarse < 0 else coarse))
        assert c.memory_usage()["coarse_bytes"] == want


@pytest.mark.gpu
def test_mode_b_coarse_table_changes_only_the_read_count():
    """Mode B with and without the table: the same frame and hit records (voxel, face, material, flags, step count); only the
    descriptor reads differ -- and those are restated in the oracle for both (tests/test_mode_b_gpu.py compares them)."""
    import bench
    sc = bench.build_scene(10)
    c = bench.make_caster(sc, 640, 360, 0)
    assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 1) and c.add_to_settings_buffer("coarse_log2", "COARSE_LOG2", 0)
    img0, hits0, ctr0 = _frame(c)
    for coarse in (-1, 5, 8):
        assert c.overwrite_setting("coarse_log2", coarse)
        img, hits, ctr = _frame(c)
        assert np.array_equal(img, img0) and np.array_equal(hits[..., :7], hits0[..., :7])
        assert {k: v for k, v in ctr.items() if k != "descriptor_reads"} == {k: v for k, v in ctr0.items() if k != "descriptor_reads"}
        assert ctr["descriptor_reads"] != ctr0["descriptor_reads"]
