"""BASELINE configs at their full sizes (-m gpu) -- configs[3] (4096^3, 3840x2160, 2 lights, 8 ranks), configs[4] (65536^3,
~198 GB resident, 7680x4320, 4 lights), the headline frame with the reference's octree bias active (ray_caster_kernel.cl:353-354),
the deepest trees the traversal stack allows: sampled rows against the oracle, which reads the descriptors it needs from the GPU
page by page."""
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

def test_configs4_scene_200GB_resident_sampled_rows(atlas):
    """BASELINE configs[4]: 65536^3 (depth 16) sparse SVO, >= 2e10 descriptors (~190 GB) resident in HBM, built on the
    device in seconds with a few MB of host memory; 7680x4320, 4 lights.  Octree::Validate on the device over 2^28
    voxels, sampled columns against the procedural scene function, and sampled rows of the frame bit-exact against
    the oracle, which reads the descriptors it needs from the GPU page by page."""
    depth, thickness, floor, w, h = 16, int(os.environ.get("VRC_C5_THICKNESS", "33")), 2, 7680, 4320
    dim = 1 << depth
    # configs[4] row-tiles the frame over 8 GPUs with the SVO replicated: an 8-rank group handle, every rank on the one
    # GPU this box has (ranks on rank 0's GPU share its 198 GB array; on an 8-GPU node each rank gets a peer copy)
    c = vrc.CLCaster()
    assert c.init_group([0] * 8, band_rows=8) and c.group_size() == 8
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    cur0, hwm_reset = _rss_now_kb(), _reset_peak_rss()
    rng = np.random.default_rng(16)
    probe = rng.integers(0, dim, size=(256, 2)).astype(np.int32)
    t0 = time.perf_counter()
    info, lohi = c.build_shell_terrain(depth, 1, thickness, floor, validate_samples=1 << 28, probe_xy=probe)
    wall = time.perf_counter() - t0
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    print(f"\nconfigs[4] scene: {info['n_descriptors'] / 1e9:.2f} G descriptors = {info['n_descriptors'] * 8 / 1e9:.1f} GB in HBM, "
          f"{info['n_bricks']} bricks, built in {wall:.1f} s (height {info['seconds_height']:.2f} count {info['seconds_count']:.2f} "
          f"emit {info['seconds_emit']:.2f}), host tables {info['host_bytes'] / 1e6:.0f} MB, device peak {info['device_bytes_peak'] / 1e9:.1f} GB")
    assert info["n_descriptors"] >= 20_000_000_000
    assert info["validate_mismatches"] == 0 and info["validate_samples"] == 1 << 28
    # host memory: the build may not grow the process by more than 2 GB.  The peak counter of the process is reset right
    # before the build (/proc/self/clear_refs), so the high-water mark read here belongs to the build alone; where the
    # kernel does not allow that, ru_maxrss (high-water mark of the whole pytest process) stands in
    grown_kb = (_peak_rss_kb() - cur0) if hwm_reset else (rss1 - rss0)
    print(f"host memory added by the build: {grown_kb / 1e6:.3f} GB ({'peak counter reset before the build' if hwm_reset else 'ru_maxrss delta'})")
    assert wall < 60.0 and grown_kb < 2e6
    for (x, y), (lo, hi) in zip(probe, lohi):
        assert (lo, hi) == vrc.shell_column(depth, x, y, seed=1, thickness=thickness, octave_floor=floor)

    cam_dir, cam_pos = survey_camera(depth, thickness=thickness, octave_floor=floor)
    configure(c, dim, atlas, cam_dir, cam_pos, lights4(dim), w, h, light_count=4)
    assert c.validate() and c.compute(), c.last_error()
    n_launch, ms = c.timing()
    ctr = c.counters()
    print(f"configs[4] frame, 8 row-sliced ranks on one GPU: slowest rank {ms / n_launch:.1f} ms of kernel time, "
          f"{ctr['primary_rays'] + ctr['shadow_rays']} rays, {ctr['steps'] / 1e9:.1f} G steps, {ctr['descriptor_reads'] / 1e6:.1f} M descriptor reads")
    mem = [c.memory_usage(r) for r in range(8)]
    assert sum(m["rows"] for m in mem) == h and {m["rows"] for m in mem} == {536, 544}      # 540 bands of 8 rows over 8 ranks
    assert all(m["image_bytes"] == 16 * w * m["rows"] for m in mem)
    assert mem[0]["octree_bytes"] == info["n_descriptors"] * 8 and all(m["octree_shared"] == 1 for m in mem[1:])
    assert ctr["primary_rays"] == w * h and ctr["shadow_rays"] > w * h
    img, hits = c.read_image(), c.read_hits()
    assert ctr["descriptor_reads"] == int(hits[..., 7].astype(np.int64).sum())
    n, root = c.octree_size()
    paged = orc.PagedDescriptors(n, c.read_descriptors)
    threads = max(1, min(32, len(os.sched_getaffinity(0))))
    t0 = time.perf_counter()
    # 34 rows: the first row of every rank's first band and the last row of its last band (row 0 and row 4319 among them),
    # and 18 rows spread over the frame
    n_bands = h // 8
    rows = set()
    for rank in range(8):
        last_band = max(b for b in range(n_bands) if b % 8 == rank)
        rows.update((8 * rank, 8 * last_band + 7))
    rows.update(range(121, h, 241))
    assert 0 in rows and h - 1 in rows and len(rows) >= 32
    for y0 in sorted(rows):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=c._li, atlas=atlas, tile_dim=(16, 16),
                                     descriptors=paged, root_index=root, octree_dim=dim, using_octree=0, max_distance=3 * dim,
                                     rows=(y0, y0 + 1), threads=threads, active_lights=4)
        assert hits_match(c, hits[y0], ohits[y0]), f"row {y0}: {int((hits[y0][..., :7] != ohits[y0][..., :7]).any(-1).sum())} pixels differ"
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))
    print(f"oracle: {len(rows)} rows in {time.perf_counter() - t0:.1f} s on {threads} threads, {paged.bytes_fetched / 1e6:.0f} MB of descriptors fetched")


def test_configs3_eight_rank_group_sampled_rows():
    """BASELINE configs[3]: 4096^3, 3840x2160, 2 lights, row-tiled over 8 ranks with the SVO replicated -- an 8-rank group
    handle (all ranks on this box's one GPU), sampled rows bit-exact vs the oracle, an eighth of the rows per rank."""
    sc = bench_scene(12)
    dim, w, h = sc["dim"], 3840, 2160
    g = vrc.CLCaster()
    assert g.init_group([0] * 8, band_rows=8) and g.assign_octree(sc["octree"])
    configure(g, dim, sc["atlas"], sc["cam_dir"], sc["cam_pos"], sc["lights"], w, h, light_count=2)
    assert g.validate() and g.compute(), g.last_error()
    rows = [g.memory_usage(r)["rows"] for r in range(8)]
    assert sum(rows) == h and set(rows) == {264, 272}                # 270 bands of 8 rows over 8 ranks
    img, hits, ctr = g.read_image(), g.read_hits(), g.counters()
    assert ctr["primary_rays"] == w * h and ctr["descriptor_reads"] == int(hits[..., 7].astype(np.int64).sum())
    for y0 in range(17, h, 307):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"], lights=g._li, atlas=sc["atlas"],
                                     tile_dim=(16, 16), descriptors=sc["octree"].descriptor_buffer, root_index=sc["octree"].root_index,
                                     octree_dim=dim, using_octree=0, max_distance=3 * dim, rows=(y0, y0 + 1), threads=16, active_lights=2)
        bad = (hits[y0][..., :7] != ohits[y0][..., :7])
        assert hits_match(g, hits[y0], ohits[y0]), (f"row {y0}: {int(bad.any(-1).sum())} pixels differ in fields {sorted(set(np.nonzero(bad)[1].tolist()))}; "
                                                      f"boxes used: {[g.memory_usage2(r)['empty_boxes'] for r in range(8)]}, note {g.memory_usage2()['note']!r}")
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))


def test_headline_size_frame_with_the_reference_bias_active(atlas):
    """BASELINE-size frame (depth 12, 1920x1080) from SURVEY 8d's camera as written, where the reference's
    intersection_t bias (ray_caster_kernel.cl:353-354) is not zero: sampled rows bit-exact vs the oracle."""
    depth, w, h = 12, 1920, 1080
    dim = 1 << depth
    sc = bench_scene(depth)
    cam_dir, cam_pos = survey_camera(depth)
    found, res, sub = sc["octree"].GetVoxel(tuple(int(np.floor(v)) for v in cam_pos))
    bias = [(s - int(np.floor(v))) * res // 2 for s, v in zip(sub, cam_pos)]
    assert not found and any(b != 0 for b in bias), "this camera must exercise the bias term"
    c = vrc.CLCaster()
    assert c.init(0) and c.assign_octree(sc["octree"])
    configure(c, dim, atlas, cam_dir, cam_pos, sc["lights"], w, h)
    assert c.validate() and c.compute(), c.last_error()
    img, hits = c.read_image(), c.read_hits()
    for y0 in range(11, h, 97):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=c._li, atlas=atlas, tile_dim=(16, 16),
                                     descriptors=sc["octree"].descriptor_buffer, root_index=sc["octree"].root_index, octree_dim=dim,
                                     using_octree=0, max_distance=3 * dim, rows=(y0, y0 + 1), threads=16)
        assert hits_match(c, hits[y0], ohits[y0])
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))
    # and it is not the frame the unbiased kernel renders
    assert c.add_to_settings_buffer("octree_bias", "OCTREE_BIAS", 0) and c.compute()
    assert not np.array_equal(c.read_hits()[540], hits[540])


@pytest.mark.parametrize("mode", [0, 1], ids=["exact", "mode_b"])
@pytest.mark.parametrize("depth", [20, 24], ids=["d20", "d24-deepest-the-stack-allows"])
def test_deepest_trees(depth, mode, atlas):
    """Maximum depth: a hand-laid tree of 20 / 24 levels (kMaxLevels = 24: dim 16 777 216) that holds one small
    scene a million voxels from the origin -- the LDS stack at its largest, voxel coordinates at the edge of what a
    float carries exactly, both stepping modes against the oracle, every pixel."""
    from test_parity_gpu import assert_same
    import treetools
    dim = 1 << depth
    bx, by, bz = (1000000, 1000000, 1000000) if depth == 24 else (300000, 700000, 500000)
    vox = [(bx + x, by + y, bz) for x in range(-20, 21) for y in range(-20, 21)]                  # a floor
    vox += [(bx + x, by + y, bz - 1) for x in range(-20, 21) for y in range(-20, 21)]
    vox += [(bx + px, by + py, bz + 1 + k) for px, py in ((-6, 3), (5, 9), (2, -7)) for k in range(7)]   # pillars
    desc, root = treetools.sparse_octree(vox, depth)
    oct_ = vrc.Octree(desc, root, dim)
    cam_dir, cam_pos = (1.85, 1.5708), (bx + 0.375, by - 14.625, bz + 9.25)    # the slab's far edge crosses the frame
    li = np.zeros((8, 10), dtype=np.float32)
    li[0] = [0.01, 0.01, 0.01, 0.2, bx + 11.5, by - 5.25, bz + 17.75, -1.0, -1.0, -1.5]
    w, h, md = 320, 200, 700
    c = vrc.CLCaster()
    assert c.init(0) and c.assign_octree(oct_)
    configure(c, dim, atlas, cam_dir, cam_pos, li, w, h)
    assert c.overwrite_setting("max_distance", md) and c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", mode)
    assert c.validate() and c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=li, atlas=atlas, tile_dim=(16, 16),
                                    descriptors=desc, root_index=root, octree_dim=dim, using_octree=0, max_distance=md,
                                    stepping_mode=mode)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    hit = float((ohits[..., 3] == 5).mean())
    assert 0.5 < hit < 0.9 and octr["shadow_rays"] > 0       # slab and pillars below, 700 steps of 2^23-voxel nodes above
