"""GPU suite (-m gpu), round 2: the device-side SVO builder and the real BASELINE configs[4] scene, row slices and
the single-process multi-GPU group handle, the streamed upload against the oracle, the reference camera with its
octree bias active, and the error paths the advisor asked for.  Everything goes through the C ABI (libvrc.so)."""
import functools
import os
import resource
import time

import numpy as np
import pytest

import scenes
import voxel_raycaster_amd as vrc
from oracle import orc
from test_parity_gpu import hits_match

pytestmark = pytest.mark.gpu


@functools.lru_cache(maxsize=4)
def bench_scene(depth):
    import bench
    return bench.build_scene(depth)


def lights4(dim):
    """L0..L3 at the quarter points, 3/4 up, all rgbi (0.01, 0.01, 0.01, 0.2); fractional offsets keep
    shadow rays off exact voxel boundaries."""
    li = np.zeros((8, 10), dtype=np.float32)
    for l, (fx, fy) in enumerate(((0.25, 0.25), (0.75, 0.25), (0.25, 0.75), (0.75, 0.75))):
        li[l] = [0.01, 0.01, 0.01, 0.2, fx * dim + 0.3 * l, fy * dim + 0.2 * l, 0.75 * dim + 0.1 * l, -1.0, -1.0, -1.5]
    return li


def configure(c, dim, atlas, cam_dir, cam_pos, lights, w, h, light_count=1, shadow_rays=1, table=None):
    assert c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim)
    assert c.add_to_settings_buffer("using_octree", "OCTENABLED", 0)
    assert c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", 3 * dim)
    assert c.add_to_settings_buffer("shadow_rays", "SHADOW_RAYS", shadow_rays)
    assert c.add_to_settings_buffer("light_count", "LIGHT_COUNT", light_count)
    cd, cp = np.array(cam_dir, dtype=np.float32), np.array(cam_pos, dtype=np.float32)
    assert c.assign_camera(cd, cp)
    assert c.create_viewport(w, h) if table is None else c.create_viewport_table(table)
    assert c.assign_lights(lights)
    assert c.create_texture_atlas(atlas, (16, 16))
    c._li = lights
    return c


def _rss_now_kb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS:"):
            return int(line.split()[1])
    return 0


def _peak_rss_kb():
    for line in open("/proc/self/status"):
        if line.startswith("VmHWM:"):
            return int(line.split()[1])
    return 0


def _reset_peak_rss():
    """Resets the process's peak-RSS counter (VmHWM) so that a later reading belongs to what ran in between."""
    try:
        with open("/proc/self/clear_refs", "w") as f:
            f.write("5")
        return abs(_peak_rss_kb() - _rss_now_kb()) < 64 * 1024
    except OSError:
        return False


def survey_camera(depth, seed=1, thickness=2, octave_floor=2):
    """SURVEY 8d's camera exactly as written: (D/2 + 0.37, D/8 + 0.41, h(D/2, D/8) + D/16 + 0.29), looking
    (inclination 2.0, azimuth 1.5708); no search for a voxel whose octree bias is zero."""
    dim = 1 << depth
    _, hi = vrc.shell_column(depth, dim // 2, dim // 8, seed=seed, thickness=thickness, octave_floor=octave_floor)
    return (2.0, 1.5708), (dim / 2 + 0.37, dim / 8 + 0.41, hi + dim // 16 + 0.29)


# ------------------------------------------------------------------ device builder
@pytest.mark.parametrize("depth,thickness,floor", [(3, 2, 2), (5, 2, 2), (6, 0, 2), (7, 3, 0), (8, 2, 2), (9, 7, 1), (10, 2, 2), (12, 2, 2)])
def test_device_builder_equals_host_builder(depth, thickness, floor):
    """vrc_build_shell_terrain: the array built in HBM is bit-identical to the sequential host emitter's brick
    layout, Octree::Validate on the device finds no mismatch, and the device height field equals the procedural
    column function."""
    dim = 1 << depth
    host, _ = vrc.shell_terrain_ex(depth, seed=1, thickness=thickness, octave_floor=floor, layout=vrc.LAYOUT_NO_PAGE_HEADERS)
    c = vrc.CLCaster()
    assert c.init(0)
    rng = np.random.default_rng(depth)
    probe = rng.integers(0, dim, size=(64, 2)).astype(np.int32)
    counted, _ = c.build_shell_terrain(depth, 1, thickness, floor, count_only=True)
    assert counted["n_descriptors"] == host.descriptor_buffer.size
    info, lohi = c.build_shell_terrain(depth, 1, thickness, floor, validate_samples=1 << 20, probe_xy=probe)
    assert info["n_descriptors"] == host.descriptor_buffer.size and info["root_index"] == host.root_index
    assert info["validate_samples"] == 1 << 20 and info["validate_mismatches"] == 0
    assert c.octree_size() == (host.descriptor_buffer.size, host.root_index)
    dev = c.read_descriptors()
    assert np.array_equal(dev, host.descriptor_buffer), f"{int((dev != host.descriptor_buffer).sum())} slots differ"
    for (x, y), (lo, hi) in zip(probe, lohi):
        assert (lo, hi) == vrc.shell_column(depth, x, y, seed=1, thickness=thickness, octave_floor=floor)


def test_device_built_scene_renders_like_the_oracle(atlas):
    """A frame of a device-built tree (depth 9, thick shell, survey camera with the octree bias active) against the
    oracle rendering the host-built paged-layout tree of the same scene: the layout never shows in the picture."""
    depth, thickness, w, h = 9, 6, 320, 200
    dim = 1 << depth
    cam_dir, cam_pos = survey_camera(depth, thickness=thickness)
    c = vrc.CLCaster()
    assert c.init(0)
    c.build_shell_terrain(depth, 1, thickness, 2)
    configure(c, dim, atlas, cam_dir, cam_pos, lights4(dim), w, h, light_count=2)
    assert c.validate() and c.compute(), c.last_error()
    host, _ = vrc.shell_terrain_ex(depth, thickness=thickness)     # the reference-style paged layout
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=c._li, atlas=atlas, tile_dim=(16, 16),
                                    descriptors=host.descriptor_buffer, root_index=host.root_index, octree_dim=dim,
                                    using_octree=0, max_distance=3 * dim, active_lights=2, threads=8)
    assert hits_match(c, c.read_hits(), ohits)
    assert np.array_equal(c.read_image().view(np.uint32), oimg.view(np.uint32))
    ctr = c.counters()
    assert (ctr["descriptor_reads"] == octr["n_desc"] or not ctr["canonical_reads"]) and ctr["steps"] == octr["n_steps"]


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


# ------------------------------------------------------------------ reference camera, bias active
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


# ------------------------------------------------------------------ row slices and the group handle
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
        assert hits_match(g, hits[y0], ohits[y0])
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))


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


# ------------------------------------------------------------------ streamed upload vs the ORACLE
@pytest.mark.parametrize("depth,chunk", [(10, 1 << 20), (12, 64 << 20), (13, 64 << 20)], ids=["d10-1MB-chunks", "d12-3-chunks", "d13-11-chunks"])
def test_streamed_upload_multi_chunk_against_the_oracle(depth, chunk, tmp_path):
    """vrc_assign_octree_file with trees that need many staging-buffer cycles (depth 12: 161 MB, depth 13: 674 MB with
    far pointers) and attachments (mirrors every 64th voxel): sampled rows bit-exact vs the oracle on the in-memory
    arrays."""
    sc = bench_scene(depth)
    dim, w, h = sc["dim"], 1024, 576
    tree = sc["octree"]
    if tree.attachment_lookup is None:
        tree.attach_materials_procedural(depth, seed=1, mirror_period=64)
    path = str(tmp_path / "scene.svo")
    tree.Save(path)
    assert os.path.getsize(path) > 2 * chunk
    c = vrc.CLCaster()
    assert c.init(0)
    assert c.add_to_settings_buffer("upload_chunk_bytes", "UPLOAD_CHUNK_BYTES", chunk)
    assert c.assign_octree_file(path) == dim, c.last_error()
    os.remove(path)
    configure(c, dim, sc["atlas"], sc["cam_dir"], sc["cam_pos"], sc["lights"], w, h)
    assert c.validate() and c.compute(), c.last_error()
    img, hits = c.read_image(), c.read_hits()
    assert (hits[..., 3] == 6).sum() > 100                          # mirrors came through the file
    for y0 in range(5, h, 57):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"], lights=c._li, atlas=sc["atlas"],
                                     tile_dim=(16, 16), descriptors=tree.descriptor_buffer, root_index=tree.root_index, octree_dim=dim,
                                     using_octree=0, max_distance=3 * dim, rows=(y0, y0 + 1), threads=16,
                                     attachment_lookup=tree.attachment_lookup, attachments=tree.attachment_buffer)
        assert hits_match(c, hits[y0], ohits[y0])
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))


# ------------------------------------------------------------------ advisor findings
def test_a_new_tree_never_inherits_the_old_trees_materials(atlas):
    """vrc_assign_octree after a tree with attachments: the material buffers of the old tree are gone (they are indexed
    by the old tree's descriptor indices)."""
    from test_oracle_cpu import _with_pass_through
    s = _with_pass_through(scenes.mirror_wall())
    dim, w, h = s["dim"], 128, 96
    with_mat = vrc.Octree.Generate(s["grid"], dim).attach_materials_from_grid(s["grid"])
    t = scenes.random_sparse()
    tdim = t["dim"]
    bigger = vrc.Octree.Generate(t["grid"], tdim)                   # another, larger tree without attachments
    assert bigger.descriptor_buffer.size > with_mat.descriptor_buffer.size
    li = np.zeros((8, 10), dtype=np.float32)
    li[:1] = t["lights"]
    c = vrc.CLCaster()
    assert c.init(0) and c.assign_octree(with_mat)
    configure(c, dim, atlas, t["cam_dir"], t["cam_pos"], li, w, h)
    assert c.validate() and c.compute()
    assert c.assign_octree(bigger) and c.overwrite_setting("octree_dimensions", tdim) and c.overwrite_setting("max_distance", 3 * tdim)
    assert c.validate() and c.compute(), c.last_error()
    oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=t["cam_dir"], cam_pos=t["cam_pos"], lights=li, atlas=atlas, tile_dim=(16, 16),
                                 descriptors=bigger.descriptor_buffer, root_index=bigger.root_index, octree_dim=tdim, using_octree=0,
                                 max_distance=3 * tdim)
    assert hits_match(c, c.read_hits(), ohits) and np.array_equal(c.read_image().view(np.uint32), oimg.view(np.uint32))
    bad = with_mat.attachment_lookup.copy()
    bad[5] = with_mat.attachment_buffer.size + 3
    import ctypes as C
    assert c.assign_octree(with_mat) and c.overwrite_setting("octree_dimensions", dim)
    rc = vrc.lib.vrc_assign_octree_attachments(c._h, bad.ctypes.data_as(C.POINTER(C.c_uint32)), bad.size,
                                               with_mat.attachment_buffer.ctypes.data_as(C.POINTER(C.c_uint64)), with_mat.attachment_buffer.size)
    assert rc == 1 and "past the attachment buffer" in c.last_error()


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


# ------------------------------------------------------------------ round 3: the distinct-GPU code paths, rehearsed on one GPU
def test_group_with_own_copies_takes_the_cross_device_path(atlas):
    """VRC_GROUP_OWN_COPIES: every rank sits on GPU 0 (all this box has) but takes the path of a rank on another GPU --
    own allocation + hipMemcpyPeerAsync of the tree, attachment re-copy, release -- on the depth-12 tree with
    attachments, 4 ranks.  Frame, hit records, RGBA8 and counters equal the single handle's; no rank shares rank 0's
    arrays.  (A same-GPU rehearsal: no 8-GPU hardware run exists, and no scaling number is claimed.)"""
    sc = bench_scene(12)
    dim, w, h = sc["dim"], 1280, 720
    tree = sc["octree"]
    if tree.attachment_lookup is None:
        tree.attach_materials_procedural(12, seed=1, mirror_period=64)
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
