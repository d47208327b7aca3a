"""SVO builders (-m gpu) -- SURVEY 8 rows a1 / a2 (descriptor format, Octree::Generate, src/map/Octree.cpp:13-43,171-323), f3
((de)serialisation + streamed upload, Octree::Load include/map/Octree.h:38) and f4 (diamond-square terrain, src/map/Map.cpp:144-262):
the device builders against the sequential host emitter bit for bit, trees in the reference builder's own buffer layout through
both branches, device-built scenes and streamed uploads against the oracle.  Everything goes through the C ABI (libvrc.so)."""
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


@pytest.mark.parametrize("depth", [8, 9, 10])
def test_device_heightfield_builder_equals_the_host_emitter(depth):
    """vrc_build_heightfield on the reference's diamond-square height field (Map::GenerateHeightBitmap, Map.cpp:144-262):
    the array built in HBM is bit-identical to the sequential host emitter's (vrc_octree_from_columns), with solid
    columns (lo = 0) and with a shell (lo = hi - 5); the device-side Octree::Validate finds no mismatch."""
    dim = 1 << depth
    height, _ = vrc.diamond_square(dim, want_grid=False)
    hi = np.minimum(height.astype(np.uint16), dim - 1)
    for lo in (None, np.maximum(hi.astype(np.int32) - 5, 0).astype(np.uint16)):
        host = vrc.octree_from_columns(depth, hi, lo, layout=2)
        c = vrc.CLCaster()
        assert c.init(0)
        info = c.build_heightfield(depth, hi, lo, validate_samples=1 << 20)
        assert info["validate_mismatches"] == 0 and info["validate_samples"] == 1 << 20
        n, root = c.octree_size()
        assert n == host.descriptor_buffer.size == info["n_descriptors"] and root == host.root_index
        assert np.array_equal(c.read_descriptors(), host.descriptor_buffer)
    with pytest.raises(vrc.VrcError):
        c.build_heightfield(depth, np.full((dim, dim), dim, dtype=np.uint16))        # a column taller than the map


@pytest.mark.parametrize("depth,density", [(3, 0.3), (5, 0.5), (6, 0.02), (7, 0.1), (8, 0.004), (8, 0.6)])
def test_device_dense_grid_builder_equals_the_host_emitter(depth, density, atlas):
    """vrc_build_dense_grid = Octree::Generate (src/map/Octree.cpp:13-43) on the device: for random grids (plus an empty
    and a solid one) the array built in HBM is bit-identical to the sequential host emitter's in the same layout, the
    device-side validate finds no mismatch, and a frame rendered from it equals the oracle's frame of the host-built tree."""
    dim = 1 << depth
    rng = np.random.default_rng(depth * 1000 + int(density * 1000))
    grids = [(rng.random(dim ** 3) < density).astype(np.int8) * 5]
    if depth == 5:
        grids += [np.zeros(dim ** 3, dtype=np.int8), np.full(dim ** 3, 5, dtype=np.int8)]
    if depth == 7:                                             # a few voxels only: most bricks are no candidates at all
        g = np.zeros(dim ** 3, dtype=np.int8)
        g[rng.integers(0, dim ** 3, 7)] = 5
        grids.append(g)
    for g in grids:
        host = vrc.Octree.Generate(g, dim, layout=2)
        c = vrc.CLCaster()
        assert c.init(0)
        info = c.build_dense_grid(depth, g, validate_samples=1 << 18)
        assert info["validate_mismatches"] == 0 and info["validate_samples"] == 1 << 18
        n, root = c.octree_size()
        assert n == host.descriptor_buffer.size == info["n_descriptors"] and root == host.root_index
        assert np.array_equal(c.read_descriptors(), host.descriptor_buffer)
        assert c.build_dense_grid(depth, g, count_only=True)["n_descriptors"] == n
        # the same tree from the map of the array branch, once that is resident (no second upload)
        with pytest.raises(vrc.VrcError):
            c.build_dense_grid(depth, None)
        assert c.assign_map(g, (dim, dim, dim))
        assert c.build_dense_grid(depth, None, validate_samples=1 << 16)["validate_mismatches"] == 0
        assert np.array_equal(c.read_descriptors(), host.descriptor_buffer)
    # the last grid, rendered straight from the device-built tree
    w, h, md = 96, 64, 3 * dim
    g3 = g.reshape(dim, dim, dim)                                  # the camera sits in the empty voxel nearest to the centre
    empty = np.argwhere(g3 == 0)                                   # (rays that start outside the map end at once, :563)
    cz, cy, cx = empty[np.abs(empty - dim // 2).sum(1).argmin()] if len(empty) else (dim // 2,) * 3
    cam_pos, cam_dir = (float(cx) + 0.3, float(cy) + 0.6, float(cz) + 0.2), (1.7, 1.6)
    li = np.zeros((8, 10), dtype=np.float32)
    li[0] = [0.01, 0.01, 0.01, 0.2, dim * 0.8, dim * 0.2, dim * 1.1, 0, 0, -1]
    ok = (c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim) and c.add_to_settings_buffer("using_octree", "OCTENABLED", 0)
          and c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", md)
          and c.assign_camera(np.array(cam_dir, dtype=np.float32), np.array(cam_pos, dtype=np.float32)) and c.create_viewport(w, h)
          and c.assign_lights(li) and c.create_texture_atlas(atlas, (16, 16)) and c.validate() and c.compute())
    assert ok, c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=li, atlas=atlas, tile_dim=(16, 16),
                                    descriptors=host.descriptor_buffer, root_index=host.root_index, octree_dim=dim, using_octree=0,
                                    max_distance=md)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    with pytest.raises(vrc.VrcError):
        c.build_dense_grid(13, np.zeros(8, dtype=np.int8))
    # the map's materials (mirrors among them) for the device-built tree: attachments from the array read back once
    mats = rng.choice(np.array([5, 6, 1], dtype=np.int8), size=g.size, p=[0.7, 0.2, 0.1])
    gm = np.where(g != 0, mats, 0).astype(np.int8)
    tree = vrc.Octree(c.read_descriptors(), c.octree_size()[1], dim).attach_materials_from_grid(gm)
    assert c.assign_octree_attachments(tree) and c.validate() and c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=li, atlas=atlas, tile_dim=(16, 16),
                                    descriptors=host.descriptor_buffer, root_index=host.root_index, octree_dim=dim, using_octree=0,
                                    max_distance=md, attachment_lookup=tree.attachment_lookup, attachments=tree.attachment_buffer)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    # ... and the same materials made on the device in the build itself (VRC_BUILD_ATTACHMENTS): the same frame
    c2 = vrc.CLCaster()
    assert c2.init(0)
    info = c2.build_dense_grid(depth, gm, validate_samples=1 << 16, attachments=True)
    assert info["validate_mismatches"] == 0 and np.array_equal(c2.read_descriptors(), host.descriptor_buffer)
    ok = (c2.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim) and c2.add_to_settings_buffer("using_octree", "OCTENABLED", 0)
          and c2.add_to_settings_buffer("max_distance", "MAX_DISTANCE", md)
          and c2.assign_camera(np.array(cam_dir, dtype=np.float32), np.array(cam_pos, dtype=np.float32)) and c2.create_viewport(w, h)
          and c2.assign_lights(li) and c2.create_texture_atlas(atlas, (16, 16)) and c2.validate() and c2.compute())
    assert ok, c2.last_error()
    assert_same(c2.read_image(), c2.read_hits(), c2.counters(), oimg, ohits, octr)
    if depth >= 6 and 0.01 <= float((g != 0).mean()) <= 0.3:
        assert len(np.unique(c.read_hits()[..., 3])) >= 3           # several materials are in the picture


@pytest.mark.parametrize("own_copies", [False, True], ids=["shared", "own-copies"])
def test_dense_grid_build_on_a_group_handle(own_copies, atlas):
    """vrc_build_dense_grid on rank 0 of a 3-rank group (all on this GPU; with own copies the ranks take the copy path of
    ranks on other GPUs): the tree and its device-made material attachments reach every rank -- the gathered frame equals
    the single handle's, mirrors included."""
    import soak_array_vs_svo_gpu
    depth, dim, w, h = 7, 128, 200, 136
    grid = soak_array_vs_svo_gpu.make_map(np.random.default_rng(77), depth)
    empty = np.argwhere(grid == 0)
    z, y, x = empty[len(empty) // 2]
    cam = (np.array([1.6, 0.9], dtype=np.float32), np.array([x + 0.4, y + 0.5, z + 0.3], dtype=np.float32))
    views = [(1.6, 0.9), (1.6, 2.5), (1.6, 4.0), (1.6, 5.6), (0.4, 1.0), (2.7, 1.0)]      # the single handle keeps the busiest one
    li = np.zeros((8, 10), dtype=np.float32)
    li[:, 0:4] = 0.6
    li[:, 4:7] = np.random.default_rng(78).random((8, 3)) * dim
    frames = []
    for ranks in (1, 3):
        c = vrc.CLCaster()
        assert c.init(0) if ranks == 1 else c.init_group([0] * ranks, band_rows=8, own_copies=own_copies)
        assert c.assign_map(grid, (dim, dim, dim))
        info = c.build_dense_grid(depth, None, validate_samples=1 << 16, attachments=True)
        assert info["validate_mismatches"] == 0
        ok = (c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim) and c.add_to_settings_buffer("using_octree", "OCTENABLED", 0)
              and c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", 3 * dim) and c.add_to_settings_buffer("light_count", "LIGHT_COUNT", 2)
              and c.assign_camera(*cam) and c.create_viewport(w, h) and c.assign_lights(li) and c.create_texture_atlas(atlas, (16, 16))
              and c.validate() and c.compute())
        assert ok, c.last_error()
        if ranks == 1:
            seen = []
            for v in views:
                cam[0][:] = v
                assert c.compute()
                seen.append(len(np.unique(c.read_hits()[..., 3])))
            cam[0][:] = views[int(np.argmax(seen))]
            assert c.compute()
        frames.append((c.read_image(), c.read_hits(), c.counters()))
        if ranks == 3:
            mem = [c.memory_usage(r) for r in range(3)]
            assert [m["octree_shared"] for m in mem[1:]] == ([0, 0] if own_copies else [1, 1]), mem
    assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32)) and np.array_equal(frames[0][1], frames[1][1])
    assert frames[0][2] == frames[1][2]
    assert len(np.unique(frames[0][1][..., 3])) >= 3                # several materials in the picture


def test_depth13_diamond_square_terrain_against_the_oracle(atlas):
    """f4 past the dense-grid limit: the 8192^2 diamond-square height field (67 M mt19937 draws on the host, in the
    reference's order) built into an SVO on the device -- no 8192^3 grid anywhere -- and rendered at 1080p; sampled rows
    bit-exact against the oracle, which reads the descriptors it needs from the GPU page by page."""
    depth = 13
    dim = 1 << depth
    height, _ = vrc.diamond_square(dim, want_grid=False)
    hi = np.minimum(height.astype(np.uint16), dim - 1)
    c = vrc.CLCaster()
    assert c.init(0)
    info = c.build_heightfield(depth, hi, None, validate_samples=1 << 22)
    assert info["validate_mismatches"] == 0
    print(f"\ndepth-13 diamond-square terrain: {info['n_descriptors'] / 1e6:.1f} M descriptors built in {info['seconds_total']:.2f} s "
          f"(host tables {info['host_bytes'] / 1e6:.0f} MB)")
    w, h, md = 1920, 1080, 3 * dim
    cam_pos = np.array([dim / 2 + 0.37, dim / 8 + 0.41, float(height.max()) + 40.29], dtype=np.float32)
    cam_dir = np.array([1.75, 1.5708], dtype=np.float32)
    li = np.zeros((8, 10), dtype=np.float32)
    li[0] = [0.01, 0.01, 0.01, 0.2, dim / 2, dim / 3, 400.0, -1.0, -1.0, -1.5]
    ok = (c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim) and c.add_to_settings_buffer("using_octree", "OCTENABLED", 0)
          and c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", md) and c.assign_camera(cam_dir, cam_pos)
          and c.create_viewport(w, h) and c.assign_lights(li) and c.create_texture_atlas(atlas, (16, 16)) and c.validate() and c.compute())
    assert ok, c.last_error()
    img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
    assert (hits[..., 3] == 5).sum() > w * h // 4 and ctr["shadow_rays"] > w * h // 4
    n, root = c.octree_size()
    paged = orc.PagedDescriptors(n, c.read_descriptors)
    for y0 in range(37, h, 131):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=li, atlas=atlas, tile_dim=(16, 16),
                                     descriptors=paged, root_index=root, octree_dim=dim, using_octree=0, max_distance=md,
                                     rows=(y0, y0 + 1), threads=16)
        assert hits_match(c, hits[y0], ohits[y0]), f"row {y0}: {int((hits[y0][..., :7] != ohits[y0][..., :7]).any(-1).sum())} pixels differ"
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))


@pytest.mark.parametrize("depth,chunk", [(10, 1 << 20), (12, 64 << 20), (13, 64 << 20)], ids=["d10-1MB-chunks", "d12-3-chunks", "d13-11-chunks"])
def test_streamed_upload_multi_chunk_against_the_oracle(depth, chunk, tmp_path):
    """vrc_assign_octree_file with trees that need many staging-buffer cycles (depth 12: 161 MB, depth 13: 674 MB with
    far pointers) and attachments (mirrors every 64th voxel): sampled rows bit-exact vs the oracle on the in-memory
    arrays."""
    sc = bench_scene(depth)
    dim, w, h = sc["dim"], 1024, 576
    # (a fresh Octree object over the cached scene's arrays: the materials must not follow the scene into other tests)
    tree = vrc.Octree(sc["octree"].descriptor_buffer, sc["octree"].root_index, dim).attach_materials_procedural(depth, seed=1, mirror_period=64)
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


@pytest.mark.parametrize("dim,density,seed", [(64, 0.5, 7), (128, 0.02, 5)], ids=["64^3-half-full", "128^3-sparse"])
def test_strict_reference_buffer_through_both_branches(dim, density, seed, atlas):
    """The nearest thing to consuming the reference builder's output: a tree in `strict_reference` layout -- the fixed
    100 000-entry buffer filled from the end (include/map/Octree.h:29), all-ones page-header slots every 0x8000 entries
    and far pointers (src/map/Octree.cpp:251-315), including the builder's own far-pointer quirks -- rendered through
    the SVO branch and, with its dense twin, through the array branch; both equal the oracle and each other."""
    rng = np.random.default_rng(seed)
    grid = (rng.random(dim ** 3) < density).astype(np.int8) * 5
    if dim == 64:
        grid.reshape(dim, dim, dim)[dim // 2 - 2: dim // 2 + 2, :, :] = 0      # a corridor to look along
    o = vrc.Octree.Generate(grid, dim, buffer_size=100000, strict_reference=True)
    buf = o.descriptor_buffer
    assert buf.size == 100000
    far = int(((buf >> np.uint64(15)) & np.uint64(1))[buf != np.uint64(0xFFFFFFFFFFFFFFFF)].sum())
    headers = int((buf == np.uint64(0xFFFFFFFFFFFFFFFF)).sum())
    print(f"\nstrict tree {dim}^3: root at {o.root_index}, {int((buf != 0).sum())} non-zero slots, {far} far pointers, {headers} page headers")
    assert headers > 0 and (far > 0 or dim == 64)                            # the layout features this test is about
    # the oracle's builder produces the same array bit for bit (two independent implementations of Octree.cpp)
    obuf, oroot = orc.octree_generate(grid, dim)
    assert oroot == o.root_index and np.array_equal(obuf, buf)
    w, h, md = 160, 120, 3 * dim
    cam_pos, cam_dir = (dim / 2 + 0.31, 1.37, dim / 2 + 0.43), (1.45, 1.5708)
    li = np.zeros((1, 10), dtype=np.float32)
    li[0] = [0.01, 0.01, 0.01, 0.2, dim * 0.8, dim * 0.2, dim * 0.9, 0, 0, -1]
    frames = []
    for using_octree in (0, 1):
        c = make_caster(o, dim, using_octree, cam_dir, cam_pos, li, atlas, w, h, md, grid=grid)
        assert c.compute(), c.last_error()
        oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=c._li, atlas=atlas, tile_dim=(16, 16),
                                        descriptors=buf, root_index=o.root_index, octree_dim=dim, using_octree=using_octree,
                                        grid=grid, max_distance=md)
        img, hits = c.read_image(), c.read_hits()
        assert_same(img, hits, c.counters(), oimg, ohits, octr)
        frames.append((img, hits))
    assert (frames[0][1][..., 3] == 5).sum() > 1000
    assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32))
    assert np.array_equal(frames[0][1][..., :7], frames[1][1][..., :7])       # (the descriptor-read count differs by construction)


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
