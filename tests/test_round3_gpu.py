"""GPU suite (-m gpu), round 3: slices of the two soaks inside the suite, the reference builder's own buffer layout
through both branches, the exact closed-form jumps on every kind of frame."""
import numpy as np
import pytest

import scenes
import voxel_raycaster_amd as vrc
from oracle import orc
from test_parity_gpu import assert_same, hits_match, make_caster

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ soak slices (fixed seeds AND fixed volumes: the driver runs them)
# (until round 5 these were time-budgeted -- what they covered depended on the box's speed; now every slice
# renders a fixed number of poses / frames / maps, ~10 s each on an MI355X, and the time budget is only a safety net)
def test_soak_slice_product_against_the_oracle():
    """400 poses of tests/soak_gpu.py with a fixed seed: random camera poses / light sets / step caps in the depth-12
    bench scene, exact mode (closed-form jumps on, the default at this depth) and mode B, single handles and 3-rank
    groups, device RGBA8 pack -- whole 8-row tile bands against the oracle, bit for bit."""
    import soak_gpu
    bad, poses, rows = soak_gpu.run(budget=120.0, seed=20261002, depth=12, limit=400)
    assert poses == 400 and rows == 400 * 24
    assert bad == 0


def test_soak_slice_reference_kernel_against_the_oracle():
    """3000 frames of tests/soak_reference_gpu.py with a fixed seed: the reference's own raycaster kernel on the MI355X
    (two image builtins redirected) against the oracle on random poses in the probe scenes."""
    import os
    import test_reference_pin_gpu as pin
    if not os.path.exists(os.path.join(pin.REF, "ref_raycaster_gfx950_strict.co")):
        pytest.fail("oracle/_ref/ is missing: build it with `make -C oracle _ref` where /root/reference exists")
    import soak_reference_gpu
    failures, frames, totals = soak_reference_gpu.run(budget=120.0, seed=20261002, limit=3000)
    n = max(totals.get("shaded", 0), 1)
    print(f"\nreference soak slice: {frames} frames, shaded pixels {totals.get('shaded', 0)}, rgb within 1e-5 "
          f"{totals.get('rgb_1e-5', 0) / n:.6f}, within 1e-4 {totals.get('rgb_1e-4', 0) / n:.6f}")
    assert frames == 3000 and failures == 0
    assert totals.get("rgb_1e-5", 0) / n >= 0.999


def test_soak_slice_exact_jumps_on_small_frames_of_deep_scenes():
    """1500 frames of tests/soak_jumps_gpu.py with a fixed seed: 640x360 frames (900 blocks: fewer than the chip holds,
    the case in which blocks once handed their jump-table slots across XCDs and 1 frame in 4000 came back with a few
    iteration counts off by one) of device-built depth-10 / 12 / 14 / 16 terrains, random poses, 1-4 lights, step caps and
    jump thresholds, through the empty boxes where the tree has them (depths 10, 12) -- image, hit records and every counter equal
    to the same frame stepped voxel by voxel from octree node to octree node."""
    import soak_jumps_gpu
    bad, frames, steps = soak_jumps_gpu.run(budget=150.0, seed=20261002, depths=(10, 12, 14, 16), limit=1500)
    assert frames == 1500 and steps > 1e10
    assert bad == 0


def test_soak_slice_group_handle_against_the_single_handle():
    """150 frames of tests/soak_groups_gpu.py with a fixed seed: 1-8 ranks on this GPU behind one handle (with and without
    own copies of the tree), bands of 8..128 rows, frame sizes from 1x1 to 1920x1080 that are multiples of nothing, both
    stepping modes, 1-4 lights, attachments, pinned and pageable read-back: frame, hit records, RGBA8 and counters equal
    the single handle's."""
    import soak_groups_gpu
    bad, frames = soak_groups_gpu.run(budget=120.0, seed=20261002, depths=(8, 10), limit=150)
    assert frames == 150 and bad == 0


def test_soak_slice_device_builder_against_the_host_emitter():
    """100 fields and grids of tests/soak_builder_gpu.py with a fixed seed: column fields the fixtures do not hold (white noise,
    slabs, cliffs, floating pillars, single layers, solid maps, ceilings) at depths 6-9: the device-built array equals the
    host emitter's bit for bit, the device validate passes, point queries of the tree agree with the field."""
    import soak_builder_gpu
    bad, fields, descriptors = soak_builder_gpu.run(budget=120.0, seed=20261002, depths=(6, 7, 8, 9), limit=100)
    assert fields == 100 and descriptors > 0 and bad == 0


def test_soak_slice_array_kernel_against_svo_kernel_on_device_built_trees():
    """30 maps (1200 frames) of tests/soak_array_vs_svo_gpu.py with a fixed seed: random dense maps of 128^3 / 256^3 voxels with
    materials and mirrors, the tree built on the device from the same grid (vrc_build_dense_grid + attachments), random
    cameras inside the map: the array kernel's frame and the SVO kernel's frame (closed-form jumps forced on) are the
    same image and the same hit records -- "SVO path == array path" beyond the sizes the oracle follows."""
    import soak_array_vs_svo_gpu
    bad, frames, maps = soak_array_vs_svo_gpu.run(budget=120.0, seed=20261002, depths=(7, 8), limit=30)
    assert maps == 30 and frames >= 30 * 40 and bad == 0


# ------------------------------------------------------------------ the array a reference host would pass
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


# ------------------------------------------------------------------ exact closed-form jumps: identical frames
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
    # (the two times are a printed figure, not an assertion: 4 frames on a shared or clock-ramping GPU are no measurement;
    # that the jump instance ran is the launch's choice by setting -- tools/jump_ab.py --stats counts its passes)


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


# ------------------------------------------------------------------ device builder for any column scene (SURVEY 8f-4 at scale)
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
