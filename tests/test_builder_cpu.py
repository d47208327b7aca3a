"""CPU suite: the host side of the large-scene builder (procedural columns, the page-header-free "brick" layout the
device builder emits) and the oracle's paged descriptor source."""
import numpy as np
import pytest

import scenes
import treetools
import voxel_raycaster_amd as vrc
from oracle import orc


@pytest.mark.parametrize("depth,thickness,floor", [(6, 2, 2), (7, 5, 0), (7, 0, 1)])
def test_procedural_column_equals_the_height_table(depth, thickness, floor):
    dim = 1 << depth
    _, height = vrc.shell_terrain_ex(depth, seed=3, thickness=thickness, octave_floor=floor, want_height=True)
    rng = np.random.default_rng(depth)
    cols = np.concatenate([rng.integers(0, dim, size=(300, 2)), [[0, 0], [dim - 1, dim - 1], [0, dim - 1], [dim - 1, 0]]])
    for x, y in cols:
        lo, hi = vrc.shell_column(depth, x, y, seed=3, thickness=thickness, octave_floor=floor)
        assert hi == height[y, x]
        nb = [height[y, x]] + [height[yy, xx] for xx, yy in ((x - 1, y), (x + 1, y), (x, y - 1), (x, y + 1))
                               if 0 <= xx < dim and 0 <= yy < dim]
        assert lo == max(0, min(nb) - thickness)


@pytest.mark.parametrize("depth,thickness", [(6, 2), (7, 2), (8, 6)])
def test_brick_layout_is_the_same_tree_without_page_headers(depth, thickness):
    dim = 1 << depth
    paged, _ = vrc.shell_terrain_ex(depth, thickness=thickness, layout=0)
    brick, _ = vrc.shell_terrain_ex(depth, thickness=thickness, layout=vrc.LAYOUT_NO_PAGE_HEADERS)
    a, (na, _) = treetools.canonical(paged.descriptor_buffer, paged.root_index, dim)
    b, (nb, _) = treetools.canonical(brick.descriptor_buffer, brick.root_index, dim)
    assert a == b and na == nb
    all_ones = np.uint64(0xFFFFFFFFFFFFFFFF)
    assert not (brick.descriptor_buffer == all_ones).any()
    if paged.descriptor_buffer.size > 0x8000:
        assert (paged.descriptor_buffer == all_ones).any()          # the paged layout has its header(s)
        assert brick.descriptor_buffer.size < paged.descriptor_buffer.size
    # every slot of the brick layout is a node or a far-pointer slot: nothing is wasted
    far = int(((brick.descriptor_buffer >> np.uint64(15)) & np.uint64(1)).sum())
    assert brick.descriptor_buffer.size >= nb and brick.descriptor_buffer.size <= nb + far + 8
    if depth <= 7:
        grid = np.zeros(dim ** 3, dtype=np.int8)
        for y in range(dim):
            for x in range(dim):
                lo, hi = vrc.shell_column(depth, x, y, thickness=thickness)
                grid.reshape(dim, dim, dim)[lo:hi + 1, y, x] = 5
        assert orc.octree_validate(grid, dim, brick.descriptor_buffer, brick.root_index) == 0


def test_oracle_paged_descriptor_source_equals_the_flat_array(atlas):
    depth, dim = 7, 128
    tree, height = vrc.shell_terrain(depth, seed=1, thickness=2)
    desc = tree.descriptor_buffer
    reads = []

    def read(first, count):
        reads.append((first, count))
        return desc[first:first + count]

    paged = orc.PagedDescriptors(desc.size, read)
    cam_pos = (dim / 2 + 0.37, dim / 8 + 0.41, float(height[dim // 8, dim // 2]) + 9.29)
    kw = dict(width=96, height=64, cam_dir=(2.0, 1.5708), cam_pos=cam_pos, lights=scenes.floor_pillars()["lights"], atlas=atlas,
              tile_dim=(16, 16), root_index=tree.root_index, octree_dim=dim, using_octree=0, max_distance=3 * dim)
    img_a, hits_a, ctr_a = orc.raycast(descriptors=desc, **kw)
    img_b, hits_b, ctr_b = orc.raycast(descriptors=paged, threads=4, **kw)
    assert np.array_equal(img_a.view(np.uint32), img_b.view(np.uint32)) and np.array_equal(hits_a, hits_b) and ctr_a == ctr_b
    assert len(reads) == len(set(reads)) and 0 < paged.bytes_fetched <= (desc.size + orc.PAGE_SIZE) * 8


def test_host_builders_under_sanitizers(tmp_path):
    """csrc/svo_builder.cpp (builders, scene functions, file format) compiled with g++ -fsanitize=address,undefined and
    driven through the C ABI: no report, no leak (GPU sanitizers are not available on the pool; this is the host side)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "host_sanitize.cpp"),
                           os.path.join(root, "voxel-raycaster_amd", "csrc", "svo_builder.cpp"), "-o", exe])
    out = subprocess.run([exe, str(tmp_path / "t.svo")], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0 and "host sanitize ok" in out.stdout, out.stdout + out.stderr


def test_hand_laid_sparse_tree_encodes_the_same_octree_as_the_builder():
    """tests/treetools.sparse_octree (the deepest-tree GPU tests lay their 24-level trees with it) against the product's
    builder on grids it can build: same octree, whatever the storage order."""
    import treetools
    rng = np.random.default_rng(11)
    for depth, p in ((3, 0.2), (4, 0.05), (5, 0.01)):
        dim = 1 << depth
        g = (rng.random((dim, dim, dim)) < p).astype(np.int8) * 5
        vox = [(x, y, z) for z in range(dim) for y in range(dim) for x in range(dim) if g[z, y, x]]
        d, r = treetools.sparse_octree(vox, depth)
        o = vrc.Octree.Generate(g.reshape(-1), dim, buffer_size=0, strict_reference=False)
        assert treetools.canonical(d, r, dim)[0] == treetools.canonical(o.descriptor_buffer, o.root_index, dim)[0]


@pytest.mark.parametrize("dim,density", [(8, 0.3), (32, 0.05), (64, 0.5), (128, 0.002)])
def test_dense_grid_layouts_encode_the_same_octree(dim, density):
    """vrc_octree_generate_ex (the host twin of the device dense-grid builder): layout 0 is vrc_octree_generate's array,
    layout VRC_LAYOUT_NO_PAGE_HEADERS holds the same tree -- every voxel of the grid answers the same point query
    (Octree::GetVoxel, src/map/Octree.cpp:45-158) -- and empty / solid maps come out as the one-descriptor trees they are."""
    rng = np.random.default_rng(dim)
    g = (rng.random(dim ** 3) < density).astype(np.int8) * 5
    paged = vrc.Octree.Generate(g, dim, strict_reference=False)
    same = vrc.Octree.Generate(g, dim, layout=0)
    brick = vrc.Octree.Generate(g, dim, layout=vrc.LAYOUT_NO_PAGE_HEADERS)
    assert np.array_equal(paged.descriptor_buffer, same.descriptor_buffer) and paged.root_index == same.root_index
    assert brick.descriptor_buffer.size <= paged.descriptor_buffer.size
    g3 = g.reshape(dim, dim, dim)
    pts = rng.integers(0, dim, (4000, 3)) if dim > 16 else np.argwhere(np.ones((dim, dim, dim), dtype=bool))
    for z, y, x in pts:
        want = bool(g3[z, y, x])
        assert brick.GetVoxel((int(x), int(y), int(z)))[0] == want and paged.GetVoxel((int(x), int(y), int(z)))[0] == want
    for fill in (0, 5):
        t = vrc.Octree.Generate(np.full(dim ** 3, fill, dtype=np.int8), dim, layout=vrc.LAYOUT_NO_PAGE_HEADERS)
        assert t.GetVoxel((dim // 2, 1, dim - 1))[0] == bool(fill)
        if not fill:
            assert t.descriptor_buffer.size == 1

