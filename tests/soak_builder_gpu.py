#!/usr/bin/env python3
"""Long-running check of the device SVO builders (csrc/svo_builder_gpu.hip) on inputs the test suite does not hold.
vrc_build_heightfield: column fields of white noise, slabs, cliffs, pillars over bare ground, one-voxel and solid maps,
depths 6..11; vrc_build_dense_grid: dense grids of noise at densities from a few voxels to nearly solid, blobs, planes,
empty and full maps, depths 3..8.  The array built in HBM against the sequential host emitter (vrc_octree_from_columns /
vrc_octree_generate_ex) bit for bit, the device-side Octree::Validate, and point queries of the finished tree
(Octree::GetVoxel, src/map/Octree.cpp:45-158) against the input itself.
Not collected by pytest.  python tests/soak_builder_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import voxel_raycaster_amd as vrc  # noqa: E402


def field(rng, depth):
    dim = 1 << depth
    kind = int(rng.integers(10))
    yy, xx = np.mgrid[0:dim, 0:dim]
    if kind == 0:                                          # white noise, solid from the ground
        hi, lo = rng.integers(0, dim, (dim, dim)), None
    elif kind == 1:                                        # white noise slabs
        a, b = rng.integers(0, dim, (dim, dim)), rng.integers(0, dim, (dim, dim))
        hi, lo = np.maximum(a, b), np.minimum(a, b)
    elif kind == 2:                                        # a thin shell over waves (every column holds a voxel: lo <= hi is the API's contract)
        hi = (dim / 2 + dim / 3 * np.sin(xx * rng.random() * 0.2) * np.cos(yy * rng.random() * 0.2)).astype(np.int64)
        lo = np.maximum(hi - rng.integers(0, 4, (dim, dim)), 0)
    elif kind == 3:                                        # cliffs: blocks of constant height
        s = 1 << int(rng.integers(1, depth))
        hi = np.kron(rng.integers(0, dim, (dim // s, dim // s)), np.ones((s, s), dtype=np.int64))
        lo = None if rng.random() < 0.5 else np.maximum(hi - int(rng.integers(1, dim)), 0)
    elif kind == 4:                                        # sparse pillars (floating, too) over one layer of ground
        hi, lo = np.zeros((dim, dim), dtype=np.int64), np.zeros((dim, dim), dtype=np.int64)
        n = int(rng.integers(1, 200))
        px, py = rng.integers(0, dim, n), rng.integers(0, dim, n)
        top = rng.integers(0, dim, n)
        hi[py, px] = top
        lo[py, px] = np.minimum(rng.integers(0, dim, n), top)
    elif kind == 5:                                        # one layer somewhere, nothing else
        hi = np.full((dim, dim), int(rng.integers(0, dim)), dtype=np.int64)
        lo = hi.copy()
    elif kind == 6:                                        # a solid map / a solid slab
        hi = np.full((dim, dim), dim - 1, dtype=np.int64)
        lo = None if rng.random() < 0.5 else np.full((dim, dim), int(rng.integers(0, dim)), dtype=np.int64)
    elif kind == 7:                                        # the ground layer and one voxel above it
        hi, lo = np.zeros((dim, dim), dtype=np.int64), np.zeros((dim, dim), dtype=np.int64)
        x, y, z = (int(v) for v in rng.integers(0, dim, 3))
        hi[y, x] = lo[y, x] = z
    elif kind == 8:                                        # a ceiling: everything occupied near the top only
        hi = np.full((dim, dim), dim - 1, dtype=np.int64)
        lo = dim - 1 - rng.integers(0, 3, (dim, dim))
    else:                                                  # a ramp with noise
        hi = np.clip((xx + yy) // 2 + rng.integers(-2, 3, (dim, dim)), 0, dim - 1)
        lo = None
    hi = np.clip(hi, 0, dim - 1).astype(np.uint16)
    lo = None if lo is None else np.clip(lo, 0, dim - 1).astype(np.uint16)
    return kind, hi, lo


def grid(rng, depth):
    dim = 1 << depth
    kind = int(rng.integers(7))
    if kind == 0:                                          # noise at any density
        g = rng.random(dim ** 3) < 10.0 ** (-4.0 * rng.random())
    elif kind == 1:                                        # a handful of voxels
        g = np.zeros(dim ** 3, dtype=bool)
        g[rng.integers(0, dim ** 3, int(rng.integers(1, 20)))] = True
    elif kind == 2:                                        # nearly solid: a handful of holes
        g = np.ones(dim ** 3, dtype=bool)
        g[rng.integers(0, dim ** 3, int(rng.integers(0, 20)))] = False
    elif kind == 3:                                        # blobs
        zz, yy, xx = np.mgrid[0:dim, 0:dim, 0:dim]
        g = np.zeros((dim, dim, dim), dtype=bool)
        for _ in range(int(rng.integers(1, 6))):
            cx, cy, cz = rng.random(3) * dim
            g |= (xx - cx) ** 2 + (yy - cy) ** 2 + (zz - cz) ** 2 < (rng.random() * dim / 3) ** 2
        g = g.reshape(-1)
    elif kind == 4:                                        # axis-aligned planes and lines
        g = np.zeros((dim, dim, dim), dtype=bool)
        for _ in range(int(rng.integers(1, 5))):
            a = int(rng.integers(3))
            idx = [slice(None)] * 3
            idx[a] = int(rng.integers(dim))
            if rng.random() < 0.5:
                idx[(a + 1) % 3] = int(rng.integers(dim))
            g[tuple(idx)] = True
        g = g.reshape(-1)
    elif kind == 5:
        g = np.zeros(dim ** 3, dtype=bool)
    else:
        g = np.ones(dim ** 3, dtype=bool)
    return kind, (g.astype(np.int8) * int(rng.choice([1, 5, 6, -1])))


def run_grid(rng):
    depth = int(rng.integers(3, 9))
    dim = 1 << depth
    kind, g = grid(rng, depth)
    host = vrc.Octree.Generate(g, dim, layout=2)
    c = vrc.CLCaster()
    assert c.init(0)
    info = c.build_dense_grid(depth, g, validate_samples=1 << 18)
    cnt, root = c.octree_size()
    dev = c.read_descriptors()
    same = (cnt == host.descriptor_buffer.size and root == host.root_index and np.array_equal(dev, host.descriptor_buffer)
            and info["validate_mismatches"] == 0)
    tree = vrc.Octree(dev, root, dim)
    g3 = g.reshape(dim, dim, dim)
    solid = np.argwhere(g3 != 0)
    for i in range(200):
        if i % 2 and len(solid):
            z, y, x = (int(v) for v in solid[int(rng.integers(len(solid)))])
            x = min(max(x + int(rng.integers(-1, 2)), 0), dim - 1)
        else:
            x, y, z = (int(v) for v in rng.integers(0, dim, 3))
        if tree.GetVoxel((x, y, z))[0] != bool(g3[z, y, x]):
            same = False
            print("  GetVoxel", (x, y, z), "expected", bool(g3[z, y, x]), flush=True)
            break
    if not same:
        print("MISMATCH grid depth", depth, "kind", kind, "descriptors host", host.descriptor_buffer.size, "device", cnt, "roots",
              host.root_index, root, "validate", info["validate_mismatches"], flush=True)
    return same, int(cnt)


def run(budget=300.0, seed=1, depths=(6, 7, 8, 9, 10, 11), limit=None):     # limit: stop after this many fields + grids (fixed volume)
    """Returns (fields with a difference, fields, descriptors compared)."""
    rng = np.random.default_rng(seed)
    t0, n, bad, total = time.time(), 0, 0, 0
    grids = 0
    while time.time() - t0 < budget and (limit is None or n < limit):
        if rng.random() < 0.4:
            same, cnt = run_grid(rng)
            n += 1
            grids += 1
            total += cnt
            bad += 0 if same else 1
            continue
        depth = int(rng.choice(depths))
        dim = 1 << depth
        kind, hi, lo = field(rng, depth)
        host = vrc.octree_from_columns(depth, hi, lo, layout=2)
        c = vrc.CLCaster()
        assert c.init(0)
        info = c.build_heightfield(depth, hi, lo, validate_samples=1 << 18)
        cnt, root = c.octree_size()
        dev = c.read_descriptors()
        same = (cnt == host.descriptor_buffer.size and root == host.root_index and np.array_equal(dev, host.descriptor_buffer)
                and info["validate_mismatches"] == 0)
        # the finished tree against the field: random voxels, half of them right at a column's top or bottom
        tree = vrc.Octree(dev, root, dim)
        lo_eff = np.zeros_like(hi) if lo is None else lo
        for _ in range(200):
            x, y = int(rng.integers(dim)), int(rng.integers(dim))
            z = int(rng.integers(dim)) if rng.random() < 0.5 else int(np.clip(int(rng.choice([hi[y, x], lo_eff[y, x]])) + int(rng.integers(-1, 2)), 0, dim - 1))
            want = int(lo_eff[y, x]) <= z <= int(hi[y, x])
            if tree.GetVoxel((x, y, z))[0] != want:
                same = False
                print("  GetVoxel", (x, y, z), "expected", want, flush=True)
                break
        n += 1
        total += int(cnt)
        if not same:
            bad += 1
            print("MISMATCH field", n - 1, "depth", depth, "kind", kind, "descriptors host", host.descriptor_buffer.size, "device", cnt,
                  "roots", host.root_index, root, "validate", info["validate_mismatches"], flush=True)
        del c
    print(f"builder soak: {n - grids} column fields of depth {list(depths)} (noise, slabs, thin shells, cliffs, pillars, single layers, solid, one voxel, "
          f"ceilings, ramps) and {grids} dense grids of depth 3..8 (noise at any density, single voxels, holes, blobs, planes, empty, solid): {bad} differ from the host emitter / the device validate / point queries; {total / 1e6:.1f} M descriptors; "
          f"{time.time() - t0:.0f} s")
    return bad, n, total


if __name__ == "__main__":
    sys.exit(1 if run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)[0] else 0)
