#!/usr/bin/env python3
"""Long-running check that the SVO branch IS the reference's array branch on the grid the tree encodes ("SVO path
== array path"), at sizes where the oracle is slow: random dense maps of 128^3..512^3 voxels (noise, slabs, blobs,
materials 1 / 5 / 6 = mirrors), the tree built on the device from the same grid (vrc_build_dense_grid) with its
attachments, random cameras inside the map, 1-4 lights, step caps; the frame of the array kernel (reads the grid, the
branch the reference renders with) against the frame of the SVO kernel with closed-form jumps forced on -- image and hit
records bit for bit.  Not collected by pytest.  python tests/soak_array_vs_svo_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import voxel_raycaster_amd as vrc  # noqa: E402


def make_map(rng, depth):
    dim = 1 << depth
    kind = int(rng.integers(3))
    if kind == 0:                                          # sparse noise over a floor
        solid = rng.integers(0, 1000, (dim, dim, dim), dtype=np.int16) < int(rng.integers(1, 30))
        solid[: max(1, dim // 16)] = True
    elif kind == 1:                                        # slabs and pillars with long empty corridors between them
        solid = np.zeros((dim, dim, dim), dtype=bool)
        for _ in range(int(rng.integers(3, 12))):
            lo = rng.integers(0, dim, 3)
            hi = np.minimum(lo + rng.integers(1, dim // 3, 3), dim)
            solid[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
    else:                                                  # a hollow box with a few blobs inside
        solid = np.zeros((dim, dim, dim), dtype=bool)
        solid[[0, -1]] = True
        solid[:, [0, -1]] = True
        solid[:, :, [0, -1]] = True
        zz, yy, xx = np.ogrid[0:dim, 0:dim, 0:dim]
        for _ in range(int(rng.integers(1, 5))):
            cx, cy, cz = rng.random(3) * dim
            solid |= (xx - cx) ** 2 + (yy - cy) ** 2 + (zz - cz) ** 2 < (rng.random() * dim / 6) ** 2
    mats = rng.choice(np.array([5, 5, 5, 6, 1], dtype=np.int8), size=(dim, dim, dim))
    return np.where(solid, mats, 0).astype(np.int8)


def run(budget=300.0, seed=1, depths=(7, 8, 9), w=320, h=200, limit=None):     # limit: stop after this many maps (fixed volume)
    """Returns (frames that differ, frames, maps)."""
    rng = np.random.default_rng(seed)
    atlas = vrc.synthetic_atlas()
    t0, frames, bad, maps = time.time(), 0, 0, 0
    while time.time() - t0 < budget and (limit is None or maps < limit):
        depth = int(rng.choice(depths))
        dim = 1 << depth
        grid = make_map(rng, depth)
        cam = (np.zeros(2, dtype=np.float32), np.zeros(3, dtype=np.float32))
        li = np.zeros((8, 10), dtype=np.float32)
        casters = []
        for using_octree in (1, 0):                        # 1: the array kernel; 0: the SVO kernel (the reference's setting names)
            c = vrc.CLCaster()
            assert c.init(0)
            assert c.assign_map(grid, (dim, dim, dim))
            # one caster builds the tree from the grid it is handed, the other from the map that is already in its HBM
            # (and half of the maps get their material attachments from the device build, half from the host walk)
            on_device = maps % 2 == 0
            info = c.build_dense_grid(depth, grid if using_octree else None, validate_samples=1 << 16, attachments=on_device)
            assert info["validate_mismatches"] == 0
            if not on_device:
                tree = vrc.Octree(c.read_descriptors(), c.octree_size()[1], dim).attach_materials_from_grid(grid)
                assert c.assign_octree_attachments(tree), c.last_error()
            ok = (c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim) and c.add_to_settings_buffer("using_octree", "OCTENABLED", using_octree)
                  and c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", 3 * dim) and c.add_to_settings_buffer("light_count", "LIGHT_COUNT", 1)
                  and c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 16)
                  and c.assign_camera(*cam) and c.create_viewport(w, h) and c.assign_lights(li) and c.create_texture_atlas(atlas, (16, 16))
                  and c.validate())
            assert ok, c.last_error()
            casters.append(c)
        maps += 1
        empty = np.argwhere(grid == 0)
        for _ in range(40):
            z, y, x = empty[int(rng.integers(len(empty)))]
            cam[1][:] = (x + rng.random(), y + rng.random(), z + rng.random())
            cam[0][:] = (rng.random() * 3.1 + 0.02, rng.random() * 6.28)
            li[:, 0:4] = rng.random((8, 4)) * 0.8 + 0.2
            li[:, 4:7] = rng.random((8, 3)) * dim
            nl, md = int(rng.choice([1, 1, 2, 4])), int(rng.choice([3 * dim, 3 * dim, 300, 40]))
            out = []
            for c in casters:
                assert c.overwrite_setting("light_count", nl) and c.overwrite_setting("max_distance", md)
                assert c.compute(), c.last_error()
                out.append((c.read_image(), c.read_hits()))
            (ia, ha), (io, ho) = out
            same = np.array_equal(ia.view(np.uint32), io.view(np.uint32)) and np.array_equal(ha[..., :7], ho[..., :7])
            frames += 1
            if not same:
                bad += 1
                print("MISMATCH map", maps - 1, "depth", depth, "lights", nl, "max_distance", md, cam[1].tolist(), cam[0].tolist(),
                      int((ha[..., :7] != ho[..., :7]).any(-1).sum()), "hit records differ", flush=True)
        del casters
    print(f"array-vs-SVO soak: {frames} frames of {w}x{h} in {maps} random dense maps of depth {list(depths)} (device-built trees with attachments, "
          f"jumps forced on): {bad} differ between the array kernel and the SVO kernel; {time.time() - t0:.0f} s")
    return bad, frames, maps


if __name__ == "__main__":
    sys.exit(1 if run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)[0] else 0)
