#!/usr/bin/env python3
"""Long-running parity soak (not collected by pytest): random camera poses / lights / step caps in the depth-12 bench
scene and random small scenes, GPU vs oracle, bit for bit.  python tests/soak_gpu.py [seconds] [seed] [depth]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import voxel_raycaster_amd as vrc  # noqa: E402
from oracle import orc  # noqa: E402
from test_parity_gpu import make_caster  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sc = bench.build_scene(int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    dim, w, h = sc["dim"], 512, 288
    t0, poses, rows, bad = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        cam_pos = tuple(float(v) for v in (rng.random(3) * (dim * 1.2) - 0.1 * dim))
        if rng.random() < 0.5:                       # mostly above the terrain
            cx, cy = int(min(max(cam_pos[0], 0), dim - 1)), int(min(max(cam_pos[1], 0), dim - 1))
            cam_pos = (cam_pos[0], cam_pos[1], float(sc["height"][cy, cx]) + float(rng.random() * dim * 0.3) + 1.5)
        cam_dir = (float(rng.random() * 3.1 + 0.02), float(rng.random() * 6.28))
        nl = int(rng.choice([1, 1, 2, 4]))
        lights = sc["lights"].copy()
        lights[:, 4:7] = rng.random((8, 3)) * dim * 1.1
        md = int(rng.choice([3 * dim, 3 * dim, 700, 5000]))
        c = make_caster(sc["octree"], dim, 0, cam_dir, cam_pos, lights, sc["atlas"], w, h, md, light_count=nl)
        assert c.compute(), c.last_error()
        img, hits = c.read_image(), c.read_hits()
        for y0 in rng.choice(h // 8, size=3, replace=False) * 8:
            oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=c._li, atlas=sc["atlas"],
                                         tile_dim=(16, 16), descriptors=sc["octree"].descriptor_buffer,
                                         root_index=sc["octree"].root_index, octree_dim=dim, using_octree=0, max_distance=md,
                                         rows=(int(y0), int(y0) + 8), threads=16, active_lights=nl)
            same = np.array_equal(hits[y0:y0 + 8], ohits[y0:y0 + 8]) and \
                np.array_equal(img[y0:y0 + 8].view(np.uint32), oimg[y0:y0 + 8].view(np.uint32))
            rows += 8
            if not same:
                bad += 1
                print("MISMATCH", cam_pos, cam_dir, nl, md, int(y0), flush=True)
        poses += 1
        del c
    print(f"soak: {poses} poses, {rows} rows of {w} pixels compared, {bad} mismatching bands, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
