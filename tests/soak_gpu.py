#!/usr/bin/env python3
"""Long-running parity soak (not collected by pytest): random camera poses / lights / step caps in the depth-12 bench
scene, GPU vs oracle, bit for bit, in both stepping modes (the exact kernel against the reference restatement, the
node-exit jump kernel against its own restatement), with hit records, row slices of a 3-rank group handle and the
device-side RGBA8 pack in the loop.  python tests/soak_gpu.py [seconds] [seed] [depth]"""
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
from test_parity_gpu import hits_match, make_caster  # noqa: E402


def run(budget=300.0, seed=1, depth=12, sc=None, limit=None):     # limit: stop after this many poses (fixed volume; the budget is then a safety net)
    """Returns (mismatching bands, poses, rows compared).  tests/test_soak_slices_gpu.py runs a 10-second slice of it."""
    rng = np.random.default_rng(seed)
    sc = sc or bench.build_scene(depth)
    dim, w, h = sc["dim"], 512, 288
    t0, poses, rows, bad, by_mode = time.time(), 0, 0, 0, [0, 0]
    # the tree, its coarse table and its boxes are uploaded / built ONCE and kept by this holder; every pose's caster adopts them
    holder = vrc.CLCaster()
    assert holder.init(0) and holder.assign_octree(sc["octree"]), holder.last_error()
    while time.time() - t0 < budget and (limit is None or poses < limit):
        cam_pos = tuple(float(v) for v in (rng.random(3) * (dim * 1.2) - 0.1 * dim))
        if rng.random() < 0.5:                       # mostly above the terrain
            cx, cy = int(min(max(cam_pos[0], 0), dim - 1)), int(min(max(cam_pos[1], 0), dim - 1))
            cam_pos = (cam_pos[0], cam_pos[1], float(sc["height"][cy, cx]) + float(rng.random() * dim * 0.3) + 1.5)
        cam_dir = (float(rng.random() * 3.1 + 0.02), float(rng.random() * 6.28))
        nl = int(rng.choice([1, 1, 2, 4]))
        lights = sc["lights"].copy()
        lights[:, 4:7] = rng.random((8, 3)) * dim * 1.1
        md = int(rng.choice([3 * dim, 3 * dim, 700, 5000]))
        mode = int(rng.integers(0, 2))
        grouped = rng.random() < 0.25
        if grouped:                                  # three ranks on GPU 0 behind one handle: row slices, gathered read-back
            c = vrc.CLCaster()
            assert c.init_group([0, 0, 0], band_rows=8)
            cd, cp = np.array(cam_dir, dtype=np.float32), np.array(cam_pos, dtype=np.float32)
            ok = (c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim) and c.add_to_settings_buffer("using_octree", "OCTENABLED", 0)
                  and c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", md) and c.add_to_settings_buffer("light_count", "LIGHT_COUNT", nl)
                  and c.assign_octree_from(holder) and c.assign_camera(cd, cp) and c.create_viewport(w, h) and c.assign_lights(lights)
                  and c.create_texture_atlas(sc["atlas"], (16, 16)) and c.validate())
            assert ok, c.last_error()
            c._li = lights
        else:
            c = make_caster(sc["octree"], dim, 0, cam_dir, cam_pos, lights, sc["atlas"], w, h, md, light_count=nl, tree_from=holder)
        assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", mode)
        assert c.compute(), c.last_error()
        img, hits, rgba = c.read_image(), c.read_hits(), c.read_image_rgba8()
        if not np.array_equal(rgba, orc.image_to_rgba8(img)):
            bad += 1
            print("RGBA8 MISMATCH", cam_pos, cam_dir, flush=True)
        for y0 in rng.choice(h // 8, size=3, replace=False) * 8:
            oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=c._li, atlas=sc["atlas"],
                                         tile_dim=(16, 16), descriptors=sc["octree"].descriptor_buffer,
                                         root_index=sc["octree"].root_index, octree_dim=dim, using_octree=0, max_distance=md,
                                         rows=(int(y0), int(y0) + 8), threads=16, active_lights=nl, stepping_mode=mode)
            same = hits_match(c, hits[y0:y0 + 8], ohits[y0:y0 + 8]) and \
                np.array_equal(img[y0:y0 + 8].view(np.uint32), oimg[y0:y0 + 8].view(np.uint32))
            rows += 8
            if not same:
                bad += 1
                print("MISMATCH", "mode", mode, "grouped", grouped, cam_pos, cam_dir, nl, md, int(y0), flush=True)
        poses += 1
        by_mode[mode] += 1
        del c
    print(f"soak: {poses} poses ({by_mode[0]} exact, {by_mode[1]} mode B), {rows} rows of {w} pixels compared, {bad} mismatching bands, "
          f"{time.time() - t0:.0f} s")
    return bad, poses, rows


if __name__ == "__main__":
    sys.exit(1 if run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                      int(sys.argv[3]) if len(sys.argv) > 3 else 12)[0] else 0)
