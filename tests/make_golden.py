#!/usr/bin/env python3
"""Writes tests/golden/orc_*.npz: small regression vectors (inputs + ORACLE outputs).

They are NOT reference outputs -- the reference cannot execute in this image or on the GPU box
(oracle/vrc_oracle.h, profiles/r01_reference_kernel_on_gfx950.txt).  Each vector stores explicit
float inputs (ray table, camera sin/cos) so a different libm cannot perturb replay.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orc  # noqa: E402
import scenes  # noqa: E402


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    atlas = scenes.hash_atlas()
    np.save(os.path.join(out, "atlas.npy"), atlas)
    jobs = [(scenes.app_default(), 64, 48, 1, 20), (scenes.app_default(), 64, 48, 0, 20),
            (scenes.floor_pillars(32), 64, 48, 1, 96), (scenes.floor_pillars(32), 64, 48, 0, 96),
            (scenes.mirror_wall(32), 64, 48, 1, 96), (scenes.open_sky(32), 64, 48, 0, 96),
            (scenes.axis_aligned(16), 64, 48, 1, 20), (scenes.random_sparse(64), 96, 64, 0, 192),
            # multi-light extension: 4 active lights
            (scenes.with_lights(scenes.floor_pillars(32), 4), 64, 48, 0, 96, 4),
            (scenes.with_lights(scenes.random_sparse(64), 4), 64, 48, 1, 192, 4)]
    for job in jobs:
        s, w, h, using, md = job[:5]
        nl = job[5] if len(job) > 5 else 1
        buf, root = orc.octree_generate(s["grid"], s["dim"])
        vp = orc.create_viewport(w, h)
        trig = orc.camera_trig(np.array(s["cam_dir"], dtype=np.float32))
        lights = np.zeros((8, 10), dtype=np.float32)
        lights[:nl] = s["lights"][:nl]
        img, hits, ctr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=lights,
                                     atlas=atlas, tile_dim=(16, 16), descriptors=buf, root_index=root,
                                     octree_dim=s["dim"], using_octree=using, grid=s["grid"], max_distance=md,
                                     viewport=vp, trig=trig, active_lights=nl)
        name = f"orc_{s['name']}_{'array' if using else 'svo'}{'_lights%d' % nl if nl > 1 else ''}.npz"
        np.savez_compressed(os.path.join(out, name), dim=s["dim"], width=w, height=h, using_octree=using,
                            max_distance=md, grid=np.asarray(s["grid"], dtype=np.int8), descriptors_tail=buf[root:],
                            root_index=root, buffer_size=buf.size, viewport=vp,
                            cam_dir=np.array(s["cam_dir"], dtype=np.float32),
                            cam_pos=np.array(s["cam_pos"], dtype=np.float32), cam_trig=trig, lights=lights,
                            image=img, hits=hits, counters=json.dumps(ctr), active_lights=nl)
        print(name, ctr)


if __name__ == "__main__":
    main()
