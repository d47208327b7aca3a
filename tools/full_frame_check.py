#!/usr/bin/env python3
"""The headline frame, every pixel, against the oracle (checker role; ~10 s of host time on 16 threads): image bits, hit records.
    python tools/full_frame_check.py [--survey]   (VRC_LIB_PATH selects a library variant)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import orc

ap = argparse.ArgumentParser()
ap.add_argument("--survey", action="store_true", help="SURVEY 8d camera as written (bias active)")
ap.add_argument("--hit-records", type=int, default=1)
a = ap.parse_args()
sc = bench.build_scene(12)
pos = sc["survey_cam_pos"] if a.survey else sc["cam_pos"]
sc = dict(sc, cam_pos=np.ascontiguousarray(pos, dtype=np.float32))
c = bench.make_caster(sc, 1920, 1080, 0, hit_records=a.hit_records)
for _ in range(3):
    assert c.compute(), c.last_error()
img = c.read_image()
hits = c.read_hits() if a.hit_records else None
oimg, ohits, octr = orc.raycast(width=1920, height=1080, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"], lights=sc["lights"], atlas=sc["atlas"],
                                tile_dim=(16, 16), descriptors=sc["octree"].descriptor_buffer, root_index=sc["octree"].root_index,
                                octree_dim=sc["dim"], using_octree=0, max_distance=3 * sc["dim"], threads=bench.usable_cores(), want_hits=True)
di = (img.view(np.uint32) != oimg.view(np.uint32)).any(-1)
print("camera", [float(v) for v in sc["cam_pos"]], "image pixels differing:", int(di.sum()))
if hits is not None:
    k = 7 if c.used_empty_boxes() else 8     # with the empty boxes field 7 is the box traversal's read count, not the canonical one
    dh = (hits[..., :k] != ohits[..., :k]).any(-1)
    print("empty boxes:", c.used_empty_boxes(), "hit records differing:", int(dh.sum()), "by field:", [int((hits[..., j] != ohits[..., j]).sum()) for j in range(k)])
    di = di | dh
ys, xs = np.nonzero(di)
for y, x in list(zip(ys, xs))[:8]:
    print(" pixel", (int(x), int(y)), "gpu", img[y, x], None if hits is None else hits[y, x], "oracle", oimg[y, x], ohits[y, x])
print("counters gpu", c.counters())
print("counters orc", octr)
