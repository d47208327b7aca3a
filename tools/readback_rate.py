#!/usr/bin/env python3
"""PCIe-inclusive frame rate of the headline workload: compute() followed by the read-back a caller may ask for (the
boundary hands over host buffers only there).  Never bench.py's `value`; quoted in DESIGN.md section 7."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import voxel_raycaster_amd as vrc

sc = bench.build_scene(12)
w, h = 1920, 1080
c = bench.make_caster(sc, w, h, 0, hit_records=0)
for _ in range(40):
    assert c.compute()
rays = None


def timed(label, fn, frames=30):
    global rays
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(frames):
        fn()
    dt = (time.perf_counter() - t0) / frames
    if rays is None:
        ctr = c.counters()
        rays = ctr["primary_rays"] + ctr["shadow_rays"]
    print(json.dumps(dict(what=label, ms_per_frame=round(dt * 1e3, 3), Mrays_s=round(rays / dt / 1e6, 1))), flush=True)


rgba = np.empty((h, w, 4), dtype=np.uint8)
img = np.empty((h, w, 4), dtype=np.float32)
timed("compute only (resident; what bench.py's value measures)", lambda: c.compute())
timed("compute + RGBA8 frame to pageable host memory (8.3 MB)", lambda: (c.compute(), c.read_image_rgba8(out=rgba)))
timed("compute + float4 frame to pageable host memory (33 MB)", lambda: (c.compute(), c.read_image(out=img)))
if True:
    vrc.pin_host_buffer(rgba); vrc.pin_host_buffer(img)
    timed("compute + RGBA8 frame to pinned host memory", lambda: (c.compute(), c.read_image_rgba8(out=rgba)))
    timed("compute + float4 frame to pinned host memory", lambda: (c.compute(), c.read_image(out=img)))
