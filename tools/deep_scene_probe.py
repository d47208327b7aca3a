#!/usr/bin/env python3
"""Kernel time of the shell-terrain scene at a given depth on one MI355X:
python tools/deep_scene_probe.py <depth> [width height lights]   (depth 16 needs ~165 GB host RSS and 3.5 min to build)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
D = int(sys.argv[1]) if len(sys.argv) > 1 else 14
W, H, L = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 1)
t = time.time(); sc = bench.build_scene(D)
print("build s", round(time.time() - t, 1), "descriptors", sc["octree"].descriptor_buffer.size, flush=True)
c = bench.make_caster(sc, W, H, 0, light_count=L)
for _ in range(2): assert c.compute(), c.last_error()
c.timing_reset()
for _ in range(3): assert c.compute()
n, ms = c.timing(); ctr = c.counters()
rays = ctr["primary_rays"] + ctr["shadow_rays"]
print(json.dumps({"depth": D, "frame": f"{W}x{H}", "lights": L, "kernel_ms": round(ms / n, 3), "Mrays/s": round(rays / (ms / n) / 1e3, 1),
                  "rays": rays, "steps": ctr["steps"], "descriptor_reads": ctr["descriptor_reads"]}))
