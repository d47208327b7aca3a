#!/usr/bin/env python3
"""How much of the headline kernel's time is ramp-up / tail?  Two casters (two streams) rendering the same frame
concurrently vs one after the other: python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sc = bench.build_scene(12)
cs = [bench.make_caster(sc, 1920, 1080, 0) for _ in range(3)]
for c in cs:
    for _ in range(3): assert c.compute()
def run(k, frames=20):
    t = time.time()
    for _ in range(frames):
        for c in cs[:k]: assert c.compute_async()
        for c in cs[:k]: assert c.sync()
    dt = time.time() - t
    return dt / (frames * k) * 1e3
for k in (1, 2, 3, 1, 2, 3):
    print(f"{k} frames in flight: {run(k):.3f} ms per frame", flush=True)
