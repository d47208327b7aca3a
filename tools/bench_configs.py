#!/usr/bin/env python3
"""One-off table for DESIGN.md: the other BASELINE configs on one MI355X (kernel ms, Mrays/s)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def run(name, depth, w, h, shadow, lights=1):
    sc = bench.build_scene(depth)
    c = bench.make_caster(sc, w, h, 0, shadow_rays=shadow, light_count=lights)
    for _ in range(3): assert c.compute(), c.last_error()
    c.timing_reset()
    for _ in range(10): assert c.compute()
    n, ms = c.timing(); ctr = c.counters()
    rays = ctr["primary_rays"] + ctr["shadow_rays"]
    print(json.dumps({"config": name, "depth": depth, "frame": f"{w}x{h}", "shadow_rays": shadow, "lights": lights, "descriptors": int(sc["octree"].descriptor_buffer.size),
                      "kernel_ms": round(ms / n, 3), "Mrays/s": round(rays / (ms / n) / 1e3, 1), "rays": rays, "steps": ctr["steps"],
                      "descriptor_reads": ctr["descriptor_reads"]}), flush=True)

run("C1 geometry (d8, 640x480, primary only) on the GPU", 8, 640, 480, 0)
run("C2 (d10, 1080p, primary only)", 10, 1920, 1080, 0)
run("C3 headline (d12, 1080p, primary+shadow)", 12, 1920, 1080, 1)
run("C4 geometry (d12, 4K, 1 light) on 1 GPU", 12, 3840, 2160, 1)
run("C4 geometry (d12, 4K, 2 lights) on 1 GPU", 12, 3840, 2160, 1, lights=2)
run("C5 light count on the headline scene (d12, 1080p, 4 lights)", 12, 1920, 1080, 1, lights=4)
run("beyond BASELINE: d13 (8192^3), 1080p, primary+shadow", 13, 1920, 1080, 1)
