#!/usr/bin/env python3
"""Tuning sweep on the headline workload (GPU box): kernel ms vs a setting."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "burst_steps"
    values = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8, 16, 24, 32, 48, 64]
    depth = int(os.environ.get("DEPTH", "12"))
    sc = bench.build_scene(depth)
    c = bench.make_caster(sc, 1920, 1080, 0, light_count=int(os.environ.get("LIGHTS", "1")))
    for kv in filter(None, os.environ.get("EXTRA", "").split(",")):      # fixed settings: EXTRA="a=1,b=2"
        k, v = kv.split("=")
        c.add_to_settings_buffer(k, k.upper(), int(v))
    c.add_to_settings_buffer(name, name.upper(), values[0])
    for v in values:
        c.overwrite_setting(name, v)
        for _ in range(3):
            assert c.compute(), c.last_error()
        c.timing_reset()
        for _ in range(10):
            assert c.compute()
        n, ms = c.timing()
        ctr = c.counters()
        rays = ctr["primary_rays"] + ctr["shadow_rays"]
        print(json.dumps({name: v, "kernel_ms": round(ms / n, 3), "Mrays/s": round(rays / (ms / n) / 1e3, 1),
                          "steps": ctr["steps"], "jump_covered": ctr["map_reads"], "sched": c.scheduler_stats()}), flush=True)

if __name__ == "__main__":
    main()
