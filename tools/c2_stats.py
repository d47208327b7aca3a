#!/usr/bin/env python3
"""Scheduler statistics / kernel time for BASELINE configs[1] (depth-10, 1080p, primary rays only); run with the profiling library in place."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sc = bench.build_scene(int(os.environ.get("DEPTH", "10")))
c = bench.make_caster(sc, 1920, 1080, 0, shadow_rays=0)
for kv in filter(None, os.environ.get("EXTRA", "").split(",")):
    k, v = kv.split("="); c.add_to_settings_buffer(k, k.upper(), int(v))
for _ in range(3): assert c.compute()
c.timing_reset()
for _ in range(10): assert c.compute()
n, ms = c.timing(); ctr = c.counters()
print(json.dumps({"kernel_ms": round(ms / n, 4), "steps": ctr["steps"], "descriptor_reads": ctr["descriptor_reads"], "sched": c.scheduler_stats()}))
