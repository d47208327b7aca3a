#!/usr/bin/env python3
"""Kernel time of a deep scene with and without the opt-in exact jumps: python tools/deep_jump_probe.py <depth> <jump_min_run,...>"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
D = int(sys.argv[1]) if len(sys.argv) > 1 else 15
runs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1 << 24, 256, 1024]
sc = bench.build_scene(D)
c = bench.make_caster(sc, 1920, 1080, 0)
c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 1 << 24)
for v in runs:
    c.overwrite_setting("jump_min_run", v)
    for _ in range(2): assert c.compute(), c.last_error()
    c.timing_reset()
    for _ in range(4): assert c.compute()
    n, ms = c.timing(); ctr = c.counters()
    print(json.dumps({"depth": D, "jump_min_run": v, "kernel_ms": round(ms / n, 3), "steps": ctr["steps"]}), flush=True)
