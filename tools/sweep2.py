#!/usr/bin/env python3
"""Grid sweep of scheduling knobs on the headline workload: python tools/sweep2.py name=v1,v2 name2=v1,v2 ..."""
import sys, os, json, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
axes = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in sys.argv[1:]]
sc = bench.build_scene(int(os.environ.get("DEPTH", "12")))
c = bench.make_caster(sc, 1920, 1080, 0)
for name, vals in axes:
    c.add_to_settings_buffer(name, name.upper(), vals[0])
for combo in itertools.product(*[v for _, v in axes]):
    for (name, _), v in zip(axes, combo):
        c.overwrite_setting(name, v)
    for _ in range(2): assert c.compute(), c.last_error()
    c.timing_reset()
    for _ in range(6): assert c.compute()
    n, ms = c.timing()
    print(json.dumps({**{name: v for (name, _), v in zip(axes, combo)}, "kernel_ms": round(ms / n, 3)}), flush=True)
