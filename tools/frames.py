#!/usr/bin/env python3
"""Renders N frames of the headline workload in a given stepping mode (for rocprofv3: no compiler, no oracle, no child
process is ever started from here).  python3 tools/frames.py --mode 1 --frames 10 [--hit-records 0]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--mode", type=int, default=0)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--depth", type=int, default=12)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--lights", type=int, default=1)
ap.add_argument("--hit-records", type=int, default=1)
ap.add_argument("--set", action="append", default=[], help="name=value setting overrides")
ap.add_argument("--streams", type=int, default=1, help="casters rendering concurrently (each its own HIP stream and buffers)")
a = ap.parse_args()
if a.depth <= 13:
    sc = bench.build_scene(a.depth)
    c = bench.make_caster(sc, a.width, a.height, 0, light_count=a.lights, hit_records=a.hit_records)
else:                                                   # deeper terrains only exist on the device (the host emitter would take hours)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from run_hist import device_caster
    c = device_caster(a.depth, a.width, a.height)
    assert c.add_to_settings_buffer("light_count", "LIGHT_COUNT", a.lights) and c.add_to_settings_buffer("hit_records", "HIT_RECORDS", a.hit_records)
assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", a.mode)
for kv in a.set:
    k, v = kv.split("=")
    assert c.overwrite_setting(k, int(v)) or c.add_to_settings_buffer(k, k.upper(), int(v))
for _ in range(2):
    assert c.compute(), c.last_error()
c.timing_reset()
for _ in range(a.frames):
    assert c.compute(), c.last_error()
n, ms = c.timing()
ctr = c.counters()
if a.streams > 1:
    import time
    cs = [c] + [bench.make_caster(sc, a.width, a.height, 0, light_count=a.lights, hit_records=a.hit_records) for _ in range(a.streams - 1)]
    for q in cs[1:]:
        assert q.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", a.mode)
        for kv in a.set:
            k, v = kv.split("=")
            assert q.overwrite_setting(k, int(v)) or q.add_to_settings_buffer(k, k.upper(), int(v))
        assert q.compute(), q.last_error()
    t0 = time.perf_counter()
    for _ in range(a.frames):
        for q in cs:
            assert q.compute_async()
        for q in cs:
            assert q.sync()
    ms, n = (time.perf_counter() - t0) * 1e3, a.frames * a.streams
b = bench.algorithmic_bytes(ctr, a.width * a.height, a.width * a.height - ctr["unwritten_pixels"])
print(json.dumps({"mode": a.mode, "kernel_ms_avg": round(ms / n, 4), "rays": ctr["primary_rays"] + ctr["shadow_rays"],
                  "Mrays_s": round((ctr["primary_rays"] + ctr["shadow_rays"]) / (ms / n) / 1e3, 1), "algorithmic_bytes": b,
                  "GB_s": round(b / (ms / n) / 1e6, 1), "counters": ctr}))
