#!/usr/bin/env python3
"""Exact closed-form jumps on/off on the same frames: kernel time per jump_min_run, and the frame (image + hit records)
bit for bit against the frame rendered without jumps.  python tools/jump_ab.py [d12 d13 d14 d16] [--k 64,96,128] [--stats]
(--stats needs VRC_LIB_PATH=gpurun_variants/libvrc_stats.so)"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import voxel_raycaster_amd as vrc
from run_hist import device_caster

ap = argparse.ArgumentParser()
ap.add_argument("scenes", nargs="*", default=["d12"])
ap.add_argument("--k", default="32,64,96,128,192,256,512")
ap.add_argument("--lanes", default="1")
ap.add_argument("--lights", type=int, default=1)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--stats", action="store_true")
ap.add_argument("--time", action="store_true", help="per-phase shader clock (needs a -DVRC_TIME_STATS build)")
a = ap.parse_args()
OFF = 1 << 24


PHASES = ["jump_estimate", "euclid_fill", "jump_block", "safe_run", "single+exact", "events", "relight+shade", "unused", "kernel"]


def phase_times(c):
    buf = (C.c_ulonglong * 16)()
    assert vrc.lib.vrc_stats_time(buf, 1) == 0
    assert c.compute(), c.last_error()
    assert vrc.lib.vrc_stats_time(buf, 1) == 0
    tot = max(buf[8], 1)
    return {n: round(buf[i] / tot, 4) for i, n in enumerate(PHASES) if n != "unused"} | {"kernel_Mticks_per_wave_sum": round(buf[8] / 1e6, 1)}


def timed(c, frames):
    for _ in range(2):
        assert c.compute(), c.last_error()
    c.timing_reset()
    for _ in range(frames):
        assert c.compute(), c.last_error()
    n, ms = c.timing()
    return ms / n


for sname in a.scenes:
    d = int(sname[1:])
    if d <= 13:
        c = bench.make_caster(bench.build_scene(d), 1920, 1080, 0, light_count=a.lights)
    else:
        c = device_caster(d)
        assert c.add_to_settings_buffer("light_count", "LIGHT_COUNT", a.lights)
    assert c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", OFF)
    ms0 = timed(c, a.frames)
    img0, hits0, ctr0 = c.read_image(), c.read_hits(), c.counters()
    row0 = {"scene": sname, "jump_min_run": "off", "kernel_ms": round(ms0, 4), "steps": ctr0["steps"]}
    if a.time:
        row0["phases"] = phase_times(c)
    print(json.dumps(row0), flush=True)
    for lanes in [1]:
        for k in [int(x) for x in a.k.split(",")]:
            assert c.overwrite_setting("jump_min_run", k)
            ms = timed(c, a.frames)
            img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
            same = bool(np.array_equal(img.view(np.uint32), img0.view(np.uint32)) and np.array_equal(hits, hits0) and ctr == ctr0)
            row = {"scene": sname, "jump_min_run": k, "kernel_ms": round(ms, 4), "speedup": round(ms0 / ms, 3),
                   "bit_identical": same}
            if not same:
                row["pixels_differ"] = int((img.view(np.uint32) != img0.view(np.uint32)).any(-1).sum())
                row["hits_differ"] = int((hits != hits0).any(-1).sum())
                row["ctr"] = ctr; row["ctr0"] = ctr0
            if a.time:
                row["phases"] = phase_times(c)
            if a.stats:
                buf = (C.c_ulonglong * 112)()
                assert vrc.lib.vrc_stats_run_hist(buf, 1) == 0
                assert c.compute()
                assert vrc.lib.vrc_stats_run_hist(buf, 1) == 0
                js = list(buf[96:104])
                row["jump_stats"] = dict(block_passes=js[0], lane_jumps=js[1], iterations_covered=js[2], left_node=js[3], capped=js[4],
                                         pair_solves=js[5], lanes_wanting=js[6], rounds=js[7],
                                         lanes_per_pass=round(js[1] / max(js[0], 1), 1), iters_per_jump=round(js[2] / max(js[1], 1), 1),
                                         pairs_solved_on_the_spot=int(buf[108]), lanes_waiting_for_such_a_solve=int(buf[109]),
                                         pairs_with_a_possible_tie=int(buf[110]), lanes_in_a_warm_tie_pass=int(buf[111]))
                row["sched"] = c.scheduler_stats()
            print(json.dumps(row), flush=True)
    del c
