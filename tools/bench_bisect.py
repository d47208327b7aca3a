#!/usr/bin/env python3
"""Round 4 finding, kept as a tool: a process that loads libvrc.so BEFORE torch ends up with two HIP runtimes (the system's
ROCm 7.2 libamdhip64 that libvrc.so links, and the 7.0 one bundled with torch) and every raycast kernel in it runs 6-7 % slower
(headline frame 2.35 vs 2.19 ms).  bench.py therefore imports torch first.   python tools/bench_bisect.py [build-first|torch-first]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import __graft_entry__ as graft
order = sys.argv[1] if len(sys.argv) > 1 else "build-first"
if order == "torch-first":
    import torch
graft.build()
import torch
torch.cuda.set_device(0)
import voxel_raycaster_amd
import ctypes
print(order, [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l and "r-xp" in l])
sc, _ = bench.shared_scene(12, 0, 1, "0", False, 2)
c = bench.make_caster(sc, 1920, 1080, 0, table=None, row_slice=None, octree_file=None, hit_records=0, shadow_rays=1, light_count=1)
def timed(tag, n=20):
    c.timing_reset()
    for _ in range(n):
        assert c.compute()
    k, ms = c.timing()
    print(tag, round(ms / k, 4), flush=True)
assert c.compute()
timed("after first compute")
c.counters()
timed("after counters()")
torch.cuda.synchronize()
timed("after torch sync")
for _ in range(40):
    c.compute()
timed("after 40 prewarm")
c2 = bench.make_caster(sc, 1920, 1080, 0, hit_records=0)
timed("with a second caster alive")
