#!/usr/bin/env python3
"""Wave-scheduler statistics of the exact kernel on the headline frame (needs the -DVRC_SCHED_STATS build, see
tools/mode_b_stats.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sc = bench.build_scene(12)
c = bench.make_caster(sc, 1920, 1080, 0, light_count=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
assert c.compute()
print(c.scheduler_stats(), c.counters())
