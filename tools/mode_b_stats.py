#!/usr/bin/env python3
"""Lane-round statistics of the mode-B kernel on the headline frame: what the lanes of a wave are doing when a round
starts.  Needs a profiling build of the library: bash tools/build_variant.sh stats -DVRC_SCHED_STATS, copied over
voxel-raycaster_amd/libvrc.so on the GPU box (tools/gpu_variants.sh does the copying)."""
import sys, os, json
sys.path.insert(0, '/root/repo')
import bench
sc = bench.build_scene(12)
c = bench.make_caster(sc, 1920, 1080, 0)
assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 1)
assert c.compute()
st = c.scheduler_stats()
names = dict(wave_step_iterations="wave rounds", bursts="lane-rounds jumping", event_passes="lane-rounds descending", event_lanes="lane-rounds parked for shading", shade_passes="hit-block passes (wave)", shade_lanes="lane-rounds finished/idle")
print({names.get(k, k): v for k, v in st.items()})
tot = st["bursts"] + st["event_passes"] + st["event_lanes"] + st["shade_lanes"]
print("lane-rounds total", tot, "= 64 x wave rounds", 64 * st["wave_step_iterations"])
for k in ("bursts", "event_passes", "event_lanes", "shade_lanes"):
    print(names[k], round(st[k] / tot, 3))
