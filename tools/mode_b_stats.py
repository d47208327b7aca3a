#!/usr/bin/env python3
"""Block statistics of the mode-B kernel on the headline frame: how often a wave ran the jump block / a descent block
of VRC_ROUND_PROGRAM and how many of its 64 lanes had work there.  Needs a profiling build of the library:
bash tools/build_variant.sh stats -DVRC_SCHED_STATS, copied over voxel-raycaster_amd/libvrc.so on the GPU box
(tools/gpu_variants.sh does the copying)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sc = bench.build_scene(12)
c = bench.make_caster(sc, 1920, 1080, 0)
assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 1)
assert c.compute()
st = c.scheduler_stats()
ctr = c.counters()
out = dict(wave_rounds=st["wave_step_iterations"], jump_block_runs=st["bursts"], jump_lanes=st["event_passes"],
           descent_block_runs=st["event_lanes"], descent_lanes=st["shade_lanes"], hit_block_runs=st["shade_passes"],
           jump_block_fill=round(st["event_passes"] / (64.0 * max(st["bursts"], 1)), 3),
           descent_block_fill=round(st["shade_lanes"] / (64.0 * max(st["event_lanes"], 1)), 3),
           descents_per_jump=round(st["shade_lanes"] / max(st["event_passes"], 1), 3),
           descriptor_reads=ctr["descriptor_reads"], rays=ctr["primary_rays"] + ctr["shadow_rays"])
print(json.dumps(out))
