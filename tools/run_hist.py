#!/usr/bin/env python3
"""Lane steps of the exact kernel by run length (iterations between two node events) -- how much of a frame a closed-form
jump for runs of at least K iterations could cover.  Needs the profiling build:
    bash tools/build_variant.sh stats -DVRC_SCHED_STATS
    VRC_LIB_PATH=gpurun_variants/libvrc_stats.so python tools/run_hist.py [d12 d14 d16]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import voxel_raycaster_amd as vrc


def hist(c, name, sets=()):
    for k, v in sets:
        assert c.add_to_settings_buffer(k, k.upper(), v)
    assert c.compute(), c.last_error()
    buf = (C.c_ulonglong * 112)()     # 3 x 32 histogram + 12 counters
    assert vrc.lib.vrc_stats_run_hist(buf, 1) == 0
    assert c.compute(), c.last_error()
    assert vrc.lib.vrc_stats_run_hist(buf, 1) == 0
    h = np.array(buf[:96], dtype=np.uint64).reshape(3, 32)
    js = [int(v) for v in buf[96:104]]
    ctr = c.counters()
    tot_e, tot_s = int(h[0].sum()), int(h[1].sum())
    print(f"## {name}: {tot_e / 1e6:.1f} M node events, {tot_s / 1e9:.2f} G lane steps (counter: {ctr['steps'] / 1e9:.2f} G), "
          f"{tot_s / max(tot_e, 1):.0f} steps per event")
    print("| run length L | events | share of events | lane steps | share of steps | steps in runs >= 2^b | steps by ESTIMATE bucket | est >= 2^b |")
    print("|---|---|---|---|---|---|---|---|")
    for b in range(32):
        if h[0][b] == 0 and h[2][b] == 0:
            continue
        print(f"| {1 << b}..{(2 << b) - 1} | {int(h[0][b])} | {h[0][b] / tot_e:.4f} | {int(h[1][b])} | {h[1][b] / tot_s:.4f} | "
              f"{h[1][b:].sum() / tot_s:.4f} | {int(h[2][b])} | {h[2][b:].sum() / tot_s:.4f} |")
    print(json.dumps({"scene": name, "sched": c.scheduler_stats(),
                      "jumps": dict(block_passes=js[0], lane_jumps=js[1], iterations_covered=js[2], left_node=js[3], capped=js[4],
                                    pair_solves=js[5], lanes_wanting=js[6], rounds=js[7]),
                      "safe_run": dict(wave_iterations=int(buf[105]), ungated_prefix_could_cover=int(buf[104])),
                      "empty_nodes_entered": dict(above_the_table_level=int(buf[106]), below_it=int(buf[107]))}), flush=True)


def device_caster(depth, w=1920, h=1080, thickness=2):
    dim = 1 << depth
    c = vrc.CLCaster()
    assert c.init(0)
    c.build_shell_terrain(depth, 1, thickness, 2)
    _, hi = vrc.shell_column(depth, dim // 2, dim // 8, thickness=thickness)
    sc = bench.build_scene(8)
    li = sc["lights"].copy()
    li[:, 4:7] *= dim / 256.0
    ok = (c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim) and c.add_to_settings_buffer("using_octree", "OCTENABLED", 0)
          and c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", 3 * dim) and c.add_to_settings_buffer("octree_bias", "OCTREE_BIAS", 0)
          and c.assign_camera(np.array([2.0, 1.5708], dtype=np.float32),
                              np.array([dim / 2 + 0.37, dim / 8 + 0.41, hi + dim // 16 + 0.29], dtype=np.float32))
          and c.create_viewport(w, h) and c.assign_lights(li) and c.create_texture_atlas(sc["atlas"], (16, 16)) and c.validate())
    assert ok, c.last_error()
    return c


if __name__ == "__main__":
    sets = [(a[6:].split("=")[0], int(a[6:].split("=")[1])) for a in sys.argv[1:] if a.startswith("--set=")]
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["d12", "d14", "d16"]
    for w in which:
        d = int(w[1:])
        if d <= 13:
            c = bench.make_caster(bench.build_scene(d), 1920, 1080, 0)
        else:
            c = device_caster(d)
        hist(c, f"depth {d}, 1920x1080, 1 light" + "".join(f", {k}={v}" for k, v in sets), sets)
        del c
