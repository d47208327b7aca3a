#!/usr/bin/env python3
"""Empty boxes on / off on the same scenes: frames and hit records must be bit-identical (the descriptor-read field aside),
kernel time side by side, the boxes' build time and their self-check.
python3 tools/box_ab.py [--depths 12 10] [--frames 20] [--lights 1 4] [--check 4194304]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (before libvrc.so: one HIP runtime per process, see bench.py)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--depths", type=int, nargs="*", default=[12])
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--lights", type=int, nargs="*", default=[1])
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--check", type=int, default=1 << 22)
ap.add_argument("--shadow", type=int, default=1)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--modes", type=int, nargs="*", default=[0, 1], help="empty_boxes values to compare; the first is the reference frame (0: none, 1: full, 2: table cells only, -1: the default rule)")
a = ap.parse_args()


def timed(c, frames):
    for _ in range(3):
        assert c.compute(), c.last_error()
    c.timing_reset()
    for _ in range(frames):
        assert c.compute(), c.last_error()
    n, ms = c.timing()
    return ms / n


for depth in a.depths:
    sc = bench.build_scene(depth) if depth <= 13 else bench.device_scene_header(depth)
    for lights in a.lights:
        out = {"depth": depth, "lights": lights, "frame": f"{a.width}x{a.height}"}
        frames = {}
        for boxes in a.modes:
            c = bench.make_caster(sc, a.width, a.height, 0, light_count=lights, hit_records=1, shadow_rays=a.shadow)
            assert c.add_to_settings_buffer("empty_boxes", "EMPTY_BOXES", boxes)
            for kv in a.set:
                k, v = kv.split("=")
                assert c.overwrite_setting(k, int(v)) or c.add_to_settings_buffer(k, k.upper(), int(v))
            assert c.compute(), c.last_error()
            img, hits = c.read_image().copy(), c.read_hits().copy()
            ctr = c.counters()
            frames[boxes] = (img, hits, ctr)
            assert c.overwrite_setting("hit_records", 0)
            out[f"ms_boxes{boxes}"] = round(timed(c, a.frames), 4)
            out[f"desc_reads_boxes{boxes}"] = ctr["descriptor_reads"]
            if c.used_empty_boxes():
                out[f"check_boxes{boxes}"] = c.empty_boxes_check(a.check, 7)
                m = c.memory_usage2()
                out[f"box_MB_boxes{boxes}"] = round(m["box_bytes"] / 1e6)
            del c
        i0, h0, c0 = frames[a.modes[0]]
        h0 = h0.reshape(-1, 8)
        for boxes in a.modes[1:]:
            i1, h1, c1 = frames[boxes]
            h1 = h1.reshape(-1, 8)
            out[f"same_as_first_boxes{boxes}"] = dict(
                image_bits=bool(np.array_equal(i0.view(np.uint32), i1.view(np.uint32))), hits_but_reads=bool(np.array_equal(h0[:, :7], h1[:, :7])),
                differing_pixels=int(np.count_nonzero(np.any(i0.reshape(-1, 4).view(np.uint32) != i1.reshape(-1, 4).view(np.uint32), axis=1))),
                differing_hit_rows=int(np.count_nonzero(np.any(h0[:, :7] != h1[:, :7], axis=1))),
                counters_but_reads=all(c0[k] == c1[k] for k in c0 if k not in ("descriptor_reads", "canonical_reads")))
        print(json.dumps(out), flush=True)
