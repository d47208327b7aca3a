#!/usr/bin/env python3
"""Sizes the BASELINE configs[4] scene: descriptor count of shell-terrain(depth 16) for a few thickness / octave_floor
values (count pass of the device builder only, nothing is allocated).  python tools/c5_size.py [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voxel_raycaster_amd as vrc

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 16
c = vrc.CLCaster()
assert c.init(0)
for thickness, floor in [(2, 2), (16, 2), (24, 2), (30, 2), (30, 0), (36, 2)]:
    t0 = time.perf_counter()
    info, _ = c.build_shell_terrain(depth, 1, thickness, floor, count_only=True)
    print(f"depth {depth} thickness {thickness} floor {floor}: {info['n_descriptors'] / 1e9:.3f} G descriptors = "
          f"{info['n_descriptors'] * 8 / 1e9:.1f} GB, {info['n_bricks']} bricks, {time.perf_counter() - t0:.2f} s "
          f"(height {info['seconds_height']:.2f}, count {info['seconds_count']:.2f})", flush=True)
