#!/usr/bin/env python3
"""Long-running version of tests/test_parity_gpu.py::test_fuzz_random_scenes (not collected by pytest): seeded random
scenes of 4^3..128^3 voxels (materials, mirrors, attachments), random cameras inside / outside / on the grid, 1-8 lights,
ragged frame sizes, step caps -- the array kernel, the SVO kernel, the exact jumps and mode B, each against the oracle bit
for bit on the whole frame.  python tests/soak_fuzz_gpu.py [seconds] [first seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_scenes  # noqa: E402
import voxel_raycaster_amd as vrc  # noqa: E402


def run(budget=300.0, seed=100000):
    atlas = vrc.synthetic_atlas()
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        try:
            fuzz_scenes.run_case(seed + n, atlas, dims=(4, 8, 16, 32, 64, 128))
        except AssertionError as e:
            bad += 1
            print("MISMATCH", str(e)[:300], flush=True)
        n += 1
    print(f"fuzz soak: {n} random scenes (seeds {seed}..{seed + n - 1}; 4^3..128^3, array / SVO / exact jumps / mode B, each whole frame "
          f"against the oracle): {bad} with a difference; {time.time() - t0:.0f} s")
    return bad, n


if __name__ == "__main__":
    sys.exit(1 if run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 100000)[0] else 0)
