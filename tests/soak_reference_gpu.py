#!/usr/bin/env python3
"""Long-running corroboration of the oracle against the REFERENCE'S OWN raycaster kernel on the MI355X (not collected
by pytest): random camera poses / directions / light positions in the probe scenes of tests/scenes.py, the reference
kernel (oracle/_ref/ref_raycaster_gfx950_strict.co, its 20-step cap, its two image builtins redirected) against
oracle/vrc_oracle.c with tests/refcompare.compare -- hit voxel, face, material, texel, step counts of the primary ray
exactly, what follows the library normalize statistically.  python tests/soak_reference_gpu.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refcompare  # noqa: E402
import scenes  # noqa: E402
import test_reference_pin_gpu as pin  # noqa: E402
import voxel_raycaster_amd as vrc  # noqa: E402


def run(budget=300.0, seed=1, limit=None):     # limit: stop after this many frames (fixed volume; the budget is then a safety net)
    """Returns (frames with a difference in what the primary ray decides, frames, statistics).  tests/test_soak_slices_gpu.py
    runs a 10-second slice of it."""
    rng = np.random.default_rng(seed)
    probe = C.CDLL(os.path.join(pin.REF, "libref_probe.so"))
    probe.ref_probe_last_error.restype = C.c_char_p
    atlas = vrc.synthetic_atlas()
    makers = [m for m in scenes.REFERENCE_KERNEL_SCENES if m is not scenes.terrain256]
    t0, frames, pixels, shaded, failures, totals = time.time(), 0, 0, 0, 0, {}
    while time.time() - t0 < budget and (limit is None or frames < limit):
        s = dict(makers[int(rng.integers(len(makers)))]())
        dim = s["dim"]
        # a camera somewhere in or just outside the map, any direction; the reference's 20-step cap keeps what it sees local
        s["cam_pos"] = tuple(float(v) for v in (rng.random(3) * (dim + 4) - 2))
        s["cam_dir"] = (float(rng.random() * 3.0 + 0.07), float(rng.random() * 6.2 + 0.04))
        li = np.array(s["lights"], dtype=np.float32).reshape(-1, 10).copy()
        li[:, 4:7] = rng.random((li.shape[0], 3)) * dim
        s["lights"] = li
        w, h = (64, 48) if rng.random() < 0.5 else (160, 120)
        rec, trig, buf, root = pin.run_reference_raycaster(probe, "ref_raycaster_gfx950_strict.co", s, w, h, atlas)
        oimg, ohits, octr = refcompare.oracle_frame(s, w, h, atlas, buf, root, trig)
        try:
            refcompare.compare(s, w, h, rec, oimg, ohits, octr, verbose=False, totals=totals)
        except AssertionError as e:
            failures += 1
            print("MISMATCH", s["name"], s["cam_pos"], s["cam_dir"], li[0, 4:7].tolist(), (w, h), str(e)[:200], flush=True)
        frames += 1
        pixels += w * h
        shaded += int((ohits[..., 3] != 0).sum())
    n = max(totals.get("shaded", 0), 1)
    print(f"reference soak: {frames} frames, {pixels} pixels ({shaded} with a hit) compared with the reference kernel's records: "
          f"{failures} frames with a difference in what the primary ray decides (written, hit voxel, face, material, texel, "
          f"steps and colour of rays that hit nothing, bounce count); after the shadow redirect (library normalize / "
          f"fast_distance), of {totals.get('shaded', 0)} shaded pixels: in-shadow flag equal {totals.get('shadow_same', 0) / n:.6f}, "
          f"final step count equal {totals.get('same_steps', 0) / n:.6f}, alpha equal {totals.get('alpha_same', 0) / n:.6f}, "
          f"rgb within 1e-5 {totals.get('rgb_1e-5', 0) / n:.6f}, within 1e-4 {totals.get('rgb_1e-4', 0) / n:.6f} "
          f"(worst {totals.get('worst_rgb', 0.0):.2e}); {time.time() - t0:.0f} s")
    return failures, frames, totals


if __name__ == "__main__":
    sys.exit(1 if run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)[0] else 0)
