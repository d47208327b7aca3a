#!/usr/bin/env python3
"""Generates tests/golden/ref_*.npz: OUTPUTS OF THE REFERENCE'S OWN `raycaster` KERNEL, run on an MI355X.

    gpurun -- 'python tests/make_reference_golden.py gpurun_out/ref_golden'     (then copy the .npz into tests/golden/)

Each file holds what the reference kernel (kernels/ray_caster_kernel.cl, #included unmodified by
oracle/ref_raycaster_probe.cl, IEEE build oracle/_ref/ref_raycaster_gfx950_strict.co; its two image builtins are
redirected to buffers -- see that file and DESIGN.md section 2) produced for one seeded scene of tests/scenes.py:
per pixel the 32-int record of its locals at its own read_imagef / write_imagef call sites (first-hit voxel, face,
material, step count, texel coordinate; final voxel, face mask, step count, shadow flag, bounce count, colour), plus
the sin/cos of the camera angles as the code object evaluated them.  Inputs are not stored: they are the seeded scene
(tests/scenes.py), the oracle-built tree and ray table, the hash atlas -- the CPU test rebuilds them.
tests/test_oracle_cpu.py::test_oracle_matches_the_reference_kernel_vectors replays the oracle against these files, so
the oracle's ray set-up, step loop, hit block and epilogue are checked against reference-produced vectors on every CPU
run.  Needs the GPU box (the code object only runs there) -- nothing under /root/reference is read at run time."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
import test_reference_pin_gpu as pin  # noqa: E402

RESOLUTIONS = {"64x48": (64, 48), "256x192": (256, 192)}
EXTRA = {"terrain256": {"640x480": (640, 480)}}      # one wide frame of the natural scene


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    lib = C.CDLL(pin.LIB)
    lib.ref_probe_last_error.restype = C.c_char_p
    atlas = scenes.hash_atlas()
    for make in scenes.REFERENCE_KERNEL_SCENES:
        for tag, (w, h) in {**RESOLUTIONS, **EXTRA.get(make.__name__, {})}.items():
            s = make()
            rec, trig, _, _ = pin.run_reference_raycaster(lib, "ref_raycaster_gfx950_strict.co", s, w, h, atlas)
            path = os.path.join(out_dir, f"ref_{make.__name__}_{tag}.npz")
            np.savez_compressed(path, records=rec, trig=trig, scene=np.array(make.__name__), width=w, height=h,
                                code_object=np.array("oracle/_ref/ref_raycaster_gfx950_strict.co"))
            print(path, os.path.getsize(path), "bytes;", int((rec[..., 16] > 0).sum()), "pixels hit,", int((rec[..., 15] == 0).sum()), "unwritten")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ref_golden"))
