#!/usr/bin/env python3
"""Generates tests/golden/ref_get_oct_vox_*.npz and tests/golden/ref_view_light.npz: OUTPUTS OF THE REFERENCE'S OWN
FUNCTIONS get_oct_vox (kernels/ray_caster_kernel.cl:140-251) and view_light (:78-99), compiled unmodified for gfx950
(oracle/ref_probe.cl #includes the reference kernel source; oracle/_ref/ref_probe_gfx950*.co) and run on an MI355X.

    gpurun -- 'python tests/make_reference_pin_golden.py gpurun_out/ref_pin_golden'     (then copy the .npz into tests/golden/)

SURVEY 8c's G2 fixtures: the live comparison of tests/test_reference_pin_gpu.py as committed vectors, so that the oracle is
checked against reference-produced data on every CPU run (tests/test_oracle_cpu.py::test_oracle_matches_the_reference_*).
Inputs are stored with the outputs (positions / cases); the trees are rebuilt by the CPU test from the stored seed, density
and size with the oracle's builder.  get_oct_vox leaves its stack entries beyond the current depth uninitialised: they are
zeroed here (the depth comes from the function's own `scale` / `parent_stack_position` outputs), so the files are
deterministic.  Needs the GPU box; nothing under /root/reference is read at run time."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orc  # noqa: E402  (the tree builder only: the vectors are the reference's outputs)
import test_reference_pin_gpu as pin  # noqa: E402

TREES = [(16, 1.0, 0, None), (16, 0.3, 1, None), (64, 0.5, 2, 4000), (128, 0.02, 3, 4000), (256, 0.002, 4, 4000)]   # dim, density, seed, samples


def oct_vox_inputs(dim, density, seed, samples):
    rng = np.random.default_rng(seed)
    grid = (rng.random(dim ** 3) < density).astype(np.int8) * 5
    if samples is None:
        pos = np.stack(np.meshgrid(np.arange(dim), np.arange(dim), np.arange(dim), indexing="ij"), -1).reshape(-1, 3)
    else:
        pos = rng.integers(0, dim, size=(samples, 3))
    return grid, np.ascontiguousarray(pos, dtype=np.int32)


def view_light_inputs(n=6000, seed=5):
    rng = np.random.default_rng(seed)
    cases = np.zeros((n, 14), dtype=np.float32)
    cases[:, 0:4] = rng.random((n, 4))
    cases[:, 4:7] = (rng.random((n, 3)) - 0.5) * rng.choice([4.0, 60.0, 3000.0], size=(n, 1))
    cases[:, 7:11] = rng.random((n, 4)) * np.array([0.05, 0.05, 0.05, 0.5])
    cases[:, 11:14] = (rng.random((n, 3)) - 0.5) * rng.choice([4.0, 60.0, 3000.0], size=(n, 1))
    mask = np.zeros((n, 3), dtype=np.int32)
    axis = rng.integers(0, 3, n)
    mask[np.arange(n), axis] = rng.choice([-1, 1], n)
    two = rng.random(n) < 0.1
    mask[two, (axis[two] + 1) % 3] = rng.choice([-1, 1], int(two.sum()))
    cases[:50, 4:7] = 0.0                                 # light exactly at the hit: returns zero (:80-81)
    return cases, mask


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    lib = C.CDLL(pin.LIB)
    lib.ref_probe_last_error.restype = C.c_char_p
    for dim, density, seed, samples in TREES:
        grid, pos = oct_vox_inputs(dim, density, seed, samples)
        buf, root = orc.octree_generate(grid, dim)
        n = pos.shape[0]
        out = np.zeros((n, 40), dtype=np.int32)
        rc = lib.ref_probe_get_oct_vox(pin.CO.encode(), pin._i(pos), n, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size,
                                       C.c_uint64(root), C.c_int64(dim), pin._i(out))
        assert rc == 0, lib.ref_probe_last_error().decode()
        # words 14.. : idx_stack[8], parent_stack_index[8], parent_stack[8] (low words) -- live entries only
        for o in out:
            scale, psp = int(o[1]), int(o[3])
            o[14 + scale + 1:22] = 0
            o[22 + psp + 1:30] = 0
            o[30 + psp + 1:38] = 0
            o[38:] = 0
        path = os.path.join(out_dir, f"ref_get_oct_vox_{dim}_seed{seed}.npz")
        np.savez_compressed(path, dim=dim, density=density, seed=seed, positions=pos, out=out,
                            code_object=np.array("oracle/_ref/ref_probe_gfx950.co"))
        print(path, os.path.getsize(path), "bytes;", n, "voxels,", int((out[:, 0] != 0).sum()), "found")
    cases, mask = view_light_inputs()
    for tag, co in (("reference_flags", "ref_probe_gfx950.co"), ("no_fast_math", "ref_probe_gfx950_strict.co")):
        out = np.zeros((cases.shape[0], 4), dtype=np.float32)
        rc = lib.ref_probe_view_light(os.path.join(pin.REF, co).encode(), pin._f(cases), pin._i(mask), pin._f(out), cases.shape[0])
        assert rc == 0, lib.ref_probe_last_error().decode()
        path = os.path.join(out_dir, f"ref_view_light_{tag}.npz")
        np.savez_compressed(path, cases=cases, mask=mask, out=out, code_object=np.array("oracle/_ref/" + co))
        print(path, os.path.getsize(path), "bytes;", cases.shape[0], "cases")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ref_pin_golden"))
