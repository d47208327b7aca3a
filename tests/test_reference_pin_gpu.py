"""Pins the oracle against the REFERENCE'S OWN CODE running on the MI355X (-m gpu).

oracle/_ref/ref_probe_gfx950.co is kernels/ray_caster_kernel.cl of the reference compiled unmodified for gfx950 with
the reference's own build flags (oracle/Makefile target _ref, built in the development container where
/root/reference exists) plus the probe kernels of oracle/ref_probe.cl, which call the two pure functions the
raycaster is made of and store what they return:

  get_oct_vox (:140-251)  the octree point query                 -> integer fields, compared exactly
  view_light  (:78-99)    the shading arithmetic of the hit block -> floats.  The reference's normalize /
                          fast_length / divide are the OpenCL device library's (approximate) versions, so bit
                          equality with an IEEE restatement is not defined.  Measured on 20 000 random cases: 85 %
                          bit-identical, 99.98 % within BASELINE's 1e-5 relative, worst 3.1e-5 (the half-way vector
                          of nearly opposite light / view directions amplifies the library's 1-2 ulp) -- the same
                          with the reference's fast-math flags and without them (_strict code object).

The raycaster kernel itself cannot be observed on this GPU (image2d_t I/O, no image hardware on CDNA4:
profiles/r01_reference_kernel_on_gfx950.txt), so the step loop and the UV/texel code stay a restatement.
Nothing here reads /root/reference at run time: only the prebuilt code object travels to the GPU box."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
CO = os.path.join(REF, "ref_probe_gfx950.co")
LIB = os.path.join(REF, "libref_probe.so")


@pytest.fixture(scope="module")
def probe():
    if not (os.path.exists(CO) and os.path.exists(LIB)):
        pytest.skip("oracle/_ref is built only where /root/reference exists (make -C oracle _ref)")
    lib = C.CDLL(LIB)
    lib.ref_probe_last_error.restype = C.c_char_p
    return lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


@pytest.mark.parametrize("code_object", ["ref_probe_gfx950.co", "ref_probe_gfx950_strict.co"], ids=["reference-flags", "no-fast-math"])
def test_view_light_of_the_oracle_matches_the_reference_function(probe, code_object):
    rng = np.random.default_rng(5)
    n = 20000
    cases = np.zeros((n, 14), dtype=np.float32)
    cases[:, 0:4] = rng.random((n, 4))                                   # in_color
    cases[:, 4:7] = (rng.random((n, 3)) - 0.5) * rng.choice([4.0, 60.0, 3000.0], size=(n, 1))   # hit - light
    cases[:, 7:11] = rng.random((n, 4)) * np.array([0.05, 0.05, 0.05, 0.5])                     # light rgbi
    cases[:, 11:14] = (rng.random((n, 3)) - 0.5) * rng.choice([4.0, 60.0, 3000.0], size=(n, 1))  # hit - camera
    mask = np.zeros((n, 3), dtype=np.int32)
    axis = rng.integers(0, 3, n)
    mask[np.arange(n), axis] = rng.choice([-1, 1], n)                    # face_mask * voxel_step: one axis
    two = rng.random(n) < 0.1
    mask[two, (axis[two] + 1) % 3] = rng.choice([-1, 1], int(two.sum()))  # a tie steps two axes
    cases[:50, 4:7] = 0.0                                                # light exactly at the hit: returns zero (:80-81)
    out = np.zeros((n, 4), dtype=np.float32)
    rc = probe.ref_probe_view_light(os.path.join(REF, code_object).encode(), _f(cases), _i(mask), _f(out), n)
    assert rc == 0, probe.ref_probe_last_error().decode()
    mine = np.stack([orc.view_light(c[0:4], c[4:7], c[7:11], c[11:14], m) for c, m in zip(cases, mask)])
    assert np.isfinite(out).all()
    assert (out[:50] == 0).all() and (mine[:50] == 0).all()
    rel = np.abs(mine - out) / np.maximum(np.abs(out), 1e-6)
    ulp = np.abs(mine.view(np.int32).astype(np.int64) - out.view(np.int32).astype(np.int64))
    print(f"view_light vs {code_object}: max rel {rel.max():.3g}, within 1e-5 {float((rel <= 1e-5).mean()):.5f}, "
          f"max ulp {int(ulp.max())}, bit-identical {float((ulp == 0).mean()):.3f}")
    assert rel.max() <= 1e-4, f"max relative difference {rel.max():.3g}"
    assert (rel <= 1e-5).mean() >= 0.999                                  # BASELINE north_star tolerance for RGB
    assert np.median(rel) == 0.0 and (ulp == 0).mean() > 0.5


@pytest.mark.parametrize("dim,density,seed", [(16, 1.0, 0), (16, 0.3, 1), (64, 0.5, 2), (128, 0.02, 3), (256, 0.002, 4)])
def test_get_oct_vox_of_the_oracle_matches_the_reference_function(probe, dim, density, seed):
    rng = np.random.default_rng(seed)
    grid = (rng.random(dim ** 3) < density).astype(np.int8) * 5
    buf, root = orc.octree_generate(grid, dim)
    if dim <= 16:
        pos = np.stack(np.meshgrid(np.arange(dim), np.arange(dim), np.arange(dim), indexing="ij"), -1).reshape(-1, 3)
    else:
        pos = rng.integers(0, dim, size=(20000, 3))
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    n = pos.shape[0]
    out = np.zeros((n, 40), dtype=np.int32)
    rc = probe.ref_probe_get_oct_vox(CO.encode(), _i(pos), n, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size,
                                     C.c_uint64(root), C.c_int64(dim), _i(out))
    assert rc == 0, probe.ref_probe_last_error().decode()
    found_any = False
    for p, o in zip(pos, out):
        ts = orc.get_oct_vox(p, buf, root, dim)
        mine = [ts.found, ts.scale, ts.resolution, ts.parent_stack_position, *ts.sub_oct_pos, *ts.oct_pos,
                ts.current_descriptor_index & 0xffffffff, ts.current_descriptor_index >> 32,
                ts.current_descriptor & 0xffffffff, ts.current_descriptor >> 32]
        assert [int(v) & 0xffffffff for v in o[:14]] == [int(v) & 0xffffffff for v in mine], (p, o[:14], mine)
        # the reference leaves stack entries beyond the current depth uninitialised: compare the live ones
        k = ts.scale + 1
        assert list(o[14:14 + k]) == [ts.idx_stack[j] for j in range(k)]
        k = ts.parent_stack_position + 1
        assert [int(v) & 0xffffffff for v in o[22:22 + k]] == [ts.parent_stack_index[j] & 0xffffffff for j in range(k)]
        assert [int(v) & 0xffffffff for v in o[30:30 + k]] == [ts.parent_stack[j] & 0xffffffff for j in range(k)]
        found_any = found_any or bool(ts.found)
        # and the grid agrees (Octree::Validate, src/map/Octree.cpp:329-352)
        assert bool(ts.found) == bool(grid[p[0] + dim * (p[1] + dim * p[2])])
    assert found_any
