"""Host check of voxel-raycaster_amd/csrc/exact_jump.hpp (the closed-form multi-iteration DDA jump):
the same header the gfx950 kernel compiles, driven on random ray states against the plain float
loop of kernels/ray_caster_kernel.cl:558-560.  The plain loop records the state after every iteration;
after EVERY jump the three intersection_t values (bits), the countdowns and the iteration count must sit
on a recorded state, and the face mask of the leaving iteration and the exit / cap verdict must agree.
The states cover ties (equal, rational and power-of-two direction ratios), frozen axes (negative, zero
and tiny t, slow axes below delta_t, unsettled half-way roundings), binade ends, binades up to 2^18,
countdowns up to 40000, step caps inside the stretch, tables kept up, cut short or absent (pairs solved on
the spot), jumps mixed with plain steps.  pair_solve (extended Euclid in floats) is checked against its
definition first.  A second build perturbs the reciprocal estimate by +-1 ulp like the hardware
v_rcp_f32 to exercise the integer fix-ups."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "jumptest", "jump_vs_loop.cpp")


@pytest.mark.parametrize("flags,seed", [([], 101), (["-DVRC_JUMP_FUZZ_RCP"], 202), (["-DVRC_JUMP_RING=3"], 303),
                                        (["-DVRC_EUCLID_FLOOR", "-DVRC_JUMP_ROOM", "-DVRC_JUMP_FUZZ_RCP"], 404)],
                         ids=["exact-rcp", "fuzzed-rcp", "ring-of-3-rows", "round-3-forms"])
def test_jump_equals_plain_loop(tmp_path, flags, seed):
    exe = str(tmp_path / "jump_vs_loop")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", *flags, "-o", exe, SRC])
    out = subprocess.run([exe, "150000", str(seed)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "mismatches 0" in out.stdout
    # the jumps must actually cover the bulk of the iterations, otherwise the test proves nothing
    words = out.stdout.split()
    covered, plain = int(words[words.index("covering") + 1]), int(words[words.index("iterations,") + 1])
    assert covered > 5 * plain
