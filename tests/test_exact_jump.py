"""Host check of voxel-raycaster_amd/csrc/exact_jump.hpp (the closed-form multi-iteration DDA jump):
the same header the gfx950 kernel compiles, driven on random ray states against the plain float
loop of kernels/ray_caster_kernel.cl:558-560.  Every field must match bit for bit: the three
intersection_t values, the countdowns, the number of loop iterations, the face mask of the last
iteration and the exit/cap verdict.  A second build perturbs the reciprocal estimate by +-1 ulp
like the hardware v_rcp_f32 to exercise the integer fix-ups."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "jumptest", "jump_vs_loop.cpp")


@pytest.mark.parametrize("flags,seed", [([], 101), (["-DVRC_JUMP_FUZZ_RCP"], 202)], ids=["exact-rcp", "fuzzed-rcp"])
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
