"""Host check of voxel-raycaster_amd/csrc/safe_run.hpp (the SVO kernel's countdown-free "safe run" and its
arithmetic face mask): the same header the gfx950 kernel compiles, driven on random ray states inside empty
nodes against the plain float loop of kernels/ray_caster_kernel.cl:558-560.  The safe run must never take the
step that leaves the node, must leave intersection_t bit-identical to the plain loop after the same number of
iterations, and must report exactly the steps the plain loop took on every axis (ties included)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "jumptest", "safe_vs_loop.cpp")


@pytest.mark.parametrize("seed", [11, 12])
def test_safe_run_equals_plain_loop(tmp_path, seed):
    exe = str(tmp_path / "safe_vs_loop")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-o", exe, SRC])
    out = subprocess.run([exe, "200000", str(seed)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    words = out.stdout.split()
    val = lambda k: int(words[words.index(k) + 1])
    assert val("mismatches") == 0
    # the test must exercise what it claims: most gates open, long runs, early stops at the threshold, ties
    assert val("opened") > 100000 and val("iterations") > 20 * val("opened")
    assert val("stopped_before_cap") > 10000 and val("tie_iterations") > 100000
