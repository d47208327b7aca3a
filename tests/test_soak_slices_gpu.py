"""Fixed-seed, FIXED-VOLUME slices of the seven soaks (-m gpu; tests/soak_*.py are the long runs): product vs oracle, the
reference's own kernel vs oracle, closed-form jumps vs the same kernel stepping, group handle vs single handle, device builders vs
host emitter, array kernel vs SVO kernel on device-built trees.  Every slice renders a fixed number of poses / frames / maps
(~10 s each on an MI355X); the time budget is only a safety net."""
import functools
import os
import resource
import time

import numpy as np
import pytest

import scenes
import voxel_raycaster_amd as vrc
from oracle import orc
from gpu_helpers import _but_reads, _frame, _peak_rss_kb, _reset_peak_rss, _rss_now_kb, bench_scene, configure, lights4, survey_camera
from test_parity_gpu import assert_same, hits_match, make_caster

pytestmark = pytest.mark.gpu

# (until round 5 these were time-budgeted -- what they covered depended on the box's speed; now every slice
# renders a fixed number of poses / frames / maps, ~10 s each on an MI355X, and the time budget is only a safety net)
def test_soak_slice_product_against_the_oracle():
    """400 poses of tests/soak_gpu.py with a fixed seed: random camera poses / light sets / step caps in the depth-12
    bench scene, exact mode (closed-form jumps on, the default at this depth) and mode B, single handles and 3-rank
    groups, device RGBA8 pack -- whole 8-row tile bands against the oracle, bit for bit."""
    import soak_gpu
    bad, poses, rows = soak_gpu.run(budget=120.0, seed=20261002, depth=12, limit=400)
    assert poses == 400 and rows == 400 * 24
    assert bad == 0


def test_soak_slice_reference_kernel_against_the_oracle():
    """3000 frames of tests/soak_reference_gpu.py with a fixed seed: the reference's own raycaster kernel on the MI355X
    (two image builtins redirected) against the oracle on random poses in the probe scenes."""
    import os
    import test_reference_pin_gpu as pin
    if not os.path.exists(os.path.join(pin.REF, "ref_raycaster_gfx950_strict.co")):
        pytest.fail("oracle/_ref/ is missing: build it with `make -C oracle _ref` where /root/reference exists")
    import soak_reference_gpu
    failures, frames, totals = soak_reference_gpu.run(budget=120.0, seed=20261002, limit=3000)
    n = max(totals.get("shaded", 0), 1)
    print(f"\nreference soak slice: {frames} frames, shaded pixels {totals.get('shaded', 0)}, rgb within 1e-5 "
          f"{totals.get('rgb_1e-5', 0) / n:.6f}, within 1e-4 {totals.get('rgb_1e-4', 0) / n:.6f}")
    assert frames == 3000 and failures == 0
    assert totals.get("rgb_1e-5", 0) / n >= 0.999


def test_soak_slice_exact_jumps_on_small_frames_of_deep_scenes():
    """1500 frames of tests/soak_jumps_gpu.py with a fixed seed: 640x360 frames (900 blocks: fewer than the chip holds,
    the case in which blocks once handed their jump-table slots across XCDs and 1 frame in 4000 came back with a few
    iteration counts off by one) of device-built depth-10 / 12 / 14 / 16 terrains, random poses, 1-4 lights, step caps and
    jump thresholds, through the empty boxes where the tree has them (depths 10, 12) -- image, hit records and every counter equal
    to the same frame stepped voxel by voxel from octree node to octree node."""
    import soak_jumps_gpu
    bad, frames, steps = soak_jumps_gpu.run(budget=150.0, seed=20261002, depths=(10, 12, 14, 16), limit=1500)
    assert frames == 1500 and steps > 1e10
    assert bad == 0


def test_soak_slice_group_handle_against_the_single_handle():
    """150 frames of tests/soak_groups_gpu.py with a fixed seed: 1-8 ranks on this GPU behind one handle (with and without
    own copies of the tree), bands of 8..128 rows, frame sizes from 1x1 to 1920x1080 that are multiples of nothing, both
    stepping modes, 1-4 lights, attachments, pinned and pageable read-back: frame, hit records, RGBA8 and counters equal
    the single handle's."""
    import soak_groups_gpu
    bad, frames = soak_groups_gpu.run(budget=120.0, seed=20261002, depths=(8, 10), limit=150)
    assert frames == 150 and bad == 0


def test_soak_slice_device_builder_against_the_host_emitter():
    """100 fields and grids of tests/soak_builder_gpu.py with a fixed seed: column fields the fixtures do not hold (white noise,
    slabs, cliffs, floating pillars, single layers, solid maps, ceilings) at depths 6-9: the device-built array equals the
    host emitter's bit for bit, the device validate passes, point queries of the tree agree with the field."""
    import soak_builder_gpu
    bad, fields, descriptors = soak_builder_gpu.run(budget=120.0, seed=20261002, depths=(6, 7, 8, 9), limit=100)
    assert fields == 100 and descriptors > 0 and bad == 0


def test_soak_slice_array_kernel_against_svo_kernel_on_device_built_trees():
    """30 maps (1200 frames) of tests/soak_array_vs_svo_gpu.py with a fixed seed: random dense maps of 128^3 / 256^3 voxels with
    materials and mirrors, the tree built on the device from the same grid (vrc_build_dense_grid + attachments), random
    cameras inside the map: the array kernel's frame and the SVO kernel's frame (closed-form jumps forced on) are the
    same image and the same hit records -- "SVO path == array path" beyond the sizes the oracle follows."""
    import soak_array_vs_svo_gpu
    bad, frames, maps = soak_array_vs_svo_gpu.run(budget=120.0, seed=20261002, depths=(7, 8), limit=30)
    assert maps == 30 and frames >= 30 * 40 and bad == 0
