#!/usr/bin/env python3
"""Long-running check of the exact closed-form jumps (csrc/exact_jump.hpp) and of the empty boxes (csrc/empty_boxes.hip)
where the oracle is too slow to follow: shell terrains of depth 10..16 built on the device, random camera poses / light
positions / light counts / step caps / jump thresholds, and the frame with jumps and -- where the tree has them: depth <= 13
-- with the empty boxes (image, hit records, every counter; the descriptor-read count only where no boxes are in play)
against the same frame stepped voxel by voxel from octree node to octree node by the same kernel.  (The stepping kernel itself is what tests/soak_gpu.py and the parity suite hold against the
oracle.)  Not collected by pytest.  python tests/soak_jumps_gpu.py [seconds] [seed] [depths, e.g. 12,14,16] [frames to replay, e.g. 17,4033]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import voxel_raycaster_amd as vrc  # noqa: E402
from run_hist import device_caster  # noqa: E402

OFF = 1 << 24
SIZES = [(64, 48), (200, 120), (320, 200), (640, 360), (640, 360), (1000, 600), (1920, 1080)]


def run(budget=300.0, seed=1, depths=(10, 12, 13, 14, 16), w=640, h=360, only=None, dump=False, vary_size=False, limit=None):
    """Returns (frames that differ, frames, lane steps covered).  only: render just these frame numbers of the seeded
    sequence (replaying a run); dump: print the differing pixels with their hit records; vary_size: every caster gets a
    new viewport now and then (64x48 .. 1920x1080: from a dozen blocks to several times what the chip holds)."""
    rng = np.random.default_rng(seed)
    t0, frames, bad, steps, boxed = time.time(), 0, 0, 0, 0
    casters = {}
    size_rng, sizes_used = np.random.default_rng(seed + 1), {(w, h)}
    index = -1
    while time.time() - t0 < budget and (only is None or index < max(only)) and (limit is None or frames < limit):   # limit: a fixed number of frames
        index += 1
        depth = int(rng.choice(depths))
        dim = 1 << depth
        if depth not in casters and (only is None or index in only):
            c = device_caster(depth, w, h)
            assert c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", OFF) and c.add_to_settings_buffer("empty_boxes", "EMPTY_BOXES", 0)
            assert c.add_to_settings_buffer("light_count", "LIGHT_COUNT", 1)
            # the camera and light arrays are re-read at every compute() (the reference's USE_HOST_PTR buffers): written in place below
            cam = (np.zeros(2, dtype=np.float32), np.zeros(3, dtype=np.float32))
            li = np.zeros((8, 10), dtype=np.float32)
            assert c.assign_camera(*cam) and c.assign_lights(li) and c.validate(), c.last_error()
            casters[depth] = (c, cam, li)
        pos = rng.random(3) * (dim * 1.2) - 0.1 * dim
        if rng.random() < 0.6:                       # mostly a little above the terrain, where the long runs are
            cx, cy = int(min(max(pos[0], 0), dim - 1)), int(min(max(pos[1], 0), dim - 1))
            _, top = vrc.shell_column(depth, cx, cy, thickness=2)
            pos[2] = top + rng.random() * dim * (0.3 if rng.random() < 0.5 else 0.02) + 1.5
        draws = (rng.random() * 3.1 + 0.02, rng.random() * 6.28), rng.random((8, 4)), rng.random((8, 3)), rng.random((8, 3))
        nl = int(rng.choice([1, 1, 2, 4]))
        md = int(rng.choice([3 * dim, 3 * dim, dim // 3, 5000]))
        k = int(rng.choice([16, 32, 64, 96, 96, 128, 256, 1024]))
        if only is not None and index not in only:
            continue
        c, cam, li = casters[depth]
        if vary_size and size_rng.random() < 0.01:
            nw, nh = SIZES[int(size_rng.integers(len(SIZES)))]
            assert c.create_viewport(nw, nh) and c.validate(), c.last_error()
            sizes_used.add((nw, nh))
        cam[0][:] = draws[0]
        cam[1][:] = pos
        li[:, 0:4] = draws[1] * 0.8 + 0.2
        li[:, 4:7] = draws[2] * dim * 1.1
        li[:, 7:10] = draws[3]
        assert c.overwrite_setting("light_count", nl) and c.overwrite_setting("max_distance", md), c.last_error()
        out = []
        for setting, boxes in ((OFF, 0), (k, -1)):          # node by node, voxel by voxel  |  closed-form jumps through empty boxes
            assert c.overwrite_setting("jump_min_run", setting) and c.overwrite_setting("empty_boxes", boxes)
            assert c.compute(), c.last_error()
            out.append((c.read_image(), c.read_hits(), c.counters()))
        a, b = out
        nf = 8 if b[2]["canonical_reads"] else 7           # with the boxes the read count is the box traversal's own
        boxed += 0 if b[2]["canonical_reads"] else 1
        drop = (lambda d: d) if nf == 8 else (lambda d: {k_: v for k_, v in d.items() if k_ not in ("descriptor_reads", "canonical_reads")})
        same = drop(a[2]) == drop(b[2]) and np.array_equal(a[1][..., :nf], b[1][..., :nf]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        frames += 1
        steps += a[2]["steps"]
        if not same:
            bad += 1
            print("MISMATCH frame", index, "depth", depth, "k", k, "lights", nl, "max_distance", md, cam[1].tolist(), cam[0].tolist(),
                  int((a[1][..., :nf] != b[1][..., :nf]).any(-1).sum()), "hit records differ", flush=True)
            if dump:
                diff = (a[1][..., :nf] != b[1][..., :nf]).any(-1) | (a[0].view(np.uint32) != b[0].view(np.uint32)).any(-1)
                for y, x in zip(*np.nonzero(diff)):
                    print("  pixel", int(x), int(y), "stepping", a[1][y, x].tolist(), a[0][y, x].tolist(), "| jumps", b[1][y, x].tolist(),
                          b[0][y, x].tolist(), flush=True)
                print("  counters stepping", a[2], "| jumps", b[2], flush=True)
    shapes = "/".join(f"{a}x{b}" for a, b in sorted(sizes_used))
    print(f"jump soak: {frames} frames of {shapes} at depths {list(depths)} (random pose, 1-4 lights, step cap, jump_min_run 16..1024; {boxed} of them "
          f"through empty boxes): {bad} differ from the same frame stepped node by node without jumps; {steps / 1e12:.2f} T lane steps; {time.time() - t0:.0f} s")
    return bad, frames, steps


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    depths = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (10, 12, 13, 14, 16)
    only = set(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else None      # replay: these frames only, with details
    sys.exit(1 if run(budget, seed, depths, only=only, dump=only is not None, vary_size=only is None)[0] else 0)
