#!/usr/bin/env python3
"""Long-running check of the multi-GPU group handle (include/vrc.h vrc_create_group, SURVEY 8e: row bands of one frame
dealt to the ranks, one gathered read-back) against the single handle, on ONE GPU (all this box has; every rank sits on
device 0, with and without VRC_GROUP_OWN_COPIES): random rank counts 1..8, band heights 8..128 (multiples of the 8-row tile), frame sizes that are not
multiples of anything, both stepping modes, 1-4 lights, attachments, step caps, jump thresholds, pinned and pageable
destinations -- image, hit records, RGBA8 frame and counters of the group equal to the single handle's.
Not collected by pytest.  python tests/soak_groups_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import voxel_raycaster_amd as vrc  # noqa: E402


def setup(c, sc, tree, w, h, cam, li, settings, holder=None):
    """holder: a caster that keeps this tree (with its coarse table and boxes) on the GPU; adopted instead of uploaded again"""
    dim = sc["dim"]
    ok = (c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim) and c.add_to_settings_buffer("using_octree", "OCTENABLED", 0))
    for k, v in settings.items():
        ok = ok and c.add_to_settings_buffer(k, k.upper(), v)
    ok = (ok and (c.assign_octree_from(holder) if holder is not None else c.assign_octree(tree)) and c.assign_camera(*cam) and c.create_viewport(w, h) and c.assign_lights(li)
          and c.create_texture_atlas(sc["atlas"], (16, 16)) and c.validate())
    assert ok, c.last_error()


def run(budget=300.0, seed=1, depths=(8, 10, 12), limit=None):     # limit: stop after this many frames (fixed volume; the budget is then a safety net)
    """Returns (frames with a difference, frames)."""
    rng = np.random.default_rng(seed)
    scenes = {}
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget and (limit is None or n < limit):
        depth = int(rng.choice(depths))
        if depth not in scenes:
            sc = bench.build_scene(depth)
            plain = sc["octree"]
            mats = vrc.Octree(plain.descriptor_buffer, plain.root_index, plain.dim).attach_materials_procedural(depth, seed=1, mirror_period=64)
            holders = []
            for t in (plain, mats):                  # one resident copy of each tree (array, coarse table, boxes) per scene
                hc = vrc.CLCaster()
                assert hc.init(0) and hc.assign_octree(t), hc.last_error()
                holders.append(hc)
            scenes[depth] = (sc, plain, mats, holders)
        sc, plain, mats, holders = scenes[depth]
        dim = sc["dim"]
        use_mats = bool(rng.random() < 0.3)
        tree, holder = (mats, holders[1]) if use_mats else (plain, holders[0])
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 400))
        if rng.random() < 0.2:
            w, h = int(rng.choice([1, 2, 63, 64, 65, 1920])), int(rng.choice([1, 2, 7, 8, 9, 1080]))
        ranks, band = int(rng.integers(1, 9)), int(rng.choice([8, 8, 16, 24, 32, 64, 128]))
        pos = rng.random(3) * (dim * 1.2) - 0.1 * dim
        if rng.random() < 0.6:
            cx, cy = int(min(max(pos[0], 0), dim - 1)), int(min(max(pos[1], 0), dim - 1))
            pos[2] = float(sc["height"][cy, cx]) + rng.random() * dim * 0.3 + 1.5
        cam = (np.array([rng.random() * 3.1 + 0.02, rng.random() * 6.28], dtype=np.float32), pos.astype(np.float32))
        li = sc["lights"].copy()
        li[:, 4:7] = rng.random((8, 3)) * dim * 1.1
        settings = {"max_distance": int(rng.choice([3 * dim, 3 * dim, 200, 5000])), "light_count": int(rng.choice([1, 1, 2, 4])),
                    "stepping_mode": int(rng.integers(0, 2)), "shadow_rays": int(rng.choice([1, 1, 0]))}
        if rng.random() < 0.5:
            settings["jump_min_run"] = int(rng.choice([16, 64, 96, 1 << 24]))
        one = vrc.CLCaster()
        assert one.init(0)
        setup(one, sc, tree, w, h, cam, li, settings, holder)
        assert one.compute(), one.last_error()
        ref = (one.read_image(), one.read_hits(), one.read_image_rgba8(), one.counters())
        g = vrc.CLCaster()
        own = bool(rng.random() < 0.3)
        assert g.init_group([0] * ranks, band_rows=band, own_copies=own) and g.group_size() == ranks, g.last_error()
        setup(g, sc, tree, w, h, cam, li, settings, None if (own or rng.random() < 0.3) else holder)   # (own copies: an upload of its own, fanned out rank by rank)
        assert g.compute(), g.last_error()
        img = np.zeros_like(ref[0])
        pinned = bool(rng.random() < 0.3)
        if pinned:
            vrc.pin_host_buffer(img)
        try:
            g.read_image(out=img)
            same = np.array_equal(img.view(np.uint32), ref[0].view(np.uint32))
        finally:
            if pinned:
                vrc.unpin_host_buffer(img)
        same = same and np.array_equal(g.read_hits(), ref[1]) and np.array_equal(g.read_image_rgba8(), ref[2]) and g.counters() == ref[3]
        # a second frame of the same handle (the slots, the counters and the staging buffers are reused)
        assert g.compute(), g.last_error()
        same = same and np.array_equal(g.read_hits(), ref[1]) and g.counters() == ref[3]
        n += 1
        if not same:
            bad += 1
            print("MISMATCH frame", n - 1, "depth", depth, (w, h), "ranks", ranks, "band", band, "own copies", own, "pinned", pinned, settings,
                  "attachments", tree is mats, cam[1].tolist(), cam[0].tolist(), flush=True)
        del one, g
    print(f"group soak: {n} frames (1..700 x 1..400 and a few special sizes, 1-8 ranks on one GPU, bands of 8..128 rows, both modes, 1-4 lights, "
          f"attachments, own copies, pinned / pageable read-back) at depths {list(depths)}: {bad} differ from the single handle; {time.time() - t0:.0f} s")
    return bad, n


if __name__ == "__main__":
    sys.exit(1 if run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)[0] else 0)
