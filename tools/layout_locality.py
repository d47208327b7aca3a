#!/usr/bin/env python3
"""SURVEY 7 step 6 / VERDICT r4 item 6: is there anything to gain from re-laying the descriptor array's LOWER levels in Morton /
brick order?  Measures, on the arrays the product renders, how far apart the descriptors of one 4^3 / 8^3 / 16^3-voxel brick lie:
index span of the brick's subtree against the number of descriptors in it (1.0 = contiguous).  CPU only.
    python tools/layout_locality.py [depth ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voxel_raycaster_amd as vrc


def spans(desc, root, depth, strict):
    """-> {brick size: (bricks, mean descriptors per brick, mean span / count, max span / count, bricks with a far pointer inside)}"""
    out = {}
    # iterative post-order walk: (index, level) -> (count, min index, max index) of the subtree
    stack = [(int(root), 0, False)]
    res = {}
    stats = {2: [], 3: [], 4: []}               # bricks of 2^k voxels per axis: nodes at level depth - k
    while stack:
        idx, level, done = stack.pop()
        d = int(desc[idx])
        valid, leaf = (d >> 16) & 0xff, (d >> 24) & 0xff
        kids = []
        if depth - level - 1 >= 1:               # children are descriptors unless they are voxels
            base = idx + (d & 0x7fff)
            if d & 0x8000:
                base = int(desc[base])
            rank = 0
            for k in range(8):
                if valid >> k & 1:
                    if not (leaf >> k & 1):
                        kids.append(base + rank)
                    rank += 1
        if not done:
            stack.append((idx, level, True))
            for c in kids:
                stack.append((c, level + 1, False))
            continue
        cnt, lo, hi = 1, idx, idx
        b_cnt, b_lo, b_hi = 0, 1 << 62, -1       # the brick's BODY: everything below its root (the root itself sits in its parent's child block)
        for c in kids:
            c_cnt, c_lo, c_hi = res.pop(c)
            cnt += c_cnt; lo = min(lo, c_lo); hi = max(hi, c_hi)
            b_cnt += c_cnt; b_lo = min(b_lo, c_lo); b_hi = max(b_hi, c_hi)
        res[idx] = (cnt, lo, hi)
        k = depth - level
        if k in stats and b_cnt:
            stats[k].append((cnt, hi - lo + 1, b_cnt, b_hi - b_lo + 1))
    for k, rows in stats.items():
        a = np.array(rows, dtype=np.float64)
        if len(a):
            out[1 << k] = dict(bricks=len(a), mean_descriptors=round(float(a[:, 0].mean()), 1), mean_span_over_count=round(float((a[:, 1] / a[:, 0]).mean()), 4),
                               body_mean_span_over_count=round(float((a[:, 3] / a[:, 2]).mean()), 4), body_max_span_over_count=round(float((a[:, 3] / a[:, 2]).max()), 2),
                               body_contiguous_share=round(float((a[:, 3] == a[:, 2]).mean()), 5),
                               body_mean_cache_lines=round(float(np.ceil(a[:, 3] * 8 / 128).mean()), 2))
    return out


if __name__ == "__main__":
    for depth in [int(a) for a in sys.argv[1:]] or [8, 10]:
        for strict, name in ((False, "brick layout (device builder / host emitter, no page headers)"), (True, "reference layout (Octree.cpp:245-319: page headers, far pointers)")):
            tree, _ = vrc.shell_terrain_ex(depth, thickness=2) if strict else vrc.shell_terrain(depth, seed=1, thickness=2, strict_reference=False)
            desc = tree.descriptor_buffer
            print(f"depth {depth}, {name}: {desc.size} slots")
            for size, row in spans(desc, tree.root_index, depth, strict).items():
                print(f"  bricks of {size}^3 voxels: {row}")
