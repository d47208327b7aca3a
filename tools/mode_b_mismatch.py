#!/usr/bin/env python3
"""Mode B (node-exit jumps) against the exact mode on the headline frame, pixel by pixel, bucketed by cause
(profiles/r04_mode_b_mismatch.txt).  Both frames come from the same caster; hit records on.
    python tools/mode_b_mismatch.py [--depth 12]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=12)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
sc = bench.build_scene(a.depth)
c = bench.make_caster(sc, a.width, a.height, 0, hit_records=1)
assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 0)
frames = {}
for mode in (0, 1):
    assert c.overwrite_setting("stepping_mode", mode) and c.compute(), c.last_error()
    frames[mode] = (c.read_image().copy(), c.read_hits().copy(), c.counters())
(ie, he, ce), (ib, hb, cb) = frames[0], frames[1]
P = ie.shape[0] * ie.shape[1]
# hit record: voxel xyz, material, face bits, flags (1 written, 2 shadow cast, 4 shadow hit, 8 left the map | bounces << 4), final step count, descriptor reads
WRITTEN, CAST, SHIT, OOB = 1, 2, 4, 8
has_hit_e, has_hit_b = he[..., 0] >= 0, hb[..., 0] >= 0
same_kind = has_hit_e == has_hit_b
same_hit = same_kind & (he[..., :5] == hb[..., :5]).all(-1)
rel = np.abs(ib[..., :3] - ie[..., :3]) / np.maximum(np.abs(ie[..., :3]), 1e-6)
relmax = rel.max(-1)
alpha_same = np.abs(ib[..., 3] - ie[..., 3]) <= 1e-5 * np.maximum(np.abs(ie[..., 3]), 1e-6)   # (alpha is a shaded float too: same tolerance)
rgb_ok = relmax <= 1e-5
rows = []
def row(name, mask, note=""):
    rows.append((name, int(mask.sum()), mask.sum() / P, note))

row("pixels", np.ones_like(same_hit))
row("primary ray hits nothing in both modes", ~has_hit_e & ~has_hit_b)
row("hit in one mode only", ~same_kind)
row("both hit, same voxel / face / material", same_hit & has_hit_e)
diff_hit = has_hit_e & has_hit_b & ~same_hit
row("both hit, different voxel / face / material", diff_hit)
dv = np.abs(he[..., :3] - hb[..., :3]).sum(-1)
row("  .. the two voxels are face neighbours (|dv|_1 = 1)", diff_hit & (dv == 1), "a grazing ray resolved one voxel apart: accumulated rounding of intersection_t vs origin + t * dir")
row("  .. same voxel, other face", diff_hit & (dv == 0), "edge / corner hits: the exact mode's ties step two axes at once")
row("  .. further apart", diff_hit & (dv > 1))
steps_e = he[..., 6].astype(np.int64)
for lo, hi in ((0, 256), (256, 1024), (1024, 4096), (4096, 1 << 30)):
    m = diff_hit & (steps_e >= lo) & (steps_e < hi)
    tot = has_hit_e & has_hit_b & (steps_e >= lo) & (steps_e < hi)
    rows.append((f"  .. rays of {lo}-{hi if hi < 1 << 30 else 'inf'} iterations (exact mode)", int(m.sum()), m.sum() / max(tot.sum(), 1), "share of the hits of that length"))

# colour of the pixels that hit the same voxel: why do they differ?
sh = same_hit & has_hit_e
row("same hit: RGB and alpha within 1e-5 relative", sh & rgb_ok & alpha_same)
bad = sh & ~(rgb_ok & alpha_same)
row("same hit: RGB or alpha differ", bad)
shadow_flag_diff = ((he[..., 5] ^ hb[..., 5]) & SHIT) != 0
row("  .. in-shadow flag differs (the shadow ray was blocked in one mode only)", bad & shadow_flag_diff, "alpha 0.1 vs lit, RGB untouched: the shadow ray starts at hit_pos, whose in-face coordinates are the differences of accumulated t above; at grazing light the first steps re-hit the surface or not")
oob_diff = ((he[..., 5] ^ hb[..., 5]) & OOB) != 0
row("  .. shadow ray left the map in one mode only", bad & ~shadow_flag_diff & oob_diff)
rest = bad & ~shadow_flag_diff & ~oob_diff
steps_diff = he[..., 6] != hb[..., 6]
row("  .. same flags, iteration count differs", rest & steps_diff, "mode B counts Manhattan steps: every tie of the exact mode (two axes stepping in one iteration) is one iteration fewer there; the fog factor 1 - d/700 (:716) and the shadow cap (:667) read the count")
rest2 = rest & ~steps_diff
row("  .. same flags, same iteration count", rest2, "the face UV is a difference of two intersection_t of magnitude ~1e3 (:592-614): accumulated vs freshly computed t")
for lo, hi, what in ((1e-5, 1e-3, "same texel, rounding of the UV-independent terms"), (1e-3, 1e-1, "fog / specular terms"), (1e-1, 1e9, "another texel of the noise atlas (a UV that crossed a texel boundary)")):
    m = bad & (relmax > lo) & (relmax <= hi)
    rows.append((f"  .. worst channel off by {lo:g} .. {hi:g} relative", int(m.sum()), m.sum() / P, what))
# shadow outcome cross table (same-hit pixels): rows exact mode, columns mode B; outcome = blocked / left the map / step cap
def outcome(h):
    f = h[..., 5]
    return np.where(f & SHIT, 0, np.where(f & OOB, 1, 2))
oe, ob = outcome(he)[sh], outcome(hb)[sh]
cross = [[int(((oe == i) & (ob == j)).sum()) for j in range(3)] for i in range(3)]
big = sh & (relmax > 0.1)
cross_big = [[int(((outcome(he) == i) & (outcome(hb) == j) & big).sum()) for j in range(3)] for i in range(3)]
# ties: how many iterations did the exact mode save on the primary segment?  (mode B's count minus the exact count, same hit)
d_steps = (hb[..., 6].astype(np.int64) - he[..., 6].astype(np.int64))[sh]
hist = {int(k): int(v) for k, v in zip(*np.unique(np.clip(d_steps, -3, 8), return_counts=True))}
out = {"frame": f"depth {a.depth}, {a.width}x{a.height}", "counters_exact": ce, "counters_mode_b": cb,
       "shadow_outcome_cross_table_same_hit [exact: blocked, left map, cap][mode B: blocked, left map, cap]": cross,
       "the same for pixels whose RGB differs by more than 0.1 relative": cross_big,
       "iteration_count_mode_b_minus_exact_same_hit_pixels (clipped to -3..8)": hist}
print(f"# mode B vs exact mode, {out['frame']}, bench camera; hit records of both modes from one caster (tools/mode_b_mismatch.py)")
print("| bucket | pixels | share | note |\n|---|---|---|---|")
for name, n, share, note in rows:
    print(f"| {name} | {n} | {share:.5f} | {note} |")
print(json.dumps(out))
