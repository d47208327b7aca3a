#!/usr/bin/env python3
"""Copies the summaries of a tools/gpu_profile.sh run (gpurun_out/<tag>/) into profiles/ (tracked) and stamps
profiles/traffic_latest.json with the hash of the kernel sources the PMC passes were taken on:
    python tools/update_profiles.py <tag> <round, e.g. r02> "<kernel label>" """
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag, rnd, label = sys.argv[1], sys.argv[2], sys.argv[3]
g = lambda *p: os.path.join(ROOT, "gpurun_out", tag, *p)
P = lambda name: os.path.join(ROOT, "profiles", name)
line = open(g("bench.json")).read().strip().splitlines()[-1]
b = json.loads(line)
if b["roofline"]["kernel_source_hash"] != bench.kernel_source_hash():
    sys.exit("the kernel sources changed since this profile run: profile again")


def pmc(path):
    txt = open(path).read()
    return txt, {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)\s+dispatches=\d+ avg=([0-9.e+]+)", txt)}


open(P(f"{rnd}_kernel_stats.txt"), "w").write(
    f"# MI355X, {label}\n"
    "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-survey-camera --no-build   (exact mode: every launch is the headline frame)\n"
    + open(g("stats", "stats_kernel_stats.csv")).read()
    + "\n# rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/frames.py --mode 1 --frames 20   (stepping mode 1, node-exit jumps)\n"
    + open(g("stats_b", "stats_kernel_stats.csv")).read()
    + "\n# bench.py JSON line of the un-profiled run in the same gpurun call\n" + line + "\n")
txt0, v0 = pmc(g("pmc_mode0_summary.txt"))
txt1, v1 = pmc(g("pmc_mode1_summary.txt"))
for name, txt, v, kern in ((f"{rnd}_pmc_exact.txt", txt0, v0, "raycast_svo_kernel<true, false, true, 3, true, true> (exact mode: closed-form jumps, Euclid tables in LDS, coarse table, empty boxes -- the headline kernel)"), (f"{rnd}_pmc_mode_b.txt", txt1, v1, "raycast_jump_kernel<false>")):
    util = v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"])
    hbm = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    open(P(name), "w").write(
        f"# MI355X, {label}\n"
        "# rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 tools/frames.py --mode <0|1> --frames 3   (tools/gpu_profile.sh: separate passes,\n"
        f"# kernel-trace only); per-dispatch averages of {kern} on the headline frame (tools/pmc_summary.py)\n"
        f"# VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = {util:.3f}\n"
        f"# HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB; gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE reports half of wide reads) = {hbm / 1e6:.1f} MB\n"
        f"# L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS) = {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f};  effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time\n"
        + txt)
open(P(f"{rnd}_valu_issue.txt"), "w").write(open(g("valu_issue.txt")).read())


def issue_intervals(path):
    """SIMD issue interval (shader cycles per wave64 instruction) per opcode at 4 waves per SIMD, from the ubench table."""
    out = {}
    for line in open(path):
        m = re.match(r"(v_\S+(?: clamp| vcc| \(sgpr\)| \(vcc\))?)\s+W=1.*?W=4\s+([0-9.]+)", line)
        if m:
            out[m.group(1).strip()] = float(m.group(2))
    return out


def time_weighted_issue(v, iv):
    """VALU issue cycles of one launch = sum over instruction classes of (wave-level count) x (measured issue interval of that
    class).  The class counters do not name every opcode (profiles/<round>_valu_classes.txt says which opcode lands where): INT32
    mixes full-rate adds / logic with the slower shifts, multiplies and bit-field ops, and what no class counter sees (min / max /
    compare / select / move, per the calibration) is `other`; both get a low and a high interval, the estimate is the middle."""
    g = lambda k: v.get(k, 0.0)
    total = g("SQ_INSTS_VALU")
    fixed = [("FMA_F32", g("SQ_INSTS_VALU_FMA_F32"), iv["v_fma_f32"]), ("ADD_F32", g("SQ_INSTS_VALU_ADD_F32"), iv["v_add_f32"]),
             ("MUL_F32", g("SQ_INSTS_VALU_MUL_F32"), iv["v_mul_f32"]), ("TRANS_F32", g("SQ_INSTS_VALU_TRANS_F32"), iv["v_rcp_f32"]),
             ("CVT", g("SQ_INSTS_VALU_CVT"), iv["v_cvt_i32_f32"]),
             ("FMA_F64", g("SQ_INSTS_VALU_FMA_F64"), iv.get("v_fma_f64", 2 * iv["v_fma_f32"])),
             ("MUL_F64", g("SQ_INSTS_VALU_MUL_F64"), iv.get("v_mul_f64", 2 * iv["v_fma_f32"])),
             ("ADD_F64", g("SQ_INSTS_VALU_ADD_F64"), iv.get("v_add_f64", 2 * iv["v_fma_f32"])),
             ("TRANS_F64", g("SQ_INSTS_VALU_TRANS_F64"), 4 * iv["v_rcp_f32"])]
    int32, int64 = g("SQ_INSTS_VALU_INT32"), g("SQ_INSTS_VALU_INT64")
    other = max(0.0, total - sum(c for _, c, _ in fixed) - int32 - int64)
    ranged = [("INT32", int32, iv["v_add_u32"], iv["v_mad_u32_u24"]),
              ("INT64", int64, 2 * iv["v_add_u32"], 2 * iv["v_mad_u32_u24"]),
              ("other (min/max/cmp/select/move ...)", other, iv["v_mov_b32"], iv["v_cndmask_b32 (sgpr)"])]
    base = sum(c * i for _, c, i in fixed)
    lo = base + sum(c * a for _, c, a, _ in ranged)
    hi = base + sum(c * b for _, c, _, b in ranged)
    rows = [dict(cls=n, insts=int(c), cycles_per_inst=i) for n, c, i in fixed] + \
           [dict(cls=n, insts=int(c), cycles_per_inst=[a, b]) for n, c, a, b in ranged]
    return dict(classes=rows, issue_cycles_lo=lo, issue_cycles_hi=hi, issue_cycles_mid=0.5 * (lo + hi), valu_insts=int(total))


iv = issue_intervals(g("valu_issue.txt"))
tw = time_weighted_issue(v0, iv) if "SQ_INSTS_VALU_FMA_F32" in v0 else None
if tw:
    simd_cycles = v0["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0     # cycles of one XCD x the chip's 1024 SIMDs
    tw["simd_cycles_per_launch"] = simd_cycles
    tw["frac_time_weighted"] = round(tw["issue_cycles_mid"] / simd_cycles, 4)
    tw["frac_time_weighted_range"] = [round(tw["issue_cycles_lo"] / simd_cycles, 4), round(tw["issue_cycles_hi"] / simd_cycles, 4)]
    tw["source"] = (f"class counts profiles/{rnd}_pmc_exact.txt (vclass passes) x issue intervals at 4 waves per SIMD of profiles/{rnd}_valu_issue.txt; "
                    f"opcode -> class calibration profiles/{rnd}_valu_classes.txt; denominator GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs")
if os.path.exists(g("valu_classes.txt")):
    open(P(f"{rnd}_valu_classes.txt"), "w").write(
        "# which SQ_INSTS_VALU_<class> counter counts which opcode on gfx950: rocprofv3 --pmc over tools/ubench/valu_issue (one opcode per kernel),\n"
        "# share of the kernel's SQ_INSTS_VALU each class counter saw (tools/pmc_classify.py)\n" + open(g("valu_classes.txt")).read())
# the instances that are not the headline (tools/gpu_profile.sh pmc2)
others = [("c4_4k_2lights", "C4 geometry: depth 12, 3840x2160, 2 lights (multi-light instance)"),
          ("headline_4lights", "headline scene with 4 lights"),
          ("c2_d10_primary", "C2: depth 10, 1920x1080, primary rays only (no jumps below depth 12)"),
          ("headline_no_jumps", "headline frame with jump_min_run off (the plain step-loop instance)")]
blocks = []
for name, title in others:
    if os.path.exists(g(f"pmc_{name}_summary.txt")):
        blocks.append(f"## {title}\n" + open(g(f"pmc_{name}_summary.txt")).read())
if blocks:
    open(P(f"{rnd}_pmc_other_instances.txt"), "w").write(
        "# per-dispatch PMC averages of the SVO kernel's other instances (tools/gpu_profile.sh pmc2: separate --pmc passes, kernel-trace only;\n"
        f"# FETCH_SIZE / WRITE_SIZE in KiB, see {rnd}_pmc_exact.txt for the gfx950 correction); the last line of each block is the frames.py line of the FETCH_SIZE pass\n\n"
        + "\n".join(blocks))
open(P(f"{rnd}_bench_line.json"), "w").write(line + "\n")
hbm = int((2 * v0["FETCH_SIZE"] + v0["WRITE_SIZE"]) * 1024)
json.dump({"kernel_source_hash": bench.kernel_source_hash(), "hbm_bytes_per_launch": hbm, "fetch_size_kib_raw": v0["FETCH_SIZE"],
           "write_size_kib_raw": v0["WRITE_SIZE"],
           "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section: rocprofv3 reports half of wide coalesced reads; applied to all reads = upper bound), WRITE_SIZE as reported",
           "source": f"profiles/{rnd}_pmc_exact.txt (separate --pmc passes, kernel raycast_svo_kernel<true, false, true, 3, true, true>, headline workload)",
           "valu_insts_per_launch": int(v0["SQ_INSTS_VALU"]), "valu_source": f"SQ_INSTS_VALU, profiles/{rnd}_pmc_exact.txt",
           "valu_time_weighted": tw,
           "mode_b": {"hbm_bytes_per_launch": int((2 * v1["FETCH_SIZE"] + v1["WRITE_SIZE"]) * 1024), "valu_insts_per_launch": int(v1["SQ_INSTS_VALU"]),
                      "source": f"profiles/{rnd}_pmc_mode_b.txt"}},
          open(P("traffic_latest.json"), "w"), indent=1)
print("exact: Mrays/s", b["value"], "kernel ms", b["roofline"]["kernel_ms_avg"], "HBM MB", hbm / 1e6, "VALU G", v0["SQ_INSTS_VALU"] / 1e9)
print("mode B:", b.get("mode_b_node_exit_jumps", {}).get("kernel_ms_avg"), "ms, VALU G", v1["SQ_INSTS_VALU"] / 1e9)
