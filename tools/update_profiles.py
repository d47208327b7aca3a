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
    "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-build   (exact mode, the headline kernel)\n"
    + open(g("stats", "stats_kernel_stats.csv")).read()
    + "\n# rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/frames.py --mode 1 --frames 20   (stepping mode 1, node-exit jumps)\n"
    + open(g("stats_b", "stats_kernel_stats.csv")).read()
    + "\n# bench.py JSON line of the un-profiled run in the same gpurun call\n" + line + "\n")
txt0, v0 = pmc(g("pmc_mode0_summary.txt"))
txt1, v1 = pmc(g("pmc_mode1_summary.txt"))
for name, txt, v, kern in ((f"{rnd}_pmc_exact.txt", txt0, v0, "raycast_svo_kernel<true, false, true> (exact mode, closed-form jumps on: the headline kernel)"), (f"{rnd}_pmc_mode_b.txt", txt1, v1, "raycast_jump_kernel<false>")):
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
hbm = int((2 * v0["FETCH_SIZE"] + v0["WRITE_SIZE"]) * 1024)
json.dump({"kernel_source_hash": bench.kernel_source_hash(), "hbm_bytes_per_launch": hbm, "fetch_size_kib_raw": v0["FETCH_SIZE"],
           "write_size_kib_raw": v0["WRITE_SIZE"],
           "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section: rocprofv3 reports half of wide coalesced reads; applied to all reads = upper bound), WRITE_SIZE as reported",
           "source": f"profiles/{rnd}_pmc_exact.txt (separate --pmc passes, kernel raycast_svo_kernel<true, false, true>, headline workload)",
           "valu_insts_per_launch": int(v0["SQ_INSTS_VALU"]), "valu_source": f"SQ_INSTS_VALU, profiles/{rnd}_pmc_exact.txt",
           "mode_b": {"hbm_bytes_per_launch": int((2 * v1["FETCH_SIZE"] + v1["WRITE_SIZE"]) * 1024), "valu_insts_per_launch": int(v1["SQ_INSTS_VALU"]),
                      "source": f"profiles/{rnd}_pmc_mode_b.txt"}},
          open(P("traffic_latest.json"), "w"), indent=1)
print("exact: Mrays/s", b["value"], "kernel ms", b["roofline"]["kernel_ms_avg"], "HBM MB", hbm / 1e6, "VALU G", v0["SQ_INSTS_VALU"] / 1e9)
print("mode B:", b.get("mode_b_node_exit_jumps", {}).get("kernel_ms_avg"), "ms, VALU G", v1["SQ_INSTS_VALU"] / 1e9)
