#!/usr/bin/env python3
"""Copies the summaries of a tools/gpu_bench_profile.sh run (gpurun_out/<tag>/, gpurun_out/pmc_<tag>/) into profiles/:
python tools/update_profiles.py <tag> "<kernel label>" """
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, label = sys.argv[1], sys.argv[2]
g = lambda *p: os.path.join(ROOT, "gpurun_out", *p)
stats = open(g(tag, "stats", "stats_kernel_stats.csv")).read()
bench = open(g(tag, "bench.json")).read().strip().splitlines()[-1]
open(os.path.join(ROOT, "profiles", "r01_kernel_stats.txt"), "w").write(
    f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline   (MI355X, {label})\n"
    + stats + "\n# bench.py JSON line of the un-profiled run in the same gpurun call (its roofline.traffic / valu_issue fields quote the previous PMC file)\n" + bench + "\n")
pm = open(g(tag, "pmc_summary.txt")).read()
vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)\s+dispatches=\d+ avg=([0-9.e+]+)", pm)}
util = vals["SQ_THREAD_CYCLES_VALU"] / (64 * vals["SQ_ACTIVE_INST_VALU"])
old = open(os.path.join(ROOT, "profiles", "r01_pmc_summary.txt")).read()
history = old[old.index("\n# the same counters for earlier kernels"):] if "# the same counters for earlier kernels" in old else ""
open(os.path.join(ROOT, "profiles", "r01_pmc_summary.txt"), "w").write(
    "# rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline\n"
    f"# separate passes (tools/gpu_pmc.sh); per-dispatch averages of raycast_svo_kernel<false, false, true> (tools/pmc_summary.py), {label}\n"
    f"# VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = {util:.2f} (lanes that take empty steps in a safe run count as active)\n"
    + pm + history)
hbm = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
json.dump({"hbm_bytes_per_launch": hbm, "fetch_size_kib_raw": vals["FETCH_SIZE"], "write_size_kib_raw": vals["WRITE_SIZE"],
           "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section: rocprofv3 reports half of wide coalesced reads; applied to all reads = upper bound), WRITE_SIZE as reported (99.5 MB float4 frame + hit records, the rest is the kernel's register-spill scratch)",
           "source": "profiles/r01_pmc_summary.txt (separate --pmc passes, kernel raycast_svo_kernel<false, false, true>, headline workload)",
           "valu_insts_per_launch": int(vals["SQ_INSTS_VALU"]), "valu_source": "SQ_INSTS_VALU, profiles/r01_pmc_summary.txt"},
          open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
b = json.loads(bench)
print("Mrays/s", b["value"], "ms/step", b["ms_per_step"], "kernel", b["roofline"]["kernel_ms_avg"], "HBM MB", hbm / 1e6, "VALU G", vals["SQ_INSTS_VALU"] / 1e9,
      "SALU G", vals["SQ_INSTS_SALU"] / 1e9, "util", round(util, 3))
