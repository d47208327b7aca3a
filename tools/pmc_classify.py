#!/usr/bin/env python3
"""Which per-class VALU counter (SQ_INSTS_VALU_ADD_F32, ..._INT32, ..._CVT, ...) counts which opcode on gfx950: rocprofv3
--pmc passes over tools/ubench/valu_issue, whose kernels each issue one opcode (k_fma, k_minf, k_cnds, ...).  Prints, per
ubench kernel, the share of its VALU instructions that each class counter saw.
    python tools/pmc_classify.py <dir with */*_counter_collection.csv>"""
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*", "*_counter_collection.csv"))):
    per = collections.defaultdict(lambda: collections.defaultdict(float))       # this pass: kernel -> counter -> sum
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if k.startswith("k_"):
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, cs in per.items():                      # every pass carries SQ_INSTS_VALU: a class counter's share of ITS pass
        tot = cs.get("SQ_INSTS_VALU", 0.0)
        for c, v in cs.items():
            if c != "SQ_INSTS_VALU" and tot:
                acc[k][c] = v / tot
        acc[k]["SQ_INSTS_VALU"] = 1.0
for k in sorted(acc):
    tot = acc[k].get("SQ_INSTS_VALU", 0.0)
    if not tot:
        continue
    shares = {c: v / tot for c, v in acc[k].items() if c != "SQ_INSTS_VALU" and v / tot > 0.02}
    print(f"{k:18s} " + (", ".join(f"{c.replace('SQ_INSTS_VALU_', '')} {s:.2f}" for c, s in sorted(shares.items())) or "(no class counter)"))
