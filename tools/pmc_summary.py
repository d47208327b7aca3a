#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes: per-counter average per dispatch of a kernel."""
import csv, glob, os, sys, collections
root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "raycast_svo_kernel"
for f in sorted(glob.glob(os.path.join(root, "*", "*_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    per_dispatch = collections.defaultdict(dict)
    for row in csv.DictReader(open(f)):
        if kern not in row["Kernel_Name"]:
            continue
        per_dispatch[row["Dispatch_Id"]][row["Counter_Name"]] = per_dispatch[row["Dispatch_Id"]].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    for d, cs in per_dispatch.items():
        for k, v in cs.items():
            acc[k].append(v)
    for k, v in acc.items():
        print(f"{os.path.basename(os.path.dirname(f)):8s} {k:28s} dispatches={len(v)} avg={sum(v)/len(v):.6g}")
