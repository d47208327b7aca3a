#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / waves per SIMD / LDS of every kernel instance, from hipcc -Rpass-analysis=kernel-resource-usage
(cross-compiles without a GPU).  python tools/resource_usage.py > profiles/rNN_resource_usage.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

rows = []
for src in ("raycast_kernel.hip", "raycast_jump_kernel.hip", "svo_builder_gpu.hip"):
    out = subprocess.run([g.HIPCC] + g.HIP_FLAGS + ["-c", os.path.join(g.CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark: (?:\s*)(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip() or v
            cur = {"kernel": re.sub(r"\(vrc::RaycastParams\)|vrc::", "", name), "file": src}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = v
print("# kernel resource usage, gfx950 (hipcc " + " ".join(f for f in g.HIP_FLAGS if f.startswith("-O") or f.startswith("-f")) + ")")
print("# raycast_svo_kernel<kJump, kMulti, kTuned, kLdsRows, kCoarse, kBox>: kJump = exact closed-form jumps compiled in (96 VGPRs: 5 blocks per")
print("# CU; the others 80: 6), kMulti = multi-light extension, kTuned = scheduling knobs at their defaults (compile-time constants; the")
print("# run-time twins exist once each, with the multi-light code), kLdsRows = rows of the jumps' Euclid-table ring in LDS (3, or 2 beside the")
print("# deeper stacks of the box instances; 0 = the tables live in global memory), kCoarse = the tree's top from the dense table, kBox = empty")
print("# boxes.  Scratch: where it is EXECUTED is what counts -- tools/spill_map.py; the headline instance <true, false, true, 3, true, true> has none.")
print("# raycast_jump_kernel<kMulti, kCoarse> = mode B.  Static LDS only.")
print("| kernel | file | VGPRs | SGPRs | scratch B/lane | waves/SIMD | static LDS B |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| `{r['kernel']}` | {r['file']} | {r.get('VGPRs')} | {r.get('TotalSGPRs')} | {r.get('ScratchSize')} | {r.get('Occupancy')} | {r.get('LDS Size')} |")
