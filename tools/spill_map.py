#!/usr/bin/env python3
"""Where the spills are: scratch loads / stores of the SVO kernel instances by phase of the round loop (the `; VRC_MARK` comments of
raycast_kernel.hip), from the gfx950 assembly (cross-compiles, no GPU).  A spill costs where it is EXECUTED: one in the event phase
runs ~28 times per wave, one in the hit block twice, one in the prologue once.
python tools/spill_map.py [instance-substring ...]   e.g.  ILb1ELb1ELb1ELb1ELb1ELb1E   (default: the two headline instances)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

want = sys.argv[1:] or ["ILb1ELb0ELb1ELb1ELb1ELb1E", "ILb1ELb1ELb1ELb1ELb1ELb1E"]
extra = [a for a in os.environ.get("VRC_EXTRA_FLAGS", "").split() if a]
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + extra + ["-S", "--cuda-device-only", "-o", out, os.path.join(g.CSRC, "raycast_kernel.hip")],
                          stderr=subprocess.DEVNULL)
    text = open(out).read().splitlines()
for w in want:
    start = next((i for i, l in enumerate(text) if l.startswith("_ZN3vrc18raycast_svo_kernel" + w) and l.rstrip().endswith(":") is False and ":" in l), None)
    if start is None:
        print(w, "not found"); continue
    phase, rows, order = "prologue", {}, []
    for l in text[start:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.search(r"; VRC_MARK (\w+)", l)
        if m:
            phase = m.group(1)
        if "Loop Header: Depth=1" in l and phase == "prologue":
            phase = "round_top"
        if phase not in rows:
            rows[phase] = [0, 0, 0]; order.append(phase)
        t = l.strip()
        if t.startswith("scratch_load"): rows[phase][0] += 1
        elif t.startswith("scratch_store"): rows[phase][1] += 1
        elif t and not t.startswith((";", ".", "_")) and not t.endswith(":"): rows[phase][2] += 1
    print(f"raycast_svo_kernel<{w}>: phase, scratch loads, scratch stores, other instructions (static)")
    for ph in order:
        print(f"  {ph:18s} {rows[ph][0]:4d} {rows[ph][1]:4d} {rows[ph][2]:6d}")
