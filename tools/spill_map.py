#!/usr/bin/env python3
"""Where the spills are: scratch loads / stores of the SVO kernel instances by phase of the round loop (the `; VRC_MARK` comments of
raycast_kernel.hip), from the gfx950 assembly (cross-compiles, no GPU).  A spill costs where it is EXECUTED: one in the event phase
runs ~28 times per wave, one in the hit block twice, one in the prologue once -- the byte size of the scratch segment says nothing
about that (round 6: the multi-light instance went from 3.66 to 3.19 ms while its segment GREW from 84 to 100 bytes).
python tools/spill_map.py [instance-substring ...]   e.g.  ILb1ELb1ELb1ELi3ELb1ELb1E = <true, true, true, 3, true, true>   (default: the two headline instances)
VRC_EXTRA_FLAGS="-D..." adds compiler flags.  tests/test_kernel_resources.py asserts that the hot phases hold none."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

HOT_PHASES = ("jump_rows_begin", "jump_rows_end", "jump_block_begin", "jump_block_end", "safe_begin", "single_begin", "exact_begin", "event_begin")


def assembly(extra_flags=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + list(extra_flags) + ["-S", "--cuda-device-only", "-o", out, os.path.join(g.CSRC, "raycast_kernel.hip")],
                              stderr=subprocess.DEVNULL)
        return open(out).read().splitlines()


def spill_map(text, instance):
    """-> ordered {phase: [scratch loads, scratch stores, other instructions]} of raycast_svo_kernel<instance>, None if absent."""
    start = next((i for i, l in enumerate(text) if l.startswith("_ZN3vrc18raycast_svo_kernel" + instance + "EEvNS_13RaycastParamsE:")), None)
    if start is None:
        return None
    phase, rows = "prologue", {}
    for l in text[start:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.search(r"; VRC_MARK (\w+)", l)
        if m:
            phase = m.group(1)
        row = rows.setdefault(phase, [0, 0, 0])
        t = l.strip()
        if t.startswith("scratch_load"): row[0] += 1
        elif t.startswith("scratch_store"): row[1] += 1
        elif t and not t.startswith((";", ".", "_")) and not t.endswith(":"): row[2] += 1
    return rows


if __name__ == "__main__":
    want = sys.argv[1:] or ["ILb1ELb0ELb1ELi3ELb1ELb1E", "ILb1ELb1ELb1ELi3ELb1ELb1E"]
    text = assembly([a for a in os.environ.get("VRC_EXTRA_FLAGS", "").split() if a])
    for w in want:
        rows = spill_map(text, w)
        if rows is None:
            print(w, "not found"); continue
        print(f"raycast_svo_kernel<{w}> (kJump, kMulti, kTuned, kLdsRows, kCoarse, kBox): phase, scratch loads, scratch stores, other instructions (static)")
        for ph, r in rows.items():
            print(f"  {ph:18s} {r[0]:4d} {r[1]:4d} {r[2]:6d}" + ("   <-- hot phase" if ph in HOT_PHASES and (r[0] or r[1]) else ""))
