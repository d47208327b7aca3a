#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow) into a depth-12 SVO at 1920x1080 on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one vrc_compute(): one pass of the raycast kernel over this rank's rows of the frame,
synchronous like the reference's compute() (src/CLCaster.cpp:224-228, clFinish :970).  All inputs
(SVO, ray table, atlas) are resident in HBM before the timed region; the frame is the production frame (the image and
nothing else, like the reference: setting hit_records = 0).  Prints ONE JSON line on rank 0.

N = 1: BASELINE.json configs[2] -- 4096^3 shell-terrain SVO (seed 1), 1920x1080, primary rays +
one shadow ray per hit + Blinn-Phong + atlas, the reference's own ray table.
N > 1, --scaling weak (default): the frame grows to 1920 x (1080*N) rows over the SAME field of view (vertical
supersampling xN, host-supplied ray table), row-tiled in interleaved 8-row bands, SVO replicated on
every GPU, no data-path collective.  torch.distributed (RCCL) is used only for the barrier and the
MAX/SUM reductions of the timing protocol.
N > 1, --scaling strong: the frame stays W x H (the reference's own ray table) and its 8-row bands are dealt to the N ranks.

The other BASELINE configs are one command each (they are parity-test cases, not the headline line):
    configs[1]  python bench.py --depth 10 --shadow-rays 0
    configs[3]  torchrun ... bench.py --gpus 8 --scaling strong --width 3840 --height 2160 --lights 2
    configs[4]  torchrun ... bench.py --gpus 8 --scaling strong --depth 16 --thickness 33 --width 7680 --height 4320 --lights 4
(depth >= 14: every rank builds the tree in its own HBM with the device builder, ~6 s for the 198 GB scene.)
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC: RCCL would fail with
# hipIpcGetMemHandle: invalid argument); exported on the boxes already, set here for a launcher that drops the environment
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PREWARM_FRAMES = 40     # untimed frames rendered during set-up so that the GPU clocks have settled before the W warm-up steps
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
# VALU issue peak: 256 CUs x 4 SIMD-32, one wave64 instruction per 2 cycles per SIMD, 2.4 GHz (MI355X_MICROARCH.md)
VALU_PEAK_GINST_S = 256 * 4 * 2.4 / 2.0
KERNEL_SOURCES = ("raycast_kernel.hip", "raycast_jump_kernel.hip", "raycast_common.hpp", "safe_run.hpp", "exact_jump.hpp", "vrc_params.h",
                  "empty_boxes.hip")      # (the boxes decide what the exact kernel reads: a changed builder orphans the traffic figure too)


def kernel_source_hash() -> str:
    """Identifies the kernel a committed PMC measurement belongs to (profiles/traffic_latest.json carries it)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        text = open(os.path.join(ROOT, "voxel-raycaster_amd", "csrc", name), "r").read()
        # comments and blank space are not code: a reworded comment must not orphan a measurement
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = "\n".join(l for l in (re.sub(r"//.*$", "", ln).rstrip() for ln in text.splitlines()) if l)
        h.update(text.encode())
    # ... and what the sources were compiled with: the flags of __graft_entry__.build() and the ROCm release (a file, not a
    # spawned `hipcc --version`: this runs inside GPU-initialised and profiled processes)
    import __graft_entry__ as graft
    h.update(" ".join(f for f in graft.HIP_FLAGS if not f.startswith("-I")).encode())
    try:
        h.update(open("/opt/rocm/.info/version").read().strip().encode())
    except OSError:
        h.update(b"rocm-unknown")
    return h.hexdigest()[:16]


# --------------------------------------------------------------------------- scene
def build_scene(depth: int, seed: int = 1):
    """Scene "shell-terrain(depth, seed)" + camera/light/atlas of SURVEY 8(d).  The camera height is
    raised to the first voxel whose get_oct_vox bias (ray_caster_kernel.cl:353-354) is zero, so the
    reference-exact bias term stays active but does not shear the picture."""
    import voxel_raycaster_amd as vrc
    dim = 1 << depth
    octree, height = vrc.shell_terrain(depth, seed=seed, thickness=2, strict_reference=False)
    cx, cy = dim // 2, dim // 8
    z0 = int(height[cy, cx]) + max(dim // 16, 4)
    cz = None
    for z in range(z0, dim):
        found, res, sub = octree.GetVoxel((cx, cy, z))
        if all((s - v) * res // 2 == 0 for s, v in zip(sub, (cx, cy, z))) and not found:
            cz = z
            break
    if cz is None:
        cz = z0
    # SURVEY 8d's camera as written: h(D/2, D/8) + D/16 + 0.29 (the reference's octree bias is NOT zero there)
    survey_pos = np.array([cx + 0.37, cy + 0.41, int(height[cy, cx]) + dim // 16 + 0.29], dtype=np.float32)
    lights = np.zeros((8, 10), dtype=np.float32)
    lights[0] = [0.01, 0.01, 0.01, 0.2, dim / 4, dim / 4, 3 * dim / 4, -1.0, -1.0, -1.5]
    # lights 1-3: only the multi-light extension reads them (setting light_count; BASELINE configs[3]/[4] geometry)
    lights[1] = [0.02, 0.01, 0.00, 0.2, 3 * dim / 4 + 0.3, dim / 2 + 0.2, 5 * dim / 8 + 0.1, 0.0, 0.0, -1.0]
    lights[2] = [0.00, 0.01, 0.02, 0.2, dim / 2 + 0.4, 7 * dim / 8 + 0.1, dim / 2 + 0.3, 0.0, 0.0, -1.0]
    lights[3] = [0.01, 0.02, 0.01, 0.2, dim / 8 + 0.2, 5 * dim / 8 + 0.3, 7 * dim / 8 + 0.4, 0.0, 0.0, -1.0]
    return dict(depth=depth, dim=dim, octree=octree, height=height,
                cam_pos=np.array([cx + 0.37, cy + 0.41, cz + 0.29], dtype=np.float32),
                cam_dir=np.array([2.0, 1.5708], dtype=np.float32), survey_cam_pos=survey_pos,
                camera_note=f"SURVEY 8d pose (D/2+0.37, D/8+0.41, h(D/2,D/8)+D/16+0.29) raised by {cz - int(survey_pos[2])} voxels to the first "
                            "voxel whose get_oct_vox bias (ray_caster_kernel.cl:353-354) is zero: the reference-exact bias term stays "
                            "active (octree_bias = 1) but adds nothing; survey_camera is the pose as written",
                lights=lights, atlas=vrc.synthetic_atlas(256, 256))


def device_scene_header(depth: int, thickness: int = 2):
    """Scenes that only exist in HBM (depth >= 14; BASELINE configs[4] is depth 16, thickness 33: 24.7 G descriptors = 198 GB):
    camera, lights and atlas by the same conventions; the tree itself is built by every rank on its own GPU (make_caster).
    The camera is SURVEY 8d's as written (no host copy of the tree to search for a bias-free voxel), bias active."""
    import voxel_raycaster_amd as vrc
    dim = 1 << depth
    _, hi = vrc.shell_column(depth, dim // 2, dim // 8, thickness=thickness)
    small = build_scene(8)
    lights = small["lights"].copy()
    lights[:, 4:7] *= dim / 256.0
    pos = np.array([dim / 2 + 0.37, dim / 8 + 0.41, hi + dim // 16 + 0.29], dtype=np.float32)
    return dict(depth=depth, dim=dim, octree=None, device_built=True, thickness=thickness, cam_pos=pos, survey_cam_pos=pos,
                cam_dir=np.array([2.0, 1.5708], dtype=np.float32), lights=lights, atlas=small["atlas"],
                camera_note="SURVEY 8d pose as written, the reference's octree bias active (octree_bias = 1)")


def supersampled_table(width: int, height: int, n: int) -> np.ndarray:
    """Ray table for a width x (height*n) frame covering the field of view of the reference's
    width x height viewport (CLCaster.cpp:233-275): vertical pixel pitch 1/n."""
    s157, c157 = math.sin(1.57), math.cos(1.57)
    x = np.arange(-(width // 2), width - width // 2, dtype=np.float64)
    y = (np.arange(height * n, dtype=np.float64) - (height * n) // 2) / n
    X, Y = np.meshgrid(x, y)
    rx = (Y * s157 + (-800.0) * c157).astype(np.float32)
    ry = X.astype(np.float32)
    rz = (Y * c157 - (-800.0) * s157).astype(np.float32)
    ln = np.sqrt(rx * rx + ry * ry + rz * rz)
    t = np.zeros((height * n, width, 4), dtype=np.float32)
    t[..., 0], t[..., 1], t[..., 2] = rx / ln, ry / ln, rz / ln
    return t


def _scene_tree(c, sc, octree_file, tree_from=None):
    """The SVO into this caster's HBM: uploaded from the host array, streamed from rank 0's file, built on the device -- or
    adopted from a caster on the same GPU that already holds it (one array, one coarse table, one set of empty boxes)."""
    if tree_from is not None:
        return c.assign_octree_from(tree_from)
    if sc.get("device_built"):
        info, _ = c.build_shell_terrain(sc["depth"], 1, sc["thickness"], 2)
        sc["n_desc"] = int(info["n_descriptors"])
        return True
    if octree_file is not None:
        return c.assign_octree_file(octree_file) == sc["dim"]
    return c.assign_octree(sc["octree"])


def make_caster(sc, width, height, device, table=None, shadow_rays=1, light_count=1, row_slice=None, octree_file=None,
                hit_records=1, tree_from=None):
    import voxel_raycaster_amd as vrc
    c = vrc.CLCaster()
    if not c.init(device):
        raise RuntimeError("vrc_create failed: no MI355X visible (there is no CPU fallback)")
    if row_slice is not None and not c.set_row_slice(*row_slice):      # (rank, world, band_rows): buffers hold 1/world of the frame
        raise RuntimeError("set_row_slice failed: " + c.last_error())
    ok = (c.add_to_settings_buffer("octree_dimensions", "OCTDIM", sc["dim"])
          and c.add_to_settings_buffer("using_octree", "OCTENABLED", 0)
          and c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", 3 * sc["dim"])
          and c.add_to_settings_buffer("shadow_rays", "SHADOW_RAYS", shadow_rays)
          and c.add_to_settings_buffer("light_count", "LIGHT_COUNT", light_count)
          and c.add_to_settings_buffer("hit_records", "HIT_RECORDS", hit_records)
          and _scene_tree(c, sc, octree_file, tree_from)
          and c.assign_camera(sc["cam_dir"], sc["cam_pos"])
          and (c.create_viewport(width, height) if table is None else c.create_viewport_table(table))
          and c.assign_lights(sc["lights"])
          and c.create_texture_atlas(sc["atlas"], (16, 16)))
    # validate() is where the one-off cost of a tree is paid (coarse table + empty boxes, vrc_prepare): timed, reported in tree_state
    t0 = time.perf_counter()
    ok = ok and c.validate()
    c._validate_seconds = time.perf_counter() - t0
    if not ok:
        raise RuntimeError("bench setup failed: " + c.last_error())
    return c


# --------------------------------------------------------------------------- distributed helpers
def barrier(local_rank: int):
    """dist.barrier on this rank's GPU (NCCL wants to be told the device; gloo has none)."""
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[local_rank])
    else:
        dist.barrier()


def reduce_over_ranks(rays: int, seconds: float):
    """(SUM of rays, MAX of seconds) over all ranks; identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return rays, seconds
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    r = torch.tensor([float(rays)], dtype=torch.float64, device=dev)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(r, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(round(r.item())), float(t.item())


def canonical_counters(c) -> dict:
    """Counters of one frame rendered by the canonical traversal (empty_boxes = 0): SURVEY 8d's N_desc."""
    if not (c.overwrite_setting("empty_boxes", 0) or c.add_to_settings_buffer("empty_boxes", "EMPTY_BOXES", 0)):
        raise SystemExit("bench: " + c.last_error())
    if not c.compute():
        raise SystemExit("compute failed: " + c.last_error())
    ctr = c.counters()
    assert ctr["canonical_reads"]
    assert c.overwrite_setting("empty_boxes", -1)
    return ctr


def kernel_instance(c, sc, lights) -> str:
    """The template instance vrc_api.cpp / raycast_kernel.hip launch_raycast picks for this caster's default settings."""
    m = c.memory_usage2()
    jump = m["coarse_log2"] > 0 and sc["depth"] >= (11 if m["empty_boxes"] else 12)      # (the jump instances need the coarse table)
    levels = sc["depth"] - m["coarse_log2"]                  # stack levels; the boxes add their index array (tools/spill_map.py names the instances)
    rows = 0 if not jump else (3 if levels * (12 if m["empty_boxes"] else 8) + 72 <= 124 else (2 if m["empty_boxes"] and levels * 12 + 48 <= 124 else 0))
    flags = ["true" if jump else "false", "true" if lights > 1 else "false", "true", str(rows), "true" if m["coarse_log2"] > 0 else "false", "true" if m["empty_boxes"] else "false"]
    return "raycast_svo_kernel<" + ", ".join(flags) + ">  (kJump, kMulti, kTuned, kLdsRows, kCoarse, kBox)"


def tree_state(c, first_frame_ms=None, warm_frame_ms=None) -> dict:
    """What the kernels derive from the tree, and where its one-off cost went: validate() builds it (vrc_prepare; the reference pays its
    kernel build there, CLCaster.cpp:157-206), so the first compute() costs a frame."""
    m = c.memory_usage2()
    out = {k: m[k] for k in ("octree_bytes", "coarse_bytes", "coarse_log2", "box_bytes", "empty_boxes", "box_build_seconds", "tree_holders", "note")}
    out["validate_seconds"] = round(getattr(c, "_validate_seconds", 0.0), 4)
    if first_frame_ms is not None:
        out["first_compute_ms_wall"] = round(first_frame_ms, 3)
        out["warm_compute_ms_wall"] = round(warm_frame_ms, 3)
    return out


def algorithmic_bytes(ctr: dict, pixels: int, written: int) -> int:
    """B = 16*P (ray table) + 16*P_written (float4 frame) + 8*N_desc + 16*N_tex + N_map."""
    return 16 * pixels + 16 * written + 8 * ctr["descriptor_reads"] + 16 * ctr["texel_reads"] + ctr["map_reads"]


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sc, width, height, gpu_frame=None, shadow_rays=1, lights=1):
    """The CPU oracle ("port" of the kernel: DDA + atlas + Blinn-Phong + shadow) timed on ALL host cores
    (OpenMP over rows) on the same frame.  The sample is bounded: the whole frame when the host has enough
    cores to finish ~170 core-seconds of work in seconds, otherwise every k-th row.  Since the oracle frame
    is there anyway it also checks the GPU frame (checker role only)."""
    from oracle import orc   # cpu_baseline leg only
    cores = usable_cores()
    stride = max(1, int(round(170.0 / (cores * 20.0))))        # whole frame ~170 core-seconds: keep the sample near 20 s
    rays, secs, same, rows_done, n_diff = 0, 0.0, True, 0, 0
    # bands of `cores` rows (one row per thread), spread evenly over the frame
    band = height if stride == 1 else cores
    starts = [0] if stride == 1 else list(range(0, height, band * stride))
    for y in starts:
        y1 = min(height, y + band)
        t0 = time.perf_counter()
        img, _, ctr = orc.raycast(width=width, height=height, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"],
                                  lights=sc["lights"], atlas=sc["atlas"], tile_dim=(16, 16),
                                  descriptors=sc["octree"].descriptor_buffer, root_index=sc["octree"].root_index,
                                  octree_dim=sc["dim"], using_octree=0, max_distance=3 * sc["dim"],
                                  rows=(y, y1), threads=cores, want_hits=False, shadow_rays=shadow_rays, active_lights=lights)
        secs += time.perf_counter() - t0
        rays += ctr["primary_rays"] + ctr["shadow_rays"]
        rows_done += y1 - y
        if gpu_frame is not None:
            d = int((img[y:y1].view(np.uint32) != gpu_frame[y:y1].view(np.uint32)).any(-1).sum())
            n_diff += d
            same = same and d == 0
    return rays, secs, cores, (os.cpu_count() or cores), rows_done, same, n_diff


def ray_cpp_baseline():
    """BASELINE configs[0]: the reference's own CPU caster src/Ray.cpp (restated, its two stubbed lines restored)
    on the 256^3 dense twin of the scene at 640x480, primary rays only, all usable host cores."""
    from oracle import orc   # cpu_baseline leg only
    import voxel_raycaster_amd as vrc
    sc = build_scene(8)
    grid = vrc.shell_terrain_dense(8, seed=1, thickness=2)
    cores = usable_cores()
    t0 = time.perf_counter()
    _, steps = orc.ray_cast_frame(grid, (256, 256, 256), 640, 480, sc["cam_dir"], sc["cam_pos"], threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(640 * 480 / dt / 1e6, 4), "unit": "Mrays/s (primary only)", "cores": cores,
            "workload": "configs[0]: 256^3 dense grid, 640x480, Ray::Cast restated (600-step cap)", "dda_steps": steps}


def shared_scene(depth, rank, world, tag, dist_on=False, thickness=2):
    """The scene is built ONCE: rank 0 builds it and saves the tree (vrc_octree_save); the other ranks stream the file
    straight into their own HBM (vrc_assign_octree_file) and never hold a host copy.  Returns (scene dict, file or None)."""
    import torch.distributed as dist
    import voxel_raycaster_amd as vrc
    if depth >= 14:
        return device_scene_header(depth, thickness), None
    if not dist_on:
        sc = build_scene(depth) if thickness == 2 else None
        if sc is None:
            raise SystemExit("--thickness other than 2 needs the device builder (--depth >= 14)")
        return sc, None
    path = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"vrc_bench_{tag}.svo")
    meta = [None]
    if rank == 0:
        sc = build_scene(depth)
        sc["octree"].Save(path)
        meta[0] = {k: sc[k] for k in ("depth", "dim", "cam_pos", "cam_dir", "lights", "survey_cam_pos", "camera_note")}
        meta[0]["n_desc"] = int(sc["octree"].descriptor_buffer.size)
    dist.broadcast_object_list(meta, src=0)
    if rank != 0:
        sc = dict(meta[0], octree=None, atlas=vrc.synthetic_atlas(256, 256))
    sc["n_desc"] = meta[0]["n_desc"]
    return sc, path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lights", type=int, default=1, help="setting light_count (multi-light extension; BASELINE configs[3]: 2, configs[4]: 4)")
    ap.add_argument("--shadow-rays", type=int, default=1, choices=(0, 1), help="0: primary rays only (BASELINE configs[1])")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = the frame grows to H*N rows over the same field of view; strong = the W x H frame is row-tiled over the N ranks")
    ap.add_argument("--thickness", type=int, default=2, help="shell thickness of the terrain (device-built scenes, depth >= 14; configs[4]: 33)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-survey-camera", action="store_true",
                    help="skip the survey_camera leg (the rocprofv3 --stats run: every launch of the kernel is then the headline frame)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other_configs leg (BASELINE configs[1], configs[3] geometry and the 4-light headline on this GPU)")
    ap.add_argument("--no-build", action="store_true",
                    help="never spawn a compiler: fail if a library is stale (required under rocprofv3: the profiler's preloaded "
                         "library must not be inherited by child processes of a GPU-initialised program)")
    ap.add_argument("--pmc-traffic", type=str, default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="JSON file with the rocprofv3 PMC measurement of HBM bytes per launch (see DESIGN.md 7)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n = args.gpus
    if world != n and world != 1:
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world}")

    # torch is imported FIRST (the import alone does not touch the GPU): it brings its own bundled HIP runtime, and a process
    # that loads libvrc.so before it ends up with TWO HIP runtimes -- the system's, which libvrc.so would bind to, and torch's --
    # in which every kernel of this benchmark runs 6-7 % slower (measured round 4: 2.35 vs 2.19 ms on the headline frame,
    # tools/bench_bisect.py).  With torch loaded first libvrc.so binds to the runtime that is already there.
    import torch
    import torch.distributed as dist
    # build BEFORE anything touches the GPU (a GPU-initialised process must not spawn compilers); under torchrun the
    # local rank 0 builds, the others meet it at the first barrier below before they import the library
    import __graft_entry__ as graft
    if args.no_build:
        if graft.stale():
            raise SystemExit("bench.py --no-build: libvrc.so / the oracle are stale; run `python __graft_entry__.py` first")
    elif local_rank == 0:
        graft.build()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    # test hook (tests/ and rehearsals on a 1-GPU box only): let several ranks share GPU 0 over gloo so that the
    # N>1 code path -- supersampled ray table, row slices, SUM/MAX reductions -- can be exercised without 8 GPUs
    rehearsal = os.environ.get("VRC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # VRC_BENCH_FORCE_DIST=1 (a 1-GPU box under `torchrun --nproc-per-node 1`): take the distributed code path -- process
    # group on the real nccl (= RCCL) backend, barriers, broadcast_object_list of the scene header, tree handed over
    # through a file, SUM/MAX all_reduce -- with a world of one, so that it has run on hardware at least once
    dist_on = world > 1 or os.environ.get("VRC_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL: barrier + scalar reductions only
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
        barrier(local_rank)                # rank 0's build is finished before any other rank imports the library
    import voxel_raycaster_amd  # noqa: F401  (fails loudly if the HIP library is missing)

    sc, tree_file = shared_scene(args.depth, rank, world, os.environ.get("MASTER_PORT", "0"), dist_on, args.thickness)
    if tree_file and rank == 0:                                # /dev/shm is memory: the file must not outlive a failed run
        import atexit
        atexit.register(lambda: os.path.exists(tree_file) and os.remove(tree_file))
    W, H = args.width, args.height
    strong = args.scaling == "strong" or world == 1
    full_h = H if strong else H * world
    table = None if strong else supersampled_table(W, H, world)
    # N > 1: each rank holds only its row bands of the ray table / frame / hit records (1/N of the frame)
    # the timed frame is the production frame: like the reference it writes the image and nothing else (the 8 x int32 hit
    # record per pixel exists for the parity tests; the same frame with records on is reported as `with_hit_records`)
    c = make_caster(sc, W, full_h, local_rank, table=table, row_slice=None if world == 1 else (rank, world, 8),
                    octree_file=None if rank == 0 else tree_file, hit_records=0, shadow_rays=args.shadow_rays, light_count=args.lights)
    del table
    if dist_on:
        barrier(local_rank)
        if rank == 0 and tree_file and os.path.exists(tree_file):
            os.remove(tree_file)

    # what a caller sees without any pre-warming: the first `steps` frames of this fresh process (clocks still ramping),
    # reported beside `value` as value_no_prewarm
    # SURVEY 8d's N_desc is the CANONICAL traversal's read count -- a property of the workload, not of the kernel.  The product
    # frame is rendered with the tree's empty boxes (setting empty_boxes), which make fewer reads; one untimed
    # frame without them gives the canonical count the algorithmic bytes are priced with.
    # the very first compute() of this caster, wall clock: validate() has built the coarse table and the boxes, so it costs a frame
    # (plus this frame's own first-use allocations and the code object's load)
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    assert c.compute(), c.last_error()
    first_compute_ms = (time.perf_counter() - t_first) * 1e3
    canonical_ctr = canonical_counters(c)
    assert c.compute(), c.last_error()
    ctr0 = c.counters()
    torch.cuda.synchronize()
    tc = time.perf_counter()
    for _ in range(args.steps):
        if not c.compute():
            raise SystemExit("compute failed: " + c.last_error())
    torch.cuda.synchronize()
    cold_dt = time.perf_counter() - tc
    value_no_prewarm = (ctr0["primary_rays"] + ctr0["shadow_rays"]) * args.steps / cold_dt / 1e6      # this rank's rows

    # set-up, not measurement: a fresh process finds the GPU at its idle clocks, and a 3 ms kernel needs a few dozen
    # launches before DVFS settles (measured: the first timed block of a run is 2-3 % slower than the second)
    for _ in range(PREWARM_FRAMES):
        if not c.compute():
            raise SystemExit("compute failed: " + c.last_error())
    for _ in range(args.warmup):
        if not c.compute():
            raise SystemExit("compute failed: " + c.last_error())
    ctr = c.counters()
    rays_per_step = ctr["primary_rays"] + ctr["shadow_rays"]
    c.timing_reset()

    if dist_on:
        barrier(local_rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if not c.compute():
            raise SystemExit("compute failed: " + c.last_error())
    if dist_on:
        barrier(local_rank)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    n_launch, kernel_ms = c.timing()
    total_rays, max_dt = reduce_over_ranks(rays_per_step, dt)
    value = total_rays * args.steps / max_dt / 1e6

    if rank == 0:
        from voxel_raycaster_amd import tiling
        local_pixels = W * len(tiling.rows_of_rank(full_h, rank, world, 8))
        written = local_pixels - ctr["unwritten_pixels"]
        bytes_per_launch = algorithmic_bytes(canonical_ctr, local_pixels, written)
        avg_kernel_s = kernel_ms / max(n_launch, 1) / 1e3
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        # HBM bytes and VALU instructions per launch come from rocprofv3 PMC passes (tools/gpu_profile.sh: counters cannot be
        # read inside this process).  The committed measurement is stamped with the hash of the kernel sources it was
        # taken on; a different kernel => null, never a stale number.
        traffic, issue, pmc_note, traffic_split = None, None, "no PMC measurement for this workload", None
        headline = (args.depth, W, H, args.lights, args.shadow_rays, args.thickness) == (12, 1920, 1080, 1, 1, 2)
        if world == 1 and headline and args.pmc_traffic and os.path.exists(args.pmc_traffic):
            pmc = json.load(open(args.pmc_traffic))
            if pmc.get("kernel_source_hash") == kernel_source_hash():
                traffic = pmc.get("hbm_bytes_per_launch")
                pmc_note = pmc.get("source")
                # what the one `traffic` number is made of: reads (FETCH_SIZE, doubled per the guide's gfx950 correction), writes
                # (WRITE_SIZE), and the part of the writes that is the frame itself -- the rest of the writes is scratch and the
                # jump tables' rows written through
                if pmc.get("fetch_size_kib_raw") is not None:
                    traffic_split = {"fetch_bytes": int(2 * pmc["fetch_size_kib_raw"] * 1024), "write_bytes": int(pmc["write_size_kib_raw"] * 1024),
                                     "frame_bytes": int(16 * written),
                                     "write_over_frame": round(pmc["write_size_kib_raw"] * 1024 / max(16 * written, 1), 2),
                                     "algorithmic_descriptor_bytes": int(8 * canonical_ctr["descriptor_reads"])}
                valu = pmc.get("valu_insts_per_launch")
                if valu:
                    # the float recurrence of ray_caster_kernel.cl:558-559 (min, three masks, three fused updates = 10 wave64
                    # instructions per 64 lane steps) is what any kernel that steps voxel by voxel must issue at least; the
                    # closed-form jumps (exact_jump.hpp) are how this one gets below 12 per step on average
                    floor = ctr["steps"] / 64.0 * 10.0
                    rate = valu / avg_kernel_s / 1e9
                    issue = {"valu_insts_per_launch": int(valu), "per_voxel_stepping_floor_insts": int(floor),
                             "valu_over_stepping_floor": round(valu / floor, 4),
                             "achieved_ginst_s": round(rate, 1), "peak_ginst_s": VALU_PEAK_GINST_S,
                             "frac": round(rate / VALU_PEAK_GINST_S, 4),
                             "note": "frac counts instructions; frac_time_weighted weighs every instruction class with its measured issue interval"}
                    tw = pmc.get("valu_time_weighted")
                    if tw:
                        # share of the launch's SIMD cycles in which a VALU instruction is being issued: per-class counts
                        # (SQ_INSTS_VALU_<class>) x the issue interval tools/ubench/valu_issue.hip measures for the class
                        issue.update({"frac_time_weighted": tw["frac_time_weighted"], "frac_time_weighted_range": tw["frac_time_weighted_range"],
                                      "issue_cycles_per_launch": int(tw["issue_cycles_mid"]), "simd_cycles_per_launch": int(tw["simd_cycles_per_launch"]),
                                      "classes": tw["classes"], "time_weighted_source": tw["source"]})
            else:
                pmc_note = "profiles/traffic_latest.json was measured on other kernel sources: re-run tools/gpu_profile.sh + tools/update_profiles.py"
        n_desc = int(sc["octree"].descriptor_buffer.size) if sc.get("octree") is not None else int(sc.get("n_desc", 0))
        rehearsal_note = ("ranks share GPU 0 over gloo (VRC_BENCH_REHEARSAL): a rehearsal of the N > 1 code path, not a scaling measurement"
                          if rehearsal else None)
        out = {
            "metric": metric_name(args, W, H),
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(max_dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak" if (world == 1 or not strong) else "strong",
            "value_no_prewarm": round(value_no_prewarm, 3) if world == 1 else None,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name(args, sc, W, H, world, strong),
                       "descriptors": n_desc,
                       "rays_per_step": int(total_rays), "parallelism": f"row-slices x{world}, SVO replicated",
                       "camera": {"position": [float(v) for v in sc["cam_pos"]], "direction_inclination_azimuth": [float(v) for v in sc["cam_dir"]],
                                  "octree_bias": 1, "note": sc.get("camera_note", "")},
                       "lights": args.lights, "shadow_rays": args.shadow_rays,
                       "stepping": "exact per-voxel DDA, bit-identical to the oracle (oracle/vrc_oracle.c); oracle vs the reference's own gfx950 "
                                   "build of kernels/ray_caster_kernel.cl: integer decisions (hit voxel, face, material, step counts) equal, RGB within "
                                   "1e-5 relative on 99.99 % of shaded pixels (the OpenCL library's 1-2 ulp normalize, DESIGN.md 2).  Long empty runs in "
                                   "closed form (exact_jump.hpp: same float sequence and iteration count), setting jump_min_run; empty nodes widened to "
                                   "the empty boxes the tree's device-side pass found around them (empty_boxes.hip, setting empty_boxes): fewer node "
                                   "events, the same steps",
                       "prewarm_frames": PREWARM_FRAMES,
                       "frame": "production frame: image only, like the reference (hit_records = 0); with_hit_records is the same frame plus the parity records",
                       "multi_gpu": ("no N > 1 hardware measurement is recorded in this repository (SCALE_r01..r03 were skipped: no 8-GPU node); "
                                     "an N > 1 line is only a measurement when the driver launched it on N distinct GPUs"
                                     + ("; THIS line: " + rehearsal_note if rehearsal_note else ""))},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_split": traffic_split,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "kernel_ms_avg": round(avg_kernel_s * 1e3, 4), "kernel": "raycast_svo_kernel",
                         "binding_roof": "valu_issue",
                         "note": "exact-parity traversal is bound by VALU issue and by the latency of its dependent chains (DESIGN.md 4), not by "
                                 "HBM: frac is kept on HBM as BASELINE defines it, valu_issue says what the kernel issues",
                         "descriptor_reads": canonical_ctr["descriptor_reads"],
                         "descriptor_reads_note": "SURVEY 8d's canonical count (one untimed frame with empty_boxes = 0); the timed frames use the "
                                                  "tree's empty boxes and read " + str(ctr["descriptor_reads"]) + " descriptors + table cells",
                         "steps": ctr["steps"], "valu_issue": issue, "kernel_instance": kernel_instance(c, sc, args.lights),
                         "tree_state": tree_state(c, first_compute_ms, max_dt / args.steps * 1e3),
                         "pmc_source": pmc_note, "kernel_source_hash": kernel_source_hash()},
        }
        if world == 1 and not sc.get("device_built") and not args.no_survey_camera:
            out["survey_camera"] = survey_camera_leg(sc, c, args)
        if world == 1 and headline and not args.no_other_configs and not args.no_survey_camera:
            try:
                out["other_configs"] = other_configs(sc, c, local_rank, args)
            except Exception as e:                 # a supplementary leg must never take the headline line down
                out["other_configs"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline and not sc.get("device_built"):
            out.update(supplementary(sc, c, W, H, local_rank, args, rays_per_step))
        if dist_on and world == 1:
            out["config"]["distributed_path"] = f"forced on a world of one ({dist.get_backend()}): barrier, broadcast_object_list, all_reduce executed"
        print(json.dumps(out), flush=True)
    if dist_on:
        barrier(local_rank)
        dist.destroy_process_group()


def metric_name(args, W, H):
    if (args.depth, W, H) == (12, 1920, 1080):
        return "Mrays/s (primary+shadow), 1920x1080 into depth-12 SVO"
    return f"Mrays/s ({'primary+shadow' if args.shadow_rays else 'primary only'}), {W}x{H} into depth-{args.depth} SVO"


def workload_name(args, sc, W, H, world, strong):
    """Names the BASELINE.json config the command line selects (the headline line is configs[2])."""
    key = (args.depth, W, H, args.lights, args.shadow_rays)
    which = {(10, 1920, 1080, 1, 0): "BASELINE configs[1]", (12, 1920, 1080, 1, 1): "BASELINE configs[2]",
             (12, 3840, 2160, 2, 1): "BASELINE configs[3]", (16, 7680, 4320, 4, 1): "BASELINE configs[4]"}.get(key, "other")
    if which == "BASELINE configs[4]" and args.thickness != 33:
        which = "BASELINE configs[4] geometry (thin shell: not the ~200 GB tree; --thickness 33 is)"
    rows = "" if world == 1 else (f", row-tiled over {world} ranks (strong scaling)" if strong else f" x{world} rows (vertical supersampling, weak scaling)")
    shading = ("primary + shadow rays toward " + (f"{args.lights} lights" if args.lights > 1 else "1 light") if args.shadow_rays else "primary rays only")
    built = f", built on the device (thickness {sc['thickness']})" if sc.get("device_built") else ""
    return (f"{which}: {sc['dim']}^3 (depth-{args.depth}) shell-terrain SVO seed 1{built}, {W}x{H}{rows}, {shading} + Blinn-Phong + texture atlas, "
            "max_distance 3*dim")


def survey_camera_leg(sc, c, args):
    """The same frame from SURVEY 8d's camera AS WRITTEN, measured in the same run: there the reference's octree bias
    (ray_caster_kernel.cl:353-354) is not zero and shears the rays (parity at this pose: tests/test_configs_gpu.py
    test_headline_size_frame_with_the_reference_bias_active).  `value` stays on the bias-free pose config.camera describes."""
    import torch
    pos = np.ascontiguousarray(sc["survey_cam_pos"], dtype=np.float32)
    assert c.assign_camera(sc["cam_dir"], pos) and c.validate(), c.last_error()
    try:
        canon = canonical_counters(c)
        for _ in range(5):
            assert c.compute(), c.last_error()
        ctr = c.counters()
        rays = ctr["primary_rays"] + ctr["shadow_rays"]
        c.timing_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            assert c.compute(), c.last_error()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nl, ms = c.timing()
    finally:
        # back to the bench pose.  The viewport is created anew because that also refills the image: like the reference, the
        # kernel leaves pixels whose ray has a zero component untouched (ray_caster_kernel.cl:293-294), so the 753 pixels the
        # bench pose never writes would keep the colours the survey pose gave them
        W, H = c.viewport_size
        assert c.assign_camera(sc["cam_dir"], sc["cam_pos"]) and c.create_viewport(W, H) and c.validate(), c.last_error()
    return {"ms_per_step": round(dt / args.steps * 1e3, 4), "value": round(rays * args.steps / dt / 1e6, 3), "unit": "Mrays/s",
            "kernel_ms_avg": round(ms / max(nl, 1), 4), "rays_per_step": int(rays), "steps": ctr["steps"], "descriptor_reads": canon["descriptor_reads"],
            "camera": {"position": [float(v) for v in pos], "octree_bias": 1, "note": "SURVEY 8d pose as written; the reference's bias term is non-zero here"}}


def other_configs(sc, c, device, args):
    """the BASELINE configs beside the headline, timed by the driver's own run on the one GPU every round --
    configs[1] (depth 10, primary rays only), configs[3]'s geometry (4K, 2 lights: the multi-light instance) and the headline
    scene with 4 lights.  Kernel time from HIP events over `steps` frames each; never part of `value`.  The depth-12 legs adopt
    the headline caster's tree (one array, one table, one set of boxes on the GPU)."""
    legs = []

    def leg(name, scene, q, w, h, lights):
        canon = canonical_counters(q)
        for _ in range(3):
            assert q.compute(), q.last_error()
        ctr = q.counters()
        q.timing_reset()
        for _ in range(args.steps):
            assert q.compute(), q.last_error()
        nl, ms = q.timing()
        k = ms / max(nl, 1) / 1e3
        rays = ctr["primary_rays"] + ctr["shadow_rays"]
        b = algorithmic_bytes(canon, w * h, w * h - canon["unwritten_pixels"])
        legs.append({"config": name, "frame": f"{w}x{h}", "lights": lights, "kernel_ms_avg": round(k * 1e3, 4), "frames": int(nl),
                     "Mrays_s": round(rays / k / 1e6, 1), "rays_per_frame": int(rays), "steps": ctr["steps"],
                     "roofline": {"bound": "hbm", "achieved": round(b / k / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(b / k / 1e9 / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": int(b)},
                     "kernel_instance": kernel_instance(q, scene, lights)})

    sc10 = build_scene(10)
    q = make_caster(sc10, 1920, 1080, device, shadow_rays=0, hit_records=0)
    leg("BASELINE configs[1]: 1024^3 (depth-10) SVO, 1920x1080, primary rays only", sc10, q, 1920, 1080, 1)
    del q
    q = make_caster(sc, 3840, 2160, device, light_count=2, hit_records=0, tree_from=c)
    leg("BASELINE configs[3] geometry on ONE GPU: 4096^3 (depth-12) SVO, 3840x2160, primary + shadow rays toward 2 lights", sc, q, 3840, 2160, 2)
    del q
    q = make_caster(sc, 1920, 1080, device, light_count=4, hit_records=0, tree_from=c)
    leg("headline scene with 4 lights (configs[4]'s light count): 4096^3 (depth-12) SVO, 1920x1080", sc, q, 1920, 1080, 4)
    del q
    return legs


def supplementary(sc, c, W, H, device, args, rays_per_step):
    """N = 1 extras, never part of `value`: production frame without hit records, two frames in flight, the node-exit
    jump mode (SURVEY D1 mode B) with its mismatch statistics, and the CPU baselines."""
    import torch
    out = {}
    # the same frame with the per-pixel hit records the parity tests read (round 1's `value` was measured this way)
    assert c.overwrite_setting("hit_records", 1)
    for _ in range(2):
        assert c.compute(), c.last_error()
    c.timing_reset()
    for _ in range(args.steps):
        assert c.compute(), c.last_error()
    nl, ms = c.timing()
    out["with_hit_records"] = {"kernel_ms_avg": round(ms / nl, 4), "value": round(rays_per_step / (ms / nl) / 1e3, 3), "unit": "Mrays/s (kernel time)"}
    # two frames in flight (a second caster = second HIP stream + its own buffers) hide the kernel's ramp-up and tail;
    # the headline `value` is one frame at a time, like CLCaster::compute
    assert c.overwrite_setting("hit_records", 0)
    c2 = make_caster(sc, W, H, device, hit_records=0, shadow_rays=args.shadow_rays, light_count=args.lights, tree_from=c)
    for _ in range(2):
        assert c2.compute(), c2.last_error()
    torch.cuda.synchronize()
    tp = time.perf_counter()
    pairs = max(args.steps // 2, 1)
    for _ in range(pairs):
        assert c.compute_async() and c2.compute_async()
        assert c.sync() and c2.sync()
    dtp = time.perf_counter() - tp
    out["two_frames_in_flight"] = {"value": round(rays_per_step * 2 * pairs / dtp / 1e6, 3), "unit": "Mrays/s",
                                   "ms_per_frame": round(dtp / (2 * pairs) * 1e3, 4),
                                   "tree_holders": c2.memory_usage2()["tree_holders"],
                                   "note": "the second caster adopts the first one's tree (vrc_assign_octree_from): one array, one table, one set of boxes"}
    del c2
    assert c.overwrite_setting("hit_records", 1) and c.compute(), c.last_error()     # mode_b_report compares hit records
    gpu_frame = c.read_image()
    try:
        out["mode_b_node_exit_jumps"] = mode_b_report(sc, c, W, H, args, gpu_frame)
    except Exception as e:                     # the supplementary mode must never take the headline line down
        out["mode_b_node_exit_jumps"] = {"error": str(e)}
    assert c.overwrite_setting("hit_records", 0)
    rays, secs, used, cores, nrows, same, n_diff = cpu_baseline(sc, W, H, gpu_frame=gpu_frame, shadow_rays=args.shadow_rays, lights=args.lights)
    out["cpu_baseline"] = {"value": round(rays / secs / 1e6, 4), "unit": "Mrays/s", "cores": used, "kind": "port",
                           "sample": f"{nrows} of {H} rows of the same frame, oracle/vrc_oracle.c (scalar C, OpenMP over pixels), "
                                     f"{used} of {cores} host threads; {rays} rays in {secs:.2f} s",
                           "gpu_frame_bit_identical_on_sample": same, "pixels_differing_on_sample": n_diff,
                           "checked_against": "the oracle (oracle/vrc_oracle.c, a CPU restatement that DEFINES the result), not the reference's own "
                                              "build: get_oct_vox and view_light of the oracle are pinned against the reference's compiled functions, its "
                                              "whole kernel is corroborated against the reference's kernel run on the MI355X with two image builtins "
                                              "redirected (RGB within 1e-5 on 99.990 % of the shaded pixels; DESIGN.md 2)",
                           "ray_cpp": ray_cpp_baseline()}
    return out


def mode_b_report(sc, c, W, H, args, exact_frame):
    """SURVEY D1 mode B on the same frame: labelled, supplementary, never `value`.  Its own roofline (same algorithmic
    byte formula, its own counters and kernel time) and how far it is from the exact mode."""
    exact_hits = c.read_hits()
    if not c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 1):
        raise RuntimeError(c.last_error())
    try:
        for _ in range(2):
            if not c.compute():
                raise RuntimeError(c.last_error())
        ctr = c.counters()
        img, hits = c.read_image(), c.read_hits()
        c.overwrite_setting("hit_records", 0)          # timed like the headline: the production frame
        for _ in range(PREWARM_FRAMES):                # the read-backs above left the GPU idle: same clock pre-warm as the headline
            assert c.compute(), c.last_error()
        c.timing_reset()
        for _ in range(args.steps):
            assert c.compute(), c.last_error()
        nl, ms = c.timing()
    finally:
        c.overwrite_setting("stepping_mode", 0)
    pixels = W * H
    b = algorithmic_bytes(ctr, pixels, pixels - ctr["unwritten_pixels"])
    k = ms / nl / 1e3
    same_voxel = (hits[..., :5] == exact_hits[..., :5]).all(-1)
    rel = np.abs(img[..., :3] - exact_frame[..., :3]) / np.maximum(np.abs(exact_frame[..., :3]), 1e-6)
    return {"kernel_ms_avg": round(k * 1e3, 4), "value": round((ctr["primary_rays"] + ctr["shadow_rays"]) / k / 1e6, 3),
            "unit": "Mrays/s (kernel time)",
            "roofline": {"bound": "hbm", "achieved": round(b / k / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(b / k / 1e9 / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": int(b),
                         "descriptor_reads": ctr["descriptor_reads"], "node_steps": ctr["steps"]},
            "vs_exact_mode": {"pixels_same_hit_voxel_face_material": round(float(same_voxel.mean()), 6),
                              "pixels_rgb_within_1e-5": round(float((rel.max(-1) <= 1e-5).mean()), 6)},
            "parity": "bit-exact against its own restatement in oracle/ (tests/test_mode_b_gpu.py); a different float "
                      "sequence from the reference array branch by construction"}


if __name__ == "__main__":
    main()
