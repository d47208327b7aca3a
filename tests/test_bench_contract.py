"""bench.py's contract (SURVEY 8 row d): the config switches (--lights, --scaling strong, device-built scenes) rehearsed on one GPU,
and the fields the bench line must carry (camera, survey_camera, traffic split, time-weighted issue figure).  The first test runs
without a GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_bench(args, nproc=1, port=29551, timeout=900):
    env = dict(os.environ)
    if nproc > 1:
        env["VRC_BENCH_REHEARSAL"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_names_the_baseline_config_the_flags_select():
    import argparse
    import bench
    sc12 = dict(dim=4096)
    mk = lambda **k: argparse.Namespace(**dict(dict(depth=12, lights=1, shadow_rays=1, thickness=2), **k))
    assert bench.workload_name(mk(), sc12, 1920, 1080, 1, True).startswith("BASELINE configs[2]:")
    assert bench.workload_name(mk(depth=10, shadow_rays=0), dict(dim=1024), 1920, 1080, 1, True).startswith("BASELINE configs[1]:")
    w3 = bench.workload_name(mk(lights=2), sc12, 3840, 2160, 8, True)
    assert w3.startswith("BASELINE configs[3]:") and "row-tiled over 8 ranks (strong scaling)" in w3 and "2 lights" in w3
    sc16 = dict(dim=65536, device_built=True, thickness=33)
    w4 = bench.workload_name(mk(depth=16, lights=4, thickness=33), sc16, 7680, 4320, 8, True)
    assert w4.startswith("BASELINE configs[4]:") and "built on the device (thickness 33)" in w4
    assert "thin shell" in bench.workload_name(mk(depth=16, lights=4), dict(sc16, thickness=2), 7680, 4320, 8, True)
    assert "weak scaling" in bench.workload_name(mk(), sc12, 1920, 1080, 2, False)
    assert bench.metric_name(mk(), 1920, 1080) == "Mrays/s (primary+shadow), 1920x1080 into depth-12 SVO"
    assert "primary only" in bench.metric_name(mk(depth=10, shadow_rays=0), 1920, 1080)


@pytest.mark.gpu
def test_bench_line_says_what_it_measures():
    """N = 1 on a small frame: the line names the camera pose, carries the survey-camera leg measured in the same run, and the
    new roofline fields exist (null without a PMC measurement for this workload: never a stale number)."""
    rec = run_bench(["--steps", "3", "--warmup", "1", "--depth", "8", "--width", "320", "--height", "240", "--no-cpu-baseline"])
    cam = rec["config"]["camera"]
    assert len(cam["position"]) == 3 and cam["octree_bias"] == 1 and "bias" in cam["note"]
    sv = rec["survey_camera"]
    assert sv["value"] > 0 and sv["ms_per_step"] > 0 and sv["camera"]["octree_bias"] == 1
    assert sv["camera"]["position"][:2] == cam["position"][:2] and sv["camera"]["position"][2] <= cam["position"][2]
    assert "bit-identical to the oracle" in rec["config"]["stepping"] and "reference array branch" not in rec["config"]["stepping"]
    assert rec["roofline"]["traffic"] is None and rec["roofline"]["traffic_split"] is None and rec["roofline"]["valu_issue"] is None
    assert rec["scaling"] == "weak" and rec["n_gpus"] == 1 and "no N > 1 hardware measurement" in rec["config"]["multi_gpu"]


@pytest.mark.gpu
def test_bench_strong_scaling_and_lights_two_rank_rehearsal():
    """BASELINE configs[3]'s command line at a small size: a FIXED frame row-tiled over the ranks (no supersampling), two
    lights; two ranks share GPU 0 over gloo.  The rays of the two-rank frame are the rays of the one-rank frame."""
    args = ["--steps", "3", "--warmup", "1", "--depth", "8", "--width", "320", "--height", "240", "--lights", "2", "--no-cpu-baseline"]
    one = run_bench(args)
    two = run_bench(args + ["--scaling", "strong"], nproc=2, port=29553)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["value"] > 0
    assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"]            # same frame, only dealt to two ranks
    assert two["config"]["lights"] == 2 and "strong scaling" in two["config"]["workload"]
    assert "rehearsal" in two["config"]["multi_gpu"] and "survey_camera" not in two
    weak = run_bench(args, nproc=2, port=29555)
    assert weak["scaling"] == "weak" and weak["config"]["rays_per_step"] > 1.8 * one["config"]["rays_per_step"]


@pytest.mark.gpu
def test_bench_device_built_scene_two_rank_rehearsal():
    """BASELINE configs[4]'s command line with a thin shell and a small frame: depth >= 14 scenes exist only in HBM, every rank
    builds its own copy with the device builder; 4 lights, strong scaling."""
    rec = run_bench(["--steps", "2", "--warmup", "1", "--depth", "14", "--width", "320", "--height", "184", "--lights", "4",
                     "--scaling", "strong", "--no-cpu-baseline"], nproc=2, port=29557)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["descriptors"] > 300e6 and "built on the device" in rec["config"]["workload"]
    assert rec["config"]["rays_per_step"] > 320 * 184
