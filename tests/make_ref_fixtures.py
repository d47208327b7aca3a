#!/usr/bin/env python3
"""Generates tests/golden/ref_*.npz: golden vectors from the REFERENCE KERNEL ITSELF.

Runs on a GPU box (no /root/reference needed there): feeds seeded scenes to
oracle/_ref/ref_run, which executes the reference's unmodified
kernels/ray_caster_kernel.cl (compiled for gfx950 here by oracle/ref_build.sh)
through the AMD OpenCL runtime, and stores inputs + the kernel's float4 frames.

  gpurun -- python tests/make_ref_fixtures.py          # writes gpurun_out/ref_fixtures/
  cp gpurun_out/ref_fixtures/*.npz tests/golden/       # commit

Fixture = data only: numpy inputs (grid, descriptors, ray table, camera trig,
lights, atlas) and the reference's outputs for two builds of the same source:
"strict" (IEEE, -ffp-contract=off) and "shipped" (the reference's own
-cl-fast-relaxed-math options, src/CLCaster.cpp:767-771).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import orc  # noqa: E402  (test infrastructure)
import scenes  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def scene_list():
    out = []
    s = scenes.app_default(); s["res"] = (64, 48); out.append(s)
    s = scenes.app_default(); s["name"] = "app_default_160"; s["res"] = (160, 120); out.append(s)
    s = scenes.floor_pillars(16, seed=5); s["cam_pos"] = (8.37, 1.41, 6.29); s["res"] = (64, 48); out.append(s)
    s = scenes.floor_pillars(32, seed=3); s["cam_pos"] = (16.37, 9.41, 7.29); s["res"] = (160, 120); out.append(s)
    s = scenes.mirror_wall(16); s["cam_pos"] = (8.21, 5.43, 4.37); s["res"] = (64, 48); out.append(s)
    s = scenes.open_sky(16); s["cam_pos"] = (8.6, 3.3, 9.2); s["res"] = (64, 48); out.append(s)
    s = scenes.axis_aligned(16); s["res"] = (64, 48); out.append(s)
    s = scenes.random_sparse(32, density=0.05, seed=11); s["res"] = (64, 48); out.append(s)
    return out


def run_ref(co, scene_dir, out_file):
    r = subprocess.run([os.path.join(REF, "ref_run"), os.path.relpath(co, os.getcwd()), scene_dir, out_file],
                       capture_output=True, text=True)
    sys.stderr.write(r.stderr)
    if r.returncode != 0:
        raise RuntimeError(f"ref_run failed ({r.returncode}) for {co}")


def pick_binaries():
    """First code-object version the runtime accepts."""
    probe = scene_list()[0]
    for cov in (6, 5):
        strict = os.path.join(REF, f"raycaster_strict_cov{cov}.co")
        try:
            with tempfile.TemporaryDirectory() as td:
                write_scene(probe, td, using_octree=1)
                run_ref(strict, td, os.path.join(td, "o.bin"))
            return cov
        except RuntimeError as e:
            print("code object v%d rejected: %s" % (cov, e), file=sys.stderr)
    raise SystemExit("no reference code object could be run")


def write_scene(s, td, using_octree):
    w, h = s["res"]
    dim = s["dim"]
    buf, root = orc.octree_generate(s["grid"], dim)            # reference-sized 100000-entry buffer
    vp = orc.create_viewport(w, h)
    atlas = scenes.hash_atlas()
    lights = np.zeros((8, 10), dtype=np.float32)               # 8 reserved slots (LightController.h:95)
    lights[: s["lights"].shape[0]] = s["lights"]
    s["_buf"], s["_root"], s["_vp"], s["_atlas"], s["_lights"] = buf, root, vp, atlas, lights
    np.asarray(s["grid"], dtype=np.int8).tofile(os.path.join(td, "map.bin"))
    vp.tofile(os.path.join(td, "viewport.bin"))
    np.array(list(s["cam_dir"]) + list(s["cam_pos"]), dtype=np.float32).tofile(os.path.join(td, "camera.bin"))
    lights.tofile(os.path.join(td, "lights.bin"))
    atlas.tofile(os.path.join(td, "atlas.bin"))
    buf.tofile(os.path.join(td, "desc.bin"))
    with open(os.path.join(td, "params.txt"), "w") as f:
        f.write(f"{w} {h} {dim} {dim} {dim} {dim} {root} {using_octree} 256 256 16 16 {buf.size}\n")


def main():
    out_dir = os.path.join(ROOT, "gpurun_out", "ref_fixtures")
    os.makedirs(out_dir, exist_ok=True)
    os.chdir(ROOT)
    cov = pick_binaries()
    print("using code object version", cov)
    for s in scene_list():
        w, h = s["res"]
        res = {}
        for variant in ("strict", "shipped"):
            for using_octree in (1, 0):     # 1 = array branch (the working renderer), 0 = unfinished octree stepper
                with tempfile.TemporaryDirectory() as td:
                    write_scene(s, td, using_octree)
                    out = os.path.join(td, "image.bin")
                    run_ref(os.path.join(REF, f"raycaster_{variant}_cov{cov}.co"), td, out)
                    res[(variant, using_octree)] = np.fromfile(out, dtype=np.float32).reshape(h, w, 4)
        trig = orc.camera_trig(np.array(s["cam_dir"], dtype=np.float32))
        used = s["_buf"][s["_root"]:]
        np.savez_compressed(
            os.path.join(out_dir, f"ref_{s['name']}.npz"),
            name=s["name"], dim=s["dim"], width=w, height=h, grid=np.asarray(s["grid"], dtype=np.int8),
            descriptors_tail=used, root_index=s["_root"], buffer_size=s["_buf"].size,
            viewport=s["_vp"], cam_dir=np.array(s["cam_dir"], dtype=np.float32),
            cam_pos=np.array(s["cam_pos"], dtype=np.float32), cam_trig=trig, lights=s["_lights"],
            tile_dim=np.array([16, 16], dtype=np.int32), max_distance=20,
            ref_strict_array=res[("strict", 1)], ref_shipped_array=res[("shipped", 1)],
            ref_strict_octree=res[("strict", 0)], ref_shipped_octree=res[("shipped", 0)])
        # quick report against the CPU oracle
        img, _, _ = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=s["_lights"],
                                atlas=s["_atlas"], tile_dim=(16, 16), descriptors=s["_buf"], root_index=s["_root"],
                                octree_dim=s["dim"], using_octree=1, grid=s["grid"], max_distance=20, trig=trig)
        for variant in ("strict", "shipped"):
            ref = res[(variant, 1)]
            err = np.abs(img - ref) / np.maximum(np.abs(ref), 1e-3)
            bad = (err > 1e-5).any(axis=-1)
            print(f"{s['name']:>20s} {variant:8s} max rel err {err.max():.3e}  pixels > 1e-5: {int(bad.sum())}/{w*h}")
    np.save(os.path.join(out_dir, "atlas.npy"), scenes.hash_atlas())
    print("fixtures in", out_dir)


if __name__ == "__main__":
    main()
