"""Test helper: compares the records of the REFERENCE'S OWN `raycaster` kernel (oracle/ref_raycaster_probe.cl, run on
the MI355X by oracle/ref_probe_run.cpp) with the oracle's frame of the same scene.  Used live by
tests/test_reference_pin_gpu.py and on the committed vectors tests/golden/ref_*.npz by tests/test_oracle_cpu.py.
Record layout (32 ints per pixel): see oracle/ref_raycaster_probe.cl."""
import numpy as np

from oracle import orc


def oracle_frame(s, w, h, atlas, buf, root, trig, threads=8):
    """The oracle on exactly what the reference kernel was given: array branch, its hard-coded 20-step cap
    (ray_caster_kernel.cl:326), the sin/cos the code object itself evaluated."""
    return orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=s["lights"], atlas=atlas,
                       tile_dim=(16, 16), descriptors=buf, root_index=root, octree_dim=s["dim"], using_octree=1,
                       grid=s["grid"], max_distance=20, trig=trig, threads=threads)


def compare(s, w, h, rec, oimg, ohits, octr, verbose=True, totals=None, strict=True):
    """Everything that depends only on the primary ray must be EQUAL.  What follows the shadow redirect goes through the
    OpenCL library's approximate normalize / fast_distance: on the FIXED scenes (strict, the default) the measured result
    is asserted -- final step count and in-shadow flag equal on every shaded pixel, RGB within BASELINE's 1e-5 relative on
    every shaded pixel; a random-pose soak (strict=False, or totals) keeps the loose floors, because a pose can put a
    pixel where the library's 1-2 ulp decide a step count.
    (oimg, ohits, octr) may come from the oracle or from libvrc.so: same layouts.)"""
    written = rec[..., 15] == 1
    # pixels the kernel returned from without writing (:293-294, :671-672, :694-695)
    assert np.array_equal(written, (ohits[..., 5] & 1) == 1), "written / unwritten pixels differ"
    assert octr["unwritten"] == int((~written).sum())
    # the first solid hit of the primary ray: recorded at the kernel's first read_imagef (:652 / :684)
    hit = rec[..., 16] > 0
    assert np.array_equal(hit, ohits[..., 3] != 0), "which pixels hit something differs"
    assert np.array_equal(rec[..., 17:20][hit], ohits[..., 0:3][hit]), "hit voxel"
    face = rec[..., 20] | (rec[..., 21] << 1) | (rec[..., 22] << 2)
    assert np.array_equal(face[hit], ohits[..., 4][hit]), "hit face"
    assert np.array_equal(rec[..., 23][hit], ohits[..., 3][hit]), "hit material"
    assert octr["n_tex"] == int(rec[..., 16].sum()), "texel fetches"
    # rays that never hit anything: the whole ray is primary, so the end state is exact too
    miss = written & ~hit
    assert np.array_equal(rec[..., 11][miss], ohits[..., 6][miss]), "step count of rays that hit nothing"
    fcol = rec[..., 0:4].view(np.float32)
    assert np.array_equal(fcol[miss].view(np.uint32), oimg[miss].view(np.uint32)), "colour of rays that hit nothing"
    # mirror bounces (:682-704)
    assert np.array_equal(rec[..., 13][written], (ohits[..., 5][written] >> 4) & 3), "bounce count"
    # after the shadow redirect: library normalize / fast_distance -> statistics, near-total agreement required
    w_ = written & hit
    if w_.any():
        same_steps = rec[..., 11][w_] == ohits[..., 6][w_]
        rel = np.abs(fcol[w_][:, :3] - oimg[w_][:, :3]) / np.maximum(np.abs(oimg[w_][:, :3]), 1e-6)
        shadow_same = (rec[..., 12][w_] != 0) == ((ohits[..., 5][w_] & 2) != 0)
        alpha_same = (np.abs(fcol[w_][:, 3] - oimg[w_][:, 3]) <= 1e-5 * np.maximum(np.abs(oimg[w_][:, 3]), 1e-6))
        if verbose:
            print(f"\n{s['name']} {w}x{h}: {int(w_.sum())} shaded pixels; final step count equal {same_steps.mean():.5f}, "
              f"rgb within 1e-5 {float((rel.max(-1) <= 1e-5).mean()):.5f} (worst {float(rel.max()):.2e}), "
              f"alpha (in-shadow flag) equal {alpha_same.mean():.5f}")
        if totals is not None:                          # a soak adds up many small frames instead of judging each one
            for k, v in (("shaded", int(w_.sum())), ("same_steps", int(same_steps.sum())), ("alpha_same", int(alpha_same.sum())),
                         ("shadow_same", int(shadow_same.sum())), ("rgb_1e-5", int((rel.max(-1) <= 1e-5).sum())),
                         ("rgb_1e-4", int((rel.max(-1) <= 1e-4).sum()))):
                totals[k] = totals.get(k, 0) + v
            totals["worst_rgb"] = max(totals.get("worst_rgb", 0.0), float(rel.max()))
            return
        assert shadow_same.all()
        if strict:
            assert same_steps.all(), f"final step count differs on {int((~same_steps).sum())} shaded pixels"
            assert alpha_same.all(), f"alpha differs on {int((~alpha_same).sum())} shaded pixels"
            assert (rel.max(-1) <= 1e-5).all(), f"rgb beyond 1e-5 on {int((rel.max(-1) > 1e-5).sum())} shaded pixels (worst {float(rel.max()):.3g})"
        else:
            assert same_steps.mean() >= 0.995 and alpha_same.mean() >= 0.995
            assert (rel.max(-1) <= 1e-4).mean() >= 0.995
