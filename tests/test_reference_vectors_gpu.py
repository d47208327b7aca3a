"""The HIP path against REFERENCE-PRODUCED vectors, directly (-m gpu).

tests/golden/ref_*.npz are per-pixel outputs of the reference's own `raycaster` kernel (kernels/ray_caster_kernel.cl
#included unmodified, IEEE build, its two image builtins redirected -- tests/make_reference_golden.py).
tests/test_oracle_cpu.py replays the ORACLE against them; this file renders the same scenes through libvrc.so (array
branch, the reference's hard-coded 20-step cap :326) and compares the HIP frame with the vectors themselves, so the chain
HIP == oracle == reference is closed without the oracle in the middle: written mask, hit voxel / face / material / texel
count / bounce count / step count and colour of rays that hit nothing equal on every pixel; final step count and in-shadow
flag equal and RGB within 1e-5 relative on every shaded pixel (ray_caster_kernel.cl:140-251,555-721).

The vectors store the sin / cos of the camera angles as the code object evaluated them (the OpenCL library's); libvrc.so
evaluates sinf / cosf on the host (vrc_api.cpp) and has no way to be handed other values.  Frames whose four
values are equal are compared exactly; for the others the rays differ in the last bit, so only the decisions a 1-ulp change
of the direction cannot move are required (>= 99 % of the pixels hit the same voxel) -- the test prints which scenes
those are."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import refcompare
import scenes
import voxel_raycaster_amd as vrc
from test_parity_gpu import make_caster

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (the raycaster kernel's per-pixel records; ref_get_oct_vox_* / ref_view_light_* are the two functions' vectors)
REF_GOLDEN = sorted(p for p in glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.npz"))
                    if not os.path.basename(p).startswith(("ref_get_oct_vox_", "ref_view_light_")))


def host_trig(cam_dir):
    """sinf / cosf of the camera angles as vrc_api.cpp evaluates them (the C library's float functions)."""
    m = C.CDLL("libm.so.6")
    for f in (m.sinf, m.cosf):
        f.restype, f.argtypes = C.c_float, [C.c_float]
    return np.array([m.sinf(float(cam_dir[0])), m.cosf(float(cam_dir[0])), m.sinf(float(cam_dir[1])), m.cosf(float(cam_dir[1]))],
                    dtype=np.float32)


@pytest.mark.skipif(not REF_GOLDEN, reason="tests/golden/ref_*.npz missing (tests/make_reference_golden.py, GPU box)")
@pytest.mark.parametrize("path", REF_GOLDEN, ids=[os.path.basename(p)[4:-4] for p in REF_GOLDEN])
def test_hip_frame_equals_the_reference_kernel_vectors(path, atlas):
    z = np.load(path)
    s = getattr(scenes, str(z["scene"]))()
    w, h, dim = int(z["width"]), int(z["height"]), s["dim"]
    m = vrc.Map(dim, s["grid"], buffer_size=100000)                  # Octree::Generate's 100000-entry buffer (Octree.h:29)
    c = make_caster(m.octree, dim, 1, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, 20, grid=s["grid"])
    assert c.compute(), c.last_error()
    img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
    rec = z["records"]
    same_trig = np.array_equal(host_trig(s["cam_dir"]).view(np.uint32), z["trig"].view(np.uint32))
    if same_trig:
        refcompare.compare(s, w, h, rec, img, hits, {"unwritten": ctr["unwritten_pixels"], "n_tex": ctr["texel_reads"]}, verbose=False)
        return
    # the code object's sin / cos differ from the host's in the last bit: another ray table, statistically the same frame
    hit_ref, hit_hip = rec[..., 16] > 0, hits[..., 3] != 0
    both = hit_ref & hit_hip
    same_voxel = (rec[..., 17:20][both] == hits[..., 0:3][both]).all(-1)
    print(f"\n{os.path.basename(path)}: stored trig {z['trig']} != host {host_trig(s['cam_dir'])}: exact comparison skipped; "
          f"hit/miss agreement {float((hit_ref == hit_hip).mean()):.5f}, same voxel {float(same_voxel.mean()) if both.any() else 1.0:.5f}")
    assert (hit_ref == hit_hip).mean() >= 0.99 and (not both.any() or same_voxel.mean() >= 0.99)


def test_most_reference_vectors_are_compared_exactly():
    """The exact branch above must not silently become the rare one."""
    exact = 0
    for path in REF_GOLDEN:
        z = np.load(path)
        s = getattr(scenes, str(z["scene"]))()
        exact += bool(np.array_equal(host_trig(s["cam_dir"]).view(np.uint32), z["trig"].view(np.uint32)))
    assert REF_GOLDEN and exact * 2 >= len(REF_GOLDEN), f"only {exact} of {len(REF_GOLDEN)} reference vectors share the host's sin / cos"
