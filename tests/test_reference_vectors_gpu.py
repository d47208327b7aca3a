"""The HIP path against REFERENCE-PRODUCED vectors, directly (-m gpu).

tests/golden/ref_*.npz are per-pixel outputs of the reference's own `raycaster` kernel (kernels/ray_caster_kernel.cl
#included unmodified, IEEE build, its two image builtins redirected -- tests/make_reference_golden.py).
tests/test_oracle_cpu.py replays the ORACLE against them; this file renders the same scenes through libvrc.so (array
branch, the reference's hard-coded 20-step cap :326) and compares the HIP frame with the vectors themselves, so the chain
HIP == oracle == reference is closed without the oracle in the middle: written mask, hit voxel / face / material / texel
count / bounce count / step count and colour of rays that hit nothing equal on every pixel; final step count and in-shadow
flag equal and RGB within 1e-5 relative on every shaded pixel (ray_caster_kernel.cl:140-251,555-721).

The vectors store the sin / cos of the camera angles as the code object evaluated them (the OpenCL library's,
ray_caster_kernel.cl:280-291); for 9 of the 19 frames they differ from the host's sinf / cosf in the last bit, i.e. the rays
differ.  vrc_assign_camera_trig hands libvrc.so exactly those four values (a host that evaluates them like its reference build
does the same), so EVERY vector is compared exactly."""
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
    trig = np.ascontiguousarray(z["trig"], dtype=np.float32)         # the code object's own sin / cos
    assert c.assign_camera_trig(trig), c.last_error()
    assert c.compute(), c.last_error()
    img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
    rec = z["records"]
    refcompare.compare(s, w, h, rec, img, hits, {"unwritten": ctr["unwritten_pixels"], "n_tex": ctr["texel_reads"]}, verbose=False)


def test_the_stored_trig_matters_for_some_vectors():
    """Without vrc_assign_camera_trig the library's sinf / cosf are what it renders with; for some of the vectors those are not
    the code object's values -- the reason the call exists.  (If this ever fails the call has become redundant, not wrong.)"""
    differing = 0
    for path in REF_GOLDEN:
        z = np.load(path)
        s = getattr(scenes, str(z["scene"]))()
        differing += not np.array_equal(host_trig(s["cam_dir"]).view(np.uint32), z["trig"].view(np.uint32))
    assert REF_GOLDEN and differing > 0


def test_camera_trig_is_a_live_pointer_and_null_restores_the_default(atlas):
    """vrc_assign_camera_trig retains the pointer like vrc_assign_camera (CLCaster.cpp:137-139): new values are seen by the
    next compute(); NULL brings the library's own sinf / cosf back."""
    s = scenes.floor_pillars()
    dim, w, h = s["dim"], 64, 48
    m = vrc.Map(dim, s["grid"])
    c = make_caster(m.octree, dim, 1, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, 3 * dim, grid=s["grid"])
    assert c.compute(), c.last_error()
    base = c.read_image().copy()
    trig = host_trig(s["cam_dir"]).copy()
    assert c.assign_camera_trig(trig) and c.compute()
    assert np.array_equal(c.read_image().view(np.uint32), base.view(np.uint32))           # the same four values: the same frame
    trig[:] = host_trig(np.array([s["cam_dir"][0] + 0.2, s["cam_dir"][1] - 0.1], np.float32))   # the host turns the camera its own way
    assert c.compute()
    turned = c.read_image().copy()
    assert not np.array_equal(turned.view(np.uint32), base.view(np.uint32))
    from oracle import orc                 # checker only
    oimg, _, _ = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas, tile_dim=(16, 16),
                             descriptors=m.octree.descriptor_buffer, root_index=m.octree.root_index, octree_dim=dim, using_octree=1,
                             grid=s["grid"], max_distance=3 * dim, trig=trig)
    assert np.array_equal(turned.view(np.uint32), oimg.view(np.uint32))
    assert c.assign_camera_trig(None) and c.compute()
    assert np.array_equal(c.read_image().view(np.uint32), base.view(np.uint32))
