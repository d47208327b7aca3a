"""GPU suite (-m gpu): stepping mode 1 -- the node-exit jump kernel (SURVEY 7 D1 "mode B", csrc/raycast_jump_kernel.hip)
-- bit-exact against its own restatement in oracle/vrc_oracle.c (jump_step & co), and how far it is from the exact
mode.  Mode B is a labelled, opt-in mode; it is never the headline and never claims reference parity."""
import functools

import numpy as np
import pytest

import scenes
import voxel_raycaster_amd as vrc
from oracle import orc
from test_parity_gpu import assert_same, hits_match, make_caster

pytestmark = pytest.mark.gpu


@functools.lru_cache(maxsize=2)
def bench_scene(depth):
    import bench
    return bench.build_scene(depth)


@pytest.mark.parametrize("make", scenes.ALL, ids=[f.__name__ for f in scenes.ALL])
@pytest.mark.parametrize("res", [(160, 120), (97, 61)], ids=["160x120", "ragged97x61"])
def test_jump_kernel_equals_its_oracle(make, res, atlas):
    s = make()
    dim, (w, h) = s["dim"], res
    m = vrc.Map(dim, s["grid"], buffer_size=100000)
    md = 20 if dim <= 16 else 3 * dim
    c = make_caster(m.octree, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, md)
    assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 1) and c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=m.octree.descriptor_buffer, root_index=m.octree.root_index,
                                    octree_dim=dim, using_octree=0, max_distance=md, stepping_mode=1)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    # and back: the setting is live, the exact kernel is untouched
    assert c.overwrite_setting("stepping_mode", 0) and c.compute()
    eimg, ehits, ectr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=m.octree.descriptor_buffer, root_index=m.octree.root_index,
                                    octree_dim=dim, using_octree=0, max_distance=md)
    assert_same(c.read_image(), c.read_hits(), c.counters(), eimg, ehits, ectr)


@pytest.mark.parametrize("make", [scenes.floor_pillars, scenes.random_sparse, scenes.mirror_wall])
@pytest.mark.parametrize("n", [2, 4])
def test_jump_kernel_multi_light_and_materials(make, n, atlas):
    from test_oracle_cpu import _with_pass_through
    s = scenes.with_lights(_with_pass_through(make()), n)
    dim, w, h = s["dim"], 128, 96
    o = vrc.Octree.Generate(s["grid"], dim).attach_materials_from_grid(s["grid"])
    c = make_caster(o, dim, 0, s["cam_dir"], s["cam_pos"], s["lights"], atlas, w, h, 3 * dim, light_count=n)
    assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 1) and c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index, octree_dim=dim,
                                    using_octree=0, max_distance=3 * dim, stepping_mode=1, active_lights=n,
                                    attachment_lookup=o.attachment_lookup, attachments=o.attachment_buffer)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)


def test_jump_kernel_headline_frame_sampled_rows_and_distance_to_exact_mode():
    """BASELINE configs[2] frame in mode B: sampled rows bit-exact vs the mode-B oracle; against the exact mode the same
    voxel / face / material is hit on nearly every pixel (the rest are grazing-edge ties), while RGB within 1e-5 holds
    only where the hit block's UV does not cross a texel boundary (the exact mode's accumulated rounding is part of its
    result, ray_caster_kernel.cl:592-614) -- which is why this mode is not a parity mode."""
    sc = bench_scene(12)
    dim, w, h = sc["dim"], 1920, 1080
    c = make_caster(sc["octree"], dim, 0, sc["cam_dir"], sc["cam_pos"], sc["lights"], sc["atlas"], w, h, 3 * dim)
    assert c.compute()
    eimg, ehits, ectr = c.read_image(), c.read_hits(), c.counters()
    assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 1) and c.compute(), c.last_error()
    img, hits, ctr = c.read_image(), c.read_hits(), c.counters()
    for y0 in range(23, h, 149):
        oimg, ohits, _ = orc.raycast(width=w, height=h, cam_dir=sc["cam_dir"], cam_pos=sc["cam_pos"], lights=c._li, atlas=sc["atlas"],
                                     tile_dim=(16, 16), descriptors=sc["octree"].descriptor_buffer, root_index=sc["octree"].root_index,
                                     octree_dim=dim, using_octree=0, max_distance=3 * dim, rows=(y0, y0 + 1), threads=16, stepping_mode=1)
        assert hits_match(c, hits[y0], ohits[y0])
        assert np.array_equal(img[y0].view(np.uint32), oimg[y0].view(np.uint32))
    same = (hits[..., :5] == ehits[..., :5]).all(-1)
    rel = np.abs(img[..., :3] - eimg[..., :3]) / np.maximum(np.abs(eimg[..., :3]), 1e-6)
    print(f"\nmode B vs exact, headline frame: same hit voxel/face/material {same.mean():.5f}, rgb within 1e-5 {(rel.max(-1) <= 1e-5).mean():.4f}, "
          f"voxel steps {ctr['steps'] / 1e9:.3f} G vs {ectr['steps'] / 1e9:.3f} G, descriptor reads {ctr['descriptor_reads'] / 1e6:.1f} M vs "
          f"{ectr['descriptor_reads'] / 1e6:.1f} M")
    assert same.mean() >= 0.98
    assert abs(ctr["steps"] - ectr["steps"]) <= 0.05 * ectr["steps"]


def test_jump_kernel_degenerate_directions_make_progress(atlas):
    """Rays with a denormal direction component that start exactly on a voxel plane: 1 / component is inf and the
    distance to that plane 0, so the node exit used to have no minimum (0 * inf = NaN) and the jump made no progress
    until the round watchdog cut the frame (advisor finding, round 2).  The reciprocal is clamped in the kernel and in
    its restatement; the frame completes and the two agree."""
    s = scenes.floor_pillars()
    dim, w, h, md = s["dim"], 64, 48, 3 * s["dim"]
    o = vrc.Octree.Generate(s["grid"], dim, buffer_size=100000)
    table = orc.create_viewport(w, h).reshape(h, w, 4).copy()
    tiny = np.float32(1e-40)
    table[::2, :, 0] = -tiny                        # down x, denormal: the plane behind the camera is at distance 0
    table[1::2, :, 1] = tiny
    table[:, ::3, 2] = -tiny
    cam_pos = (float(dim // 2), float(dim // 2), float(dim // 2 + 3))    # integer coordinates: on the planes of its voxel
    c = make_caster(o, dim, 0, (0.0, 0.0), cam_pos, s["lights"], atlas, w, h, md)    # direction (0, 0): the table rays pass through
    assert c.create_viewport_table(table) and c.validate()
    assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 1) and c.compute(), c.last_error()
    oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=(0.0, 0.0), cam_pos=cam_pos, lights=c._li, atlas=atlas,
                                    tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                    octree_dim=dim, using_octree=0, max_distance=md, viewport=table, stepping_mode=1)
    assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
    # (the exact mode is not compared on these rays: an integer camera coordinate times an infinite delta_t makes the
    # reference's intersection_t NaN (ray_caster_kernel.cl:317), and what min() does with a NaN is undefined in OpenCL C;
    # it must only terminate)
    assert c.overwrite_setting("stepping_mode", 0) and c.compute(), c.last_error()
