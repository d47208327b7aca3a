#!/usr/bin/env python3
"""Table for DESIGN.md section 7: every BASELINE config (and the deeper scenes) on ONE MI355X, kernel time from HIP
events, in both stepping modes.  Depth <= 13: the host-built bench scene and camera of bench.py; deeper: the tree is
built on the device (vrc_build_shell_terrain), the camera is SURVEY 8d's as written and the reference's octree bias
(ray_caster_kernel.cl:353-354) is switched off (setting octree_bias = 0) so that both modes cast the same rays -- the
bench camera of the host-built scenes sits where the bias is zero anyway.  python tools/bench_configs.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import voxel_raycaster_amd as vrc


def measure(c, name, extra):
    out = []
    assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 0)
    for mode in (0, 1):
        assert c.overwrite_setting("stepping_mode", mode)
        for _ in range(2):
            assert c.compute(), c.last_error()
        c.timing_reset()
        reps = 10 if mode else 5
        for _ in range(reps):
            assert c.compute(), c.last_error()
        n, ms = c.timing()
        ctr = c.counters()
        rays = ctr["primary_rays"] + ctr["shadow_rays"]
        m = c.memory_usage2()
        row = dict(config=name, mode="exact" if mode == 0 else "B (node-exit jumps)", kernel_ms=round(ms / n, 3),
                   Mrays_s=round(rays / (ms / n) / 1e3, 1), rays=rays, steps=ctr["steps"], descriptor_reads=ctr["descriptor_reads"],
                   empty_boxes=m["empty_boxes"], box_build_seconds=round(m["box_build_seconds"], 3), box_MB=round(m["box_bytes"] / 1e6), **extra)
        print(json.dumps(row), flush=True)
        out.append(row)
    return out


def host_scene(name, depth, w, h, shadow, lights=1):
    sc = bench.build_scene(depth)
    c = bench.make_caster(sc, w, h, 0, shadow_rays=shadow, light_count=lights)
    return measure(c, name, dict(depth=depth, frame=f"{w}x{h}", lights=lights, descriptors=int(sc["octree"].descriptor_buffer.size)))


def device_scene(name, depth, w, h, lights, thickness=2):
    dim = 1 << depth
    c = vrc.CLCaster()
    assert c.init(0)
    t0 = time.perf_counter()
    info, _ = c.build_shell_terrain(depth, 1, thickness, 2)
    build_s = time.perf_counter() - t0
    _, hi = vrc.shell_column(depth, dim // 2, dim // 8, thickness=thickness)
    cam_dir = np.array([2.0, 1.5708], dtype=np.float32)
    cam_pos = np.array([dim / 2 + 0.37, dim / 8 + 0.41, hi + dim // 16 + 0.29], dtype=np.float32)
    sc = bench.build_scene(8)                                      # lights / atlas conventions only
    li = sc["lights"].copy()
    li[:, 4:7] *= dim / 256.0
    ok = (c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim) and c.add_to_settings_buffer("using_octree", "OCTENABLED", 0)
          and c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", 3 * dim) and c.add_to_settings_buffer("light_count", "LIGHT_COUNT", lights)
          and c.add_to_settings_buffer("octree_bias", "OCTREE_BIAS", 0)
          and c.assign_camera(cam_dir, cam_pos) and c.create_viewport(w, h) and c.assign_lights(li)
          and c.create_texture_atlas(sc["atlas"], (16, 16)) and c.validate())
    assert ok, c.last_error()
    return measure(c, name, dict(depth=depth, frame=f"{w}x{h}", lights=lights, descriptors=int(info["n_descriptors"]),
                                 build_seconds=round(build_s, 2), thickness=thickness))


if __name__ == "__main__":
    host_scene("C1 geometry (d8, 640x480, primary only) on the GPU", 8, 640, 480, 0)
    host_scene("C2 (d10, 1080p, primary only)", 10, 1920, 1080, 0)
    host_scene("C3 headline (d12, 1080p, primary+shadow)", 12, 1920, 1080, 1)
    host_scene("C4 geometry (d12, 4K, 1 light) on 1 GPU", 12, 3840, 2160, 1)
    host_scene("C4 geometry (d12, 4K, 2 lights) on 1 GPU", 12, 3840, 2160, 1, lights=2)
    host_scene("C5 light count on the headline scene (d12, 1080p, 4 lights)", 12, 1920, 1080, 1, lights=4)
    host_scene("beyond BASELINE: d13 (8192^3), 1080p", 13, 1920, 1080, 1)
    device_scene("d14 (16384^3) built on the device, 1080p", 14, 1920, 1080, 1)
    device_scene("d15 (32768^3) built on the device, 1080p", 15, 1920, 1080, 1)
    device_scene("d16 (65536^3, thickness 2) built on the device, 1080p", 16, 1920, 1080, 1)
    device_scene("C5 scene (d16, thickness 33: ~198 GB resident) 1080p 1 light", 16, 1920, 1080, 1, thickness=33)
    device_scene("C5 (d16 ~198 GB resident, 7680x4320, 4 lights) on ONE GPU", 16, 7680, 4320, 4, thickness=33)
