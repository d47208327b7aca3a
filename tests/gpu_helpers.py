"""Helpers the GPU test files share: the bench scene cache, SURVEY 8d's lights and camera, the settings / buffers of one caster,
peak-RSS bookkeeping, one frame as (image bits, hit records, counters).  Not a test file."""
import functools
import os
import resource
import time

import numpy as np

import voxel_raycaster_amd as vrc

@functools.lru_cache(maxsize=4)
def bench_scene(depth):
    import bench
    return bench.build_scene(depth)

def lights4(dim):
    """L0..L3 at the quarter points, 3/4 up, all rgbi (0.01, 0.01, 0.01, 0.2); fractional offsets keep
    shadow rays off exact voxel boundaries."""
    li = np.zeros((8, 10), dtype=np.float32)
    for l, (fx, fy) in enumerate(((0.25, 0.25), (0.75, 0.25), (0.25, 0.75), (0.75, 0.75))):
        li[l] = [0.01, 0.01, 0.01, 0.2, fx * dim + 0.3 * l, fy * dim + 0.2 * l, 0.75 * dim + 0.1 * l, -1.0, -1.0, -1.5]
    return li

def configure(c, dim, atlas, cam_dir, cam_pos, lights, w, h, light_count=1, shadow_rays=1, table=None):
    assert c.add_to_settings_buffer("octree_dimensions", "OCTDIM", dim)
    assert c.add_to_settings_buffer("using_octree", "OCTENABLED", 0)
    assert c.add_to_settings_buffer("max_distance", "MAX_DISTANCE", 3 * dim)
    assert c.add_to_settings_buffer("shadow_rays", "SHADOW_RAYS", shadow_rays)
    assert c.add_to_settings_buffer("light_count", "LIGHT_COUNT", light_count)
    cd, cp = np.array(cam_dir, dtype=np.float32), np.array(cam_pos, dtype=np.float32)
    assert c.assign_camera(cd, cp)
    assert c.create_viewport(w, h) if table is None else c.create_viewport_table(table)
    assert c.assign_lights(lights)
    assert c.create_texture_atlas(atlas, (16, 16))
    c._li = lights
    return c

def _rss_now_kb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS:"):
            return int(line.split()[1])
    return 0

def _peak_rss_kb():
    for line in open("/proc/self/status"):
        if line.startswith("VmHWM:"):
            return int(line.split()[1])
    return 0

def _reset_peak_rss():
    """Resets the process's peak-RSS counter (VmHWM) so that a later reading belongs to what ran in between."""
    try:
        with open("/proc/self/clear_refs", "w") as f:
            f.write("5")
        return abs(_peak_rss_kb() - _rss_now_kb()) < 64 * 1024
    except OSError:
        return False

def survey_camera(depth, seed=1, thickness=2, octave_floor=2):
    """SURVEY 8d's camera exactly as written: (D/2 + 0.37, D/8 + 0.41, h(D/2, D/8) + D/16 + 0.29), looking
    (inclination 2.0, azimuth 1.5708); no search for a voxel whose octree bias is zero."""
    dim = 1 << depth
    _, hi = vrc.shell_column(depth, dim // 2, dim // 8, seed=seed, thickness=thickness, octave_floor=octave_floor)
    return (2.0, 1.5708), (dim / 2 + 0.37, dim / 8 + 0.41, hi + dim // 16 + 0.29)

def _frame(c):
    assert c.compute(), c.last_error()
    return c.read_image().view(np.uint32).copy(), c.read_hits().copy(), c.counters()

def _but_reads(ctr):
    return {k: v for k, v in ctr.items() if k not in ("descriptor_reads", "canonical_reads")}
