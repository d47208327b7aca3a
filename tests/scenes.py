"""Seeded test scenes shared by the CPU and GPU parity tests.

Each scene is a dict of plain numpy inputs for both the oracle (oracle/orc.py)
and the product (voxel_raycaster_amd.CLCaster).  The application defaults come
from the reference's src/Application.cpp:35-79.
"""
from __future__ import annotations

import numpy as np

APP_LIGHT = np.array([[0.01, 0.01, 0.01, 0.2, 10.0, 10.0, 10.0, -1.0, -1.0, -1.5]], dtype=np.float32)


def hash_atlas(w=256, h=256):
    """256x256 RGBA8 atlas, deterministic, alpha 255 (numpy twin of vrc_scene_atlas is NOT required
    to match: fixtures store the atlas explicitly)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.uint64)
    v = (x * np.uint64(0x9E3779B1) ^ (y * np.uint64(0x85EBCA77))) * np.uint64(0xC2B2AE3D)
    v ^= v >> np.uint64(15)
    a = np.zeros((h, w, 4), dtype=np.uint8)
    a[..., 0] = (v & np.uint64(255)).astype(np.uint8)
    a[..., 1] = ((v >> np.uint64(8)) & np.uint64(255)).astype(np.uint8)
    a[..., 2] = ((v >> np.uint64(16)) & np.uint64(255)).astype(np.uint8)
    a[..., 3] = 255
    return a


def _grid(dim):
    return np.zeros((dim, dim, dim), dtype=np.int8)  # indexed [z][y][x] -> flat x + dim*(y + dim*z)


def app_default():
    """The reference application's own scene: 16^3 all material 5, camera inside solid."""
    g = np.full((16, 16, 16), 5, dtype=np.int8)
    return dict(name="app_default", dim=16, grid=g.reshape(-1), cam_pos=(2.34, 2.5, 7.17), cam_dir=(2.424, 3.141),
                lights=APP_LIGHT.copy())


def floor_pillars(dim=32, seed=3):
    rng = np.random.default_rng(seed)
    g = _grid(dim)
    g[0:3, :, :] = 5                                  # floor: z < 3
    for _ in range(dim // 2):
        x, y = rng.integers(2, dim - 2, size=2)
        hgt = int(rng.integers(4, dim // 2))
        g[3:3 + hgt, y, x] = 5
    lights = np.array([[0.01, 0.01, 0.01, 0.2, dim * 0.8, dim * 0.2, dim * 0.9, -1, -1, -1.5]], dtype=np.float32)
    return dict(name=f"floor_pillars{dim}", dim=dim, grid=g.reshape(-1), cam_pos=(dim * 0.5 + 0.37, 1.41, dim * 0.45 + 0.29),
                cam_dir=(2.0, 1.5708), lights=lights)


def mirror_wall(dim=32):
    g = _grid(dim)
    g[0:2, :, :] = 5
    g[:, dim - 4:dim - 2, :] = 6                      # mirror wall facing -y
    g[2:10, dim // 2, dim // 3] = 5
    lights = np.array([[0.01, 0.01, 0.01, 0.2, dim * 0.3, dim * 0.2, dim * 0.8, -1, -1, -1.5]], dtype=np.float32)
    return dict(name=f"mirror_wall{dim}", dim=dim, grid=g.reshape(-1), cam_pos=(dim * 0.5 + 0.21, 2.43, dim * 0.3 + 0.37),
                cam_dir=(1.8, 1.5708), lights=lights)


def open_sky(dim=32):
    """Mostly rays leaving the map (fog path) + one block."""
    g = _grid(dim)
    g[dim // 2:dim // 2 + 3, dim // 2:dim // 2 + 3, dim // 2:dim // 2 + 3] = 5
    lights = np.array([[0.01, 0.01, 0.01, 0.2, 3.0, 3.0, dim - 2.0, -1, -1, -1.5]], dtype=np.float32)
    return dict(name=f"open_sky{dim}", dim=dim, grid=g.reshape(-1), cam_pos=(dim * 0.5 + 0.6, 3.3, dim * 0.5 + 1.2),
                cam_dir=(1.5708, 1.5708), lights=lights)


def axis_aligned(dim=16):
    """cam_dir.y == 0 makes sin(yaw) == 0: the x == W/2 column has ray.y == 0 and is never written."""
    g = _grid(dim)
    g[0:2, :, :] = 5
    g[:, :, 0:2] = 5
    return dict(name=f"axis_aligned{dim}", dim=dim, grid=g.reshape(-1), cam_pos=(dim - 3.5, dim * 0.5 + 0.25, dim * 0.5 + 0.4),
                cam_dir=(1.9, 0.0), lights=APP_LIGHT.copy())


def random_sparse(dim=64, density=0.02, seed=11):
    rng = np.random.default_rng(seed)
    g = (rng.random((dim, dim, dim)) < density).astype(np.int8) * 5
    g[0:2, :, :] = 5
    lights = np.array([[0.01, 0.01, 0.01, 0.2, dim * 0.25, dim * 0.25, dim * 0.75, -1, -1, -1.5]], dtype=np.float32)
    return dict(name=f"random_sparse{dim}", dim=dim, grid=g.reshape(-1), cam_pos=(dim * 0.5 + 0.37, dim * 0.125 + 0.41, dim * 0.4 + 0.29),
                cam_dir=(2.0, 1.5708), lights=lights)


def with_lights(s, n):
    """The scene with n lights (multi-light extension): light 0 is the scene's own, the others sit
    inside the map above the floor, outside the map, close to the floor and in a far corner."""
    d = float(s["dim"])
    extra = np.array([[0.30, 0.10, 0.05, 0.3, d * 0.5 + 0.3, d * 0.6 + 0.2, d * 0.7 + 0.1, 0, 0, -1],
                      [0.05, 0.25, 0.10, 0.2, d * 1.5, d * 0.5 + 0.4, d * 1.2, 0, 0, -1],
                      [0.05, 0.05, 0.30, 0.1, d * 0.2 + 0.6, d * 0.8 + 0.1, 4.3, 0, 0, -1],
                      [0.20, 0.20, 0.20, 0.4, -3.5, -2.25, d * 0.9, 0, 0, -1],
                      [0.10, 0.00, 0.10, 0.2, d - 1.5, d - 1.25, d - 0.75, 0, 0, -1],
                      [0.00, 0.10, 0.10, 0.2, d * 0.35, d * 0.15, d * 0.5, 0, 0, -1],
                      [0.15, 0.05, 0.00, 0.2, d * 0.65, d * 0.35, d * 0.25, 0, 0, -1]], dtype=np.float32)
    out = dict(s)
    out["lights"] = np.concatenate([np.asarray(s["lights"], dtype=np.float32).reshape(-1, 10)[:1], extra])[:n].copy()
    return out


def shadow_box(dim=32):
    """A block on a floor, the light behind and above it, the camera close in front: the reference kernel's hard-coded
    20-step cap (ray_caster_kernel.cl:326) is enough to reach the floor, the block and the block's shadow."""
    g = _grid(dim)
    g[0:2, :, :] = 5
    g[2:10, 14:20, 14:20] = 5
    lights = np.array([[0.01, 0.01, 0.01, 0.2, 16.5, 26.5, 12.5, -1, -1, -1.5]], dtype=np.float32)
    return dict(name=f"shadow_box{dim}", dim=dim, grid=g.reshape(-1), cam_pos=(16.37, 5.41, 6.29), cam_dir=(2.2, 1.5708), lights=lights)


def near_mirror(dim=32):
    """mirror_wall from six voxels in front of the mirror: bounce, then floor / pillar hits within 20 steps."""
    s = mirror_wall(dim)
    s.update(name=f"near_mirror{dim}", cam_pos=(dim * 0.5 + 0.21, dim - 10.57, 4.37), cam_dir=(1.9, 1.5708))
    return s


def terrain256():
    """The 256^3 shell terrain (BASELINE configs[0] geometry: 67 689 descriptors, inside the reference's 100 000-entry
    buffer and its 8-level kernel stacks) seen from 3 voxels above the ground, looking 40 degrees down: slopes, x / z
    faces, lit and shadowed ground within the reference kernel's 20 steps."""
    import voxel_raycaster_amd as vrc
    dim, cx, cy = 256, 128, 40
    grid = vrc.shell_terrain_dense(8, seed=1, thickness=2)
    lo, hi = vrc.shell_column(8, cx, cy, seed=1, thickness=2)
    lights = np.array([[0.01, 0.01, 0.01, 0.2, 64.0, 64.0, 192.0, -1, -1, -1.5]], dtype=np.float32)
    return dict(name="terrain256", dim=dim, grid=grid, cam_pos=(cx + 0.37, cy + 0.41, hi + 3.3), cam_dir=(2.3, 1.5708), lights=lights)


ALL = [app_default, floor_pillars, mirror_wall, open_sky, axis_aligned, random_sparse]
# the scenes the reference's own kernel is run on (tests/test_reference_pin_gpu.py): its step cap is 20
REFERENCE_KERNEL_SCENES = ALL + [shadow_box, near_mirror, terrain256]
