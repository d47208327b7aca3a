"""One seeded random scene through every kernel against the oracle (used by tests/test_parity_gpu.py::test_fuzz_random_scenes
with 24 fixed seeds and by tests/soak_fuzz_gpu.py for as long as one likes)."""
import numpy as np

import voxel_raycaster_amd as vrc
from oracle import orc
from test_parity_gpu import assert_same, make_caster


def run_case(seed, atlas, dims=(4, 8, 16, 32)):
    """dims 4..32 (or as given), materials {0,1,5,6}, cameras inside / outside / on the grid, random lights, ragged
    resolutions, random step caps; the array kernel, the SVO kernel (+attachments), the SVO kernel with exact closed-form
    jumps and mode B (node-exit jumps, against its own restatement) -- each equal to the oracle bit for bit."""
    rng = np.random.default_rng(seed)
    dim = int(rng.choice(dims))
    dens = float(rng.choice([0.02, 0.1, 0.4]))
    g = rng.choice(np.array([0, 1, 5, 6], dtype=np.int8), size=dim ** 3, p=[1 - dens, dens * 0.1, dens * 0.7, dens * 0.2])
    if rng.random() < 0.5:
        g.reshape(dim, dim, dim)[0:max(1, dim // 8)] = 5
    cam_pos = tuple(float(v) for v in (rng.random(3) * (dim + 4) - 2))
    if rng.random() < 0.2:
        cam_pos = tuple(float(int(v)) for v in cam_pos)               # on integer coordinates
    cam_dir = (float(rng.random() * 3.0 + 0.1), float(rng.random() * 6.2))
    nl = int(rng.choice([1, 1, 2, 3, 8]))                                # multi-light extension on half the seeds
    lights = np.array([[0.01, 0.01, 0.01, 0.2, *(rng.random(3) * dim * 1.4 - 0.2 * dim), -1, -1, -1.5] for _ in range(nl)],
                      dtype=np.float32)
    if nl > 1 and rng.random() < 0.3:
        lights[1, 4:7] = np.floor(lights[1, 4:7]) + 0.5                   # voxel centres: zero ray components (:671)
    w, h = int(rng.integers(9, 90)), int(rng.integers(9, 60))
    md = int(rng.choice([0, 1, 7, 20, 3 * dim]))
    o = vrc.Octree.Generate(g, dim, buffer_size=100000 if dim <= 32 else 0).attach_materials_from_grid(g)
    for using_octree, variant in ((1, "array"), (0, "svo"), (0, "jumps"), (0, "mode B")):
        c = make_caster(o, dim, using_octree, cam_dir, cam_pos, lights, atlas, w, h, md, grid=g, light_count=nl)
        if variant == "jumps":
            assert c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 2)
        mode = 1 if variant == "mode B" else 0
        if mode:
            assert c.add_to_settings_buffer("stepping_mode", "STEPPING_MODE", 1)
        assert c.compute(), c.last_error()
        kw = dict(attachment_lookup=o.attachment_lookup, attachments=o.attachment_buffer) if not using_octree else {}
        oimg, ohits, octr = orc.raycast(width=w, height=h, cam_dir=cam_dir, cam_pos=cam_pos, lights=c._li, atlas=atlas,
                                        tile_dim=(16, 16), descriptors=o.descriptor_buffer, root_index=o.root_index,
                                        octree_dim=dim, using_octree=using_octree, grid=g, max_distance=md,
                                        active_lights=nl, stepping_mode=mode, **kw)
        try:
            assert_same(c.read_image(), c.read_hits(), c.counters(), oimg, ohits, octr)
        except AssertionError as e:
            raise AssertionError(f"seed {seed} dim {dim} {variant}: {e}") from None
