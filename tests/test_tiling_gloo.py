"""world_size-2 gloo test of the multi-GPU path's host logic (runs on CPU): every rank renders its
interleaved row bands (oracle standing in for the kernel), frames are exchanged only for checking,
ray counts are summed with all_reduce exactly like bench.py does."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scenes
    from oracle import orc
    from voxel_raycaster_amd import tiling
    import bench
    s = scenes.floor_pillars()
    buf, root = orc.octree_generate(s["grid"], s["dim"])
    atlas = scenes.hash_atlas()
    w, h, band = 64, 48, 8
    rows = tiling.rows_of_rank(h, rank, world, band)
    img = np.zeros((h, w, 4), dtype=np.float32)
    rays = 0
    for y in rows:
        i, _, c = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=s["lights"], atlas=atlas,
                              tile_dim=(16, 16), descriptors=buf, root_index=root, octree_dim=s["dim"], using_octree=0,
                              max_distance=96, rows=(int(y), int(y) + 1))
        img[y] = i[y]
        rays += c["primary_rays"] + c["shadow_rays"]
    total_rays, max_t = bench.reduce_over_ranks(rays, 0.5 + rank)          # SUM rays, MAX time
    t = torch.from_numpy(img)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    if rank == 0:
        full = tiling.merge_tiles([g.numpy() for g in gathered], h, world, band)
        ref, _, c = orc.raycast(width=w, height=h, cam_dir=s["cam_dir"], cam_pos=s["cam_pos"], lights=s["lights"], atlas=atlas,
                                tile_dim=(16, 16), descriptors=buf, root_index=root, octree_dim=s["dim"], using_octree=0,
                                max_distance=96)
        q.put((bool(np.array_equal(full, ref)), int(total_rays) == c["primary_rays"] + c["shadow_rays"], float(max_t)))
    dist.barrier()
    dist.destroy_process_group()


def test_row_tiling_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, rays_ok, max_t = q.get(timeout=180)
    for p in procs:
        p.join(60)
    assert same and rays_ok and max_t == 1.5


def test_rows_partition():
    from voxel_raycaster_amd import tiling
    for h, world, band in [(1080, 8, 8), (1080, 3, 16), (61, 2, 8), (8, 4, 8)]:
        seen = np.concatenate([tiling.rows_of_rank(h, r, world, band) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(h))
