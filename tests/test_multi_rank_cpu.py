"""The N>1 data path on CPU: two gloo ranks each render their interleaved tiles (with the CPU oracle standing in for
the GPU kernels), exchange packed tile buffers with all_gather_into_tensor, re-assemble with the partition arithmetic
bench.py uses, and must reproduce the single-rank frame bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, W, H, tile, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from oracle import pyoracle
    from radish_pt_amd import partition, scenes

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = scenes.cornell(segments=8, bands=6)
    cam = scenes.cornell_camera(W, H)
    o = pyoracle.OracleScene(sd)
    n = W * H
    full_d, full_i = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    frame_idx, packed_idx = partition.rank_pixels(W, H, rank, world, tile)
    for p in frame_idx:  # this rank's pixels only
        o.path_trace(cam, full_d, full_i, 0, 4, 3, pix=(int(p), int(p) + 1, 1))
    shard = partition.shard_elems(W, H, world, tile)
    send_d, send_i = torch.zeros(shard, 3), torch.zeros(shard, 3)
    send_d[packed_idx] = torch.from_numpy(full_d[frame_idx])
    send_i[packed_idx] = torch.from_numpy(full_i[frame_idx])
    gath_d, gath_i = torch.zeros(world * shard, 3), torch.zeros(world * shard, 3)
    dist.all_gather_into_tensor(gath_d, send_d)
    dist.all_gather_into_tensor(gath_i, send_i)
    src = torch.from_numpy(partition.untile_indices(W, H, world, tile))
    frame_d, frame_i = gath_d[src].numpy(), gath_i[src].numpy()
    rays = torch.tensor([float(o.stats()["closestRays"] + o.stats()["anyRays"])], dtype=torch.float64)
    dist.all_reduce(rays)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), d=frame_d, i=frame_i, rays=rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,tile", [(2, 16)])
def test_two_rank_tile_partition_gloo(tmp_path, world, tile):
    import torch.multiprocessing as mp

    from oracle import pyoracle
    from radish_pt_amd import scenes

    W, H = 40, 28
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, W, H, tile, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    sd = scenes.cornell(segments=8, bands=6)
    cam = scenes.cornell_camera(W, H)
    o = pyoracle.OracleScene(sd)
    ref_d, ref_i = np.zeros((W * H, 3), np.float32), np.zeros((W * H, 3), np.float32)
    o.path_trace(cam, ref_d, ref_i, 0, 4, 3)
    total = o.stats()["closestRays"] + o.stats()["anyRays"]
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(z["d"].view(np.uint32), ref_d.view(np.uint32)), f"rank {r} direct"
        assert np.array_equal(z["i"].view(np.uint32), ref_i.view(np.uint32)), f"rank {r} indirect"
        assert int(z["rays"][0]) == total  # whole-job ray count = sum over ranks
