#!/usr/bin/env python3
"""Two (or more) real ranks over RCCL: every multi-GPU path of the repository against the one-GPU frame, bit for bit.

Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/multi_rank_check.py
(tests/test_gpu_collectives.py::test_two_ranks_over_rccl does, where two GPUs are visible).  Checked on every rank:
  1. bench.py's step: rdh_path_trace into packed tiles → torch.distributed all_gather_into_tensor → rdh_untile;
  2. the library's own collectives: rdh_comm_init (the 128-byte id travels over torch.distributed's object broadcast) →
     rdh_path_trace_gathered (3 accumulated frames);
  3. ReSTIR DI: partitioned G-buffer + rdh_gbuffer_exchange, rdh_restir_direct_gathered (image all-gather + reservoir
     exchange), 3 frames with a moving camera.
Prints "multi_rank_check ok" on rank 0.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from radish_pt_amd import api, hostlib, scenes  # noqa: E402


def bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend="nccl", device_id=dev)
    tile, depth = 32, 4

    # ---- 1 + 2: pathTrace ----
    sd = scenes.cornell(segments=16, bands=12)
    W, H = 200, 120
    cam = scenes.cornell_camera(W, H)
    ref = api.Context(local)
    ref.upload_scene(sd)
    ref.set_camera(cam)
    rd, ri = torch.zeros(W * H, 3, device=dev), torch.zeros(W * H, 3, device=dev)
    ctx = api.Context(local)
    ctx.upload_scene(sd)
    ctx.set_camera(cam)
    ctx.set_partition(rank, world, tile)
    shard = ctx.tiles_per_rank() * tile * tile
    d, i = torch.zeros(shard, 3, device=dev), torch.zeros(shard, 3, device=dev)
    gd, gi = torch.zeros(world * shard, 3, device=dev), torch.zeros(world * shard, 3, device=dev)
    fd, fi = torch.zeros(W * H, 3, device=dev), torch.zeros(W * H, 3, device=dev)
    ref.path_trace(rd, ri, 0, 9, depth)
    ctx.path_trace(d, i, 0, 9, depth)  # bench.py's step
    dist.all_gather_into_tensor(gd, d)
    dist.all_gather_into_tensor(gi, i)
    ctx.untile(gd, fd)
    ctx.untile(gi, fi)
    torch.cuda.synchronize()
    assert np.array_equal(bits(fd), bits(rd)) and np.array_equal(bits(fi), bits(ri)), "bench.py step differs from the one-GPU frame"

    uid = [api.Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx.comm_init(uid[0], rank, world)
    rd.zero_(); ri.zero_()
    ld, li = torch.zeros(W * H, 3, device=dev), torch.zeros(W * H, 3, device=dev)
    for it in range(3):
        ref.path_trace(rd, ri, it, 20 + it, depth)
        ctx.path_trace_gathered(ld, li, it, 20 + it, depth)
    torch.cuda.synchronize()
    assert np.array_equal(bits(ld), bits(rd)) and np.array_equal(bits(li), bits(ri)), "rdh_path_trace_gathered differs"
    ctx.close()
    ref.close()

    # ---- 3: ReSTIR DI ----
    sd = scenes.teapots(segments=12, bands=8, grid=2, emissive_grid=(4, 8))
    W, H = 150, 90
    n = W * H
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.08 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0) for f in range(3)]
    ref, ctx = api.Context(local), api.Context(local)
    for c in (ref, ctx):
        c.upload_scene(sd)
        c.set_camera(cams[0])
        c.restir_init()
    ctx.set_partition(rank, world, tile)
    ctx.comm_init(_fresh_id(rank), rank, world)
    gb_ref, gb = api.GBuffer(), api.GBuffer()
    gb_ref.create(W, H, local)
    gb.create(W, H, local)
    img_ref, img = torch.zeros(n, 3, device=dev), torch.zeros(n, 3, device=dev)
    for f, cam in enumerate(cams):
        ref.set_camera(cam)
        ctx.set_camera(cam)
        ref.gbuffer_render(gb_ref.c_struct(cam_fallback=cam), 0)
        ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), api.RDH_PT_PARTITION_GBUFFER)
        ctx.gbuffer_exchange(gb.c_struct(cam_fallback=cam))
        ref.restir_direct(img_ref, 0, 40 + f, gb_ref.c_struct(cam), 3)
        ctx.restir_direct_gathered(img, 0, 40 + f, gb.c_struct(cam), 3)
        torch.cuda.synchronize()
        k = gb.frameIdx
        for a, b in ((gb.albedo, gb_ref.albedo), (gb.normal[k], gb_ref.normal[k]), (gb.depth[k], gb_ref.depth[k]),
                     (gb.primId[k], gb_ref.primId[k]), (gb.motion, gb_ref.motion)):
            assert np.array_equal(bits(a), bits(b)), f"G-buffer exchange differs, frame {f}"
        assert np.array_equal(bits(img), bits(img_ref)), f"rdh_restir_direct_gathered differs, frame {f}"
        assert ctx.restir_read(1).tobytes() == ref.restir_read(1).tobytes(), f"reservoir exchange differs, frame {f}"
        gb_ref.update(cam)
        gb.update(cam)
    ctx.close()
    ref.close()
    dist.barrier()
    if rank == 0:
        print(f"multi_rank_check ok: {world} ranks", flush=True)
    dist.destroy_process_group()


def _fresh_id(rank):
    uid = [api.Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    return uid[0]


if __name__ == "__main__":
    main()
