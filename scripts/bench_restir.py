#!/usr/bin/env python3
"""BASELINE config 4 (and config 5's structure): teapots + 1024 emissive triangles, 1080p, ReSTIR DI (M = 32, temporal +
spatial), G-buffer + two ReSTIR passes per frame.  Prints one JSON line per spatial-neighbour count (5 = reference,
4 = BASELINE.json's text).

One GPU:   python scripts/bench_restir.py
N GPUs:    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_restir.py
           one rank per GPU; the frame is cut into 64x64 tiles (tile t -> rank t % N).  Every rank renders the G-buffer records
           of ITS tiles and the planes are completed by ONE all-gather of 36 B per pixel (RADISH_GBUFFER_REPLICATED=1: every
           rank renders the whole frame instead, no exchange); it shades its tiles (pass 1 on the tiles + an 8-px apron, pass 2
           on the tiles), then ONE all-gather of the packed image tiles and ONE of the packed reservoirs (next frame's temporal
           reuse) per frame over RCCL.  RADISH_RESTIR_FUSED=1 times round 1's fused pass-1 kernel instead of the split one.
           (RADISH_DIST_BACKEND=gloo RADISH_FORCE_DEVICE=0 rehearses N ranks on one GPU.)"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from radish_pt_amd import api, scenes

world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
dist = None
dev_index = 0
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = int(os.environ.get("RADISH_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("RADISH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group(backend=backend)
dev_t = torch.device("cuda", dev_index)
W, H, TILE = int(os.environ.get("RADISH_W", 1920)), int(os.environ.get("RADISH_H", 1080)), 64
sd = scenes.teapots(emissive_grid=(16, 32))
cam = scenes.teapots_camera(W, H)
ctx = api.Context(dev_index)
ctx.upload_scene(sd)
ctx.set_camera(cam)
ctx.set_partition(rank, world, TILE)
dev = api.DevScene(); dev.ctx = ctx
n_local = W * H if world == 1 else ctx.tiles_per_rank() * TILE * TILE
RESTIR_FLAGS = api.RDH_PT_RESTIR_FUSED if os.environ.get("RADISH_RESTIR_FUSED") == "1" else 0
GB_PARTITION = world > 1 and os.environ.get("RADISH_GBUFFER_REPLICATED") != "1"


def gather(t):
    if dist.get_backend() == "gloo":  # rehearsal: gloo moves host memory
        h = t.cpu(); out = torch.empty(world * h.numel(), dtype=h.dtype).view(world * h.shape[0], *h.shape[1:])
        dist.all_gather_into_tensor(out, h); return out.to(dev_t)
    out = torch.empty(world * t.shape[0], *t.shape[1:], dtype=t.dtype, device=dev_t)
    dist.all_gather_into_tensor(out, t); return out


for nsp in (5, 4):
    gb = api.GBuffer(); gb.create(W, H, dev_index)
    img = torch.zeros(n_local, 3, device=dev_t)
    frame_img = torch.zeros(W * H, 3, device=dev_t)
    packed_resv = torch.zeros(n_local, 9, device=dev_t)
    packed_gb = torch.zeros(n_local, 9, device=dev_t)
    ctx.restir_init()
    def frame(f, flags=0):
        ctx.set_camera(cam)
        ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), api.RDH_PT_PARTITION_GBUFFER if GB_PARTITION else 0)
        ctx.synchronize()  # blocking, like the reference's GBuffer::render
        t_g = ctx.last_kernel_ms()
        if GB_PARTITION:
            ctx.gbuffer_exchange_pack(gb.c_struct(cam_fallback=cam), packed_gb)
            ctx.synchronize()
            ctx.gbuffer_exchange_unpack(gb.c_struct(cam_fallback=cam), gather(packed_gb))
        ctx.restir_direct(img, 0, f, gb.c_struct(cam), 3, num_spatial=nsp, flags=flags | RESTIR_FLAGS)
        ctx.synchronize()
        t_r = ctx.last_kernel_ms()
        if world > 1:
            ctx.untile(gather(img), frame_img)
            ctx.restir_exchange_pack(packed_resv)
            ctx.synchronize()
            ctx.restir_exchange_unpack(gather(packed_resv))
            ctx.synchronize()
        gb.update(cam)
        return t_g, t_r
    for f in range(3):
        frame(f)
    ctx.counters_reset()
    frame(3, api.RDH_PT_COUNT)
    c = ctx.counters()
    K = 8
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    t0 = time.perf_counter()
    tg = tr = 0.0
    for f in range(4, 4 + K):
        a, b = frame(f); tg += a; tr += b
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    el = time.perf_counter() - t0
    stats = torch.tensor([el, float(c["closestRays"] + c["anyRays"])], dtype=torch.float64)
    if world > 1:
        mx = stats.clone(); sm = stats.clone()
        if dist.get_backend() != "gloo": mx = mx.to(dev_t); sm = sm.to(dev_t)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        el, rays = float(mx[0]), float(sm[1])
    else:
        rays = float(stats[1])
    final = frame_img if world > 1 else img
    if rank == 0:
        alg = 40 * c["closestRays"] + 28 * c["anyRays"] + 32 * c["nodeVisits"] + 36 * c["triTests"] + 64 * c["closestHits"] + 2384 * W * H
        print(json.dumps({"config": f"teapots + 1024 emissive tris, {W}x{H}, ReSTIR DI M=32 temporal+spatial", "n_gpus": world, "spatial_neighbours": nsp,
                          "faithful_ris": 1, "pass1": "fused" if RESTIR_FLAGS else "split", "gbuffer": "partitioned + all-gather" if GB_PARTITION else "whole frame per rank", "tris": sd.num_prims, "lights": sd.num_lights, "ms_frame_wall": round(el / K * 1e3, 3),
                          "ms_gbuffer_kernel": round(tg / K, 3), "ms_restir_kernels": round(tr / K, 3), "rays_per_frame": rays,
                          "mrays_s": round(rays / (el / K) / 1e6, 1),
                          "algorithmic_GBps_restir_rank0": round(alg / (tr / K * 1e-3) / 1e9, 1),
                          "finite": bool(torch.isfinite(final).all()), "mean": float(final.mean())}), flush=True)
    ctx.restir_free()
if world > 1:
    dist.barrier(); dist.destroy_process_group()
