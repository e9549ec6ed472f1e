#!/usr/bin/env python3
"""What ONE rank of an N-GPU ReSTIR job computes per frame, measured on one GPU (no collective): the G-buffer records of its
tiles + ReSTIRDirect on its tiles (pass 1 on tiles + 8-px apron, pass 2 on tiles), for N = 1, 2, 4, 8 and ownership tiles of 64
and 128 px.  The whole frame's G-buffer is rendered once by a one-rank context first, so that the rank's neighbours' records
exist as they would after the exchange.  usage: partition_times_restir.py [scene] [W H]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radish_pt_amd import api
import bench

scene = sys.argv[1] if len(sys.argv) > 1 else "teasets_1m"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
K = 6
sd = bench.make_scene(scene)
cam = bench.make_camera(scene, W, H)
dev = torch.device("cuda", 0)
full = api.Context(0); full.upload_scene(sd); full.set_camera(cam)
rows = []
for tile in (64, 128):
    for world in (1, 2, 4, 8):
        worst = None
        for rank in sorted({0, world - 1}):
            ctx = api.Context(0)
            ctx.upload_scene(sd); ctx.set_camera(cam); ctx.set_partition(rank, world, tile)
            gb = api.GBuffer(); gb.create(W, H, 0)
            n = W * H if world == 1 else ctx.tiles_per_rank() * tile * tile
            img = torch.zeros(n, 3, device=dev)
            ctx.restir_init()
            tg = tr = 0.0
            for f in range(2 + K):
                full.gbuffer_render(gb.c_struct(cam_fallback=cam), 0); full.synchronize()
                ctx.set_camera(cam)
                ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), api.RDH_PT_PARTITION_GBUFFER if world > 1 else 0)
                ctx.synchronize(); a = ctx.last_kernel_ms()
                ctx.restir_direct(img, 0, f, gb.c_struct(cam), 3)
                ctx.synchronize(); b = ctx.last_kernel_ms()
                gb.update(cam)
                if f >= 2:
                    tg += a; tr += b
            ms = (tg + tr) / K
            if worst is None or ms > worst[0]:
                worst = (ms, tg / K, tr / K)
            ctx.restir_free(); ctx.close()
        rows.append({"tile": tile, "world": world, "ms_rank_share": round(worst[0], 4), "ms_gbuffer": round(worst[1], 4), "ms_restir": round(worst[2], 4)})
        print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"scene": scene, "W": W, "H": H, "rows": rows}))
