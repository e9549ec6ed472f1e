#!/usr/bin/env python3
"""BASELINE config 4: teapots + 1024 emissive triangles, 1080p, ReSTIR DI (M = 32, temporal + spatial), one GPU.
Prints one JSON line per spatial-neighbour count (5 = reference, 4 = BASELINE.json's text)."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from radish_pt_amd import api, scenes

W, H = 1920, 1080
sd = scenes.teapots(emissive_grid=(16, 32))
cam = scenes.teapots_camera(W, H)
ctx = api.Context(0)
ctx.upload_scene(sd)
ctx.set_camera(cam)
dev = api.DevScene(); dev.ctx = ctx
for nsp in (5, 4):
    gb = api.GBuffer(); gb.create(W, H)
    img = torch.zeros(W * H, 3, device="cuda")
    ctx.restir_init()
    def frame(f, flags=0):
        gb.render(dev, cam)  # blocking, like the reference
        t_g = ctx.last_kernel_ms()
        ctx.restir_direct(img, 0, f, gb.c_struct(cam), 3, num_spatial=nsp, flags=flags)
        ctx.synchronize()
        t_r = ctx.last_kernel_ms()
        gb.update(cam)
        return t_g, t_r
    for f in range(3):
        frame(f)
    ctx.counters_reset()
    frame(3, api.RDH_PT_COUNT)
    c = ctx.counters()
    K = 8
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tg = tr = 0.0
    for f in range(4, 4 + K):
        a, b = frame(f); tg += a; tr += b
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    rays = c["closestRays"] + c["anyRays"]
    alg = 40 * c["closestRays"] + 28 * c["anyRays"] + 32 * c["nodeVisits"] + 36 * c["triTests"] + 64 * c["closestHits"] + 2384 * W * H
    print(json.dumps({"config": "teapots + 1024 emissive tris, 1080p, ReSTIR DI M=32 temporal+spatial", "spatial_neighbours": nsp,
                      "faithful_ris": 1, "tris": sd.num_prims, "lights": sd.num_lights, "ms_frame_wall": round(el / K * 1e3, 3),
                      "ms_gbuffer_kernel": round(tg / K, 3), "ms_restir_kernels": round(tr / K, 3), "rays_per_frame": rays,
                      "mrays_s": round(rays / (el / K) / 1e6, 1), "algorithmic_GBps_restir": round(alg / (tr / K * 1e-3) / 1e9, 1),
                      "finite": bool(torch.isfinite(img).all()), "mean": float(img.mean())}))
    ctx.restir_free()
