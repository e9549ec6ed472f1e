#!/usr/bin/env python3
"""Diagnostic: where ReSTIR pass 1's time goes on BASELINE config 4 — kernel time against the RIS candidate count M and
the reuse mask (M = 1 leaves the primary ray, one candidate and the shadow ray; the difference to M = 32 is the RIS loop)."""
import sys, torch
sys.path.insert(0, ".")
from radish_pt_amd import api, scenes
W, H = 1920, 1080
sd = scenes.teapots(emissive_grid=(16, 32)); cam = scenes.teapots_camera(W, H)
ctx = api.Context(0); ctx.upload_scene(sd); ctx.set_camera(cam)
dev = api.DevScene(); dev.ctx = ctx
gb = api.GBuffer(); gb.create(W, H, 0)
img = torch.zeros(W * H, 3, device="cuda")
for M in (32, 16, 8, 1):
    for reuse in (3, 0):
        ctx.restir_init(); ts = []
        for f in range(6):
            gb.render(dev, cam)
            ctx.restir_direct(img, 0, f, gb.c_struct(cam), reuse, ris_count=M); ctx.synchronize(); ts.append(ctx.last_kernel_ms())
            gb.update(cam)
        print(f"M={M:2d} reuse={reuse}: {min(ts[2:]):.3f} ms", flush=True)
        ctx.restir_free()
