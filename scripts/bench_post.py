#!/usr/bin/env python3
"""Kernel times of the rows either side of the hot path (SURVEY §8f N3 display, N4 denoisers) at 1920x1080 on the teapots
frame: per launch, with the algorithmic bytes of each (planes read once + planes written) and the rate they imply.
Prints one JSON object; the table in DESIGN.md §10 comes from it."""
import json, sys
import numpy as np, torch
sys.path.insert(0, ".")
from radish_pt_amd import api, scenes

W, H = 1920, 1080
n = W * H
sd = scenes.teapots(emissive_grid=(16, 32)); cam = scenes.teapots_camera(W, H)
ctx = api.Context(0); ctx.upload_scene(sd); ctx.set_camera(cam)
dev = api.DevScene(); dev.ctx = ctx
gb = api.GBuffer(); gb.create(W, H, 0)
gb.render(dev, cam); gb.update(cam); gb.render(dev, cam)
direct = torch.zeros(n, 3, device="cuda"); indirect = torch.zeros(n, 3, device="cuda")
for it in range(4):
    ctx.path_trace(direct, indirect, it, 17 + it, 8, api.RDH_PT_PERSISTENT)
ctx.synchronize()
noisy = (direct + indirect).contiguous()
gbc = gb.c_struct(cam)
out = {}

def timed(name, fn, bytes_per_px, reps=3, K=20):
    import time
    ts = []
    for _ in range(reps):  # K back-to-back launches on the context's stream, one synchronise (these entries do not time themselves)
        fn(); ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(K): fn()
        ctx.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e3)
    ms = min(ts)
    out[name] = {"ms": round(ms, 4), "alg_bytes_per_px": bytes_per_px, "GBps": round(bytes_per_px * n / (ms * 1e-3) / 1e9, 1)}

a = torch.zeros(n, 3, device="cuda"); b = torch.zeros(n, 3, device="cuda")
var = torch.rand(n, device="cuda"); var2 = torch.zeros(n, device="cuda"); fvar = torch.zeros(n, device="cuda")
mom = torch.rand(n, 3, device="cuda"); mom2 = torch.zeros(n, 3, device="cuda")
pbo = torch.zeros(n, dtype=torch.int32, device="cuda")
# kind 0: vec3 image (tone-mapped), kind 2: float plane as grey, kind 3: int pixel indices (the motion view); kind 1 takes vec2 images
for kind, nm, src in ((0, "float3 image", noisy), (2, "float plane", gb.depth[gb.frameIdx]), (3, "motion", gb.motion)):
    timed(f"copy_image_to_pbo[{nm}]", lambda: ctx.copy_image_to_pbo(pbo, src, W, H, kind, 2 if kind == 0 else 0, 1.0), (12 if kind == 0 else 4) + 4)
for lv in (0, 2, 4):
    timed(f"eaw_filter[level {lv}]", lambda: ctx.denoise_eaw(a, noisy, gbc, cam, 64.0, 0.2, 1.0, lv), 12 + 12 + 4 + 4 + 12)
    timed(f"svgf_filter[level {lv}]", lambda: ctx.denoise_svgf(a, noisy, var2, var, fvar, gbc, cam, 4.0, 128.0, 1.0, lv), 12 + 12 + 4 + 4 + 4 + 4 + 12 + 4)
timed("temporal_accumulate", lambda: ctx.denoise_temporal_accumulate(a, noisy, mom2, mom, noisy, gbc, 0), 12 + 12 + 12 + 4 + 4 + 4 + 12 + 12 + 12 + 12)
timed("estimate_variance", lambda: ctx.denoise_estimate_variance(var2, mom, W, H), 16)
timed("filter_variance", lambda: ctx.denoise_filter_variance(fvar, var, W, H), 8)
timed("modulate", lambda: ctx.denoise_modulate(a, gbc), 36)
timed("add", lambda: ctx.denoise_add(b, a, noisy, W, H), 36)
print(json.dumps(out, indent=1))
