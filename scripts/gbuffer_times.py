#!/usr/bin/env python3
"""Diagnostic: G-buffer kernel time for slightly different cameras (rare literal-class rays decide it)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from radish_pt_amd import api, scenes, hostlib
W, H = 1920, 1080
sd = scenes.teapots(emissive_grid=(16, 32))
ctx = api.Context(0); ctx.upload_scene(sd)
gb = api.GBuffer(); gb.create(W, H)
for k in range(8):
    cam = hostlib.make_camera(W, H, eye=(0.3 + 0.013 * k, 1.9, 7.4), rotation=(-91.5 + 0.07 * k, -11.0, 0.0), fovy=19.0)
    ctx.set_camera(cam)
    for flags, name in ((0, "persistent"), (api.RDH_PT_NO_DEFER, "no-defer"), (api.RDH_PT_MEGA_GBUFFER, "one-lane")):
        ts = []
        for r in range(3):
            ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), flags); ctx.synchronize(); ts.append(ctx.last_kernel_ms())
        print(f"cam {k} {name:10s} {min(ts):.3f} ms")
cam = scenes.teapots_camera(W, H); ctx.set_camera(cam)
ctx.counters_reset(); ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), api.RDH_PT_COUNT); ctx.synchronize()
print("teapots_camera", ctx.last_kernel_ms(), ctx.counters())
for flags, name in ((0, "persistent"), (api.RDH_PT_NO_DEFER, "no-defer")):
    ts = []
    for r in range(5):
        ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), flags); ctx.synchronize(); ts.append(ctx.last_kernel_ms())
    print(f"teapots_camera {name:10s} {min(ts):.3f} ms")
