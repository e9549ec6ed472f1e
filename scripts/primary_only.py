#!/usr/bin/env python3
"""Primary-ray-only timings on the teapots scene: G-buffer kernel vs depth-0 pathTrace in each mode."""
import sys, torch
sys.path.insert(0, ".")
from radish_pt_amd import api, scenes
W, H = 1920, 1080
import os
if os.environ.get("SCENE") == "cornell":
    sd = scenes.cornell(); cam = scenes.cornell_camera(W, H)
else:
    sd = scenes.teapots(emissive_grid=(16, 32)); cam = scenes.teapots_camera(W, H)
ctx = api.Context(0); ctx.upload_scene(sd); ctx.set_camera(cam)
d = torch.zeros(W*H, 3, device="cuda"); i = torch.zeros(W*H, 3, device="cuda")
for name, fl in (("mega", 0), ("wavefront", api.RDH_PT_WAVEFRONT), ("persistent", api.RDH_PT_PERSISTENT)):
    ts = []
    for it in range(4):
        ctx.path_trace(d, i, 0, it, 0, fl); ctx.synchronize(); ts.append(ctx.last_kernel_ms())
    print(f"pathTrace depth 0 ({name}): {min(ts):.3f} ms")
gb = api.GBuffer(); gb.create(W, H); dev = api.DevScene(); dev.ctx = ctx
ts = []
for it in range(4):
    gb.render(dev, cam); ts.append(ctx.last_kernel_ms())
print(f"G-buffer: {min(ts):.3f} ms")
ts = []
for it in range(4):
    ctx.path_trace_direct(d, 0, it); ctx.synchronize(); ts.append(ctx.last_kernel_ms())
print(f"pathTraceDirect: {min(ts):.3f} ms")
