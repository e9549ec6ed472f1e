#!/usr/bin/env python3
"""N plain frames of pathTrace (no counters, no warm-up distinction) or of the ReSTIR frame for a `rocprofv3 --pmc` / `--kernel-trace`
pass: every dispatch of the run belongs to one of the N frames, so per-frame figures are totals / N.
usage: pmc_frames.py <scene> <mode|restir> <W> <H> <frames>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radish_pt_amd import api
import bench

scene, mode, W, H, N = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sd = bench.make_scene(scene)
cam = bench.make_camera(scene, W, H)
ctx = api.Context(0)
ctx.upload_scene(sd)
ctx.set_camera(cam)
if mode == "restir":
    gb = api.GBuffer(); gb.create(W, H, 0)
    img = torch.zeros(W * H, 3, device="cuda")
    ctx.restir_init()
    for f in range(N):
        ctx.set_camera(cam)
        ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), 0)
        ctx.synchronize()
        ctx.restir_direct(img, 0, f, gb.c_struct(cam), 3)
        ctx.synchronize()
        gb.update(cam)
else:
    d = torch.zeros(W * H, 3, device="cuda"); i = torch.zeros(W * H, 3, device="cuda")
    flags = bench.mode_flags(api, mode)
    for s in range(N):
        ctx.path_trace(d, i, 0, s, 8, flags)
    ctx.synchronize()
print(f"{N} frames of {scene} {mode} {W}x{H} done")
ctx.close()
