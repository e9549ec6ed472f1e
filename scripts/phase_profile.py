#!/usr/bin/env python3
"""Diagnostic: where the persistent kernel's wave time goes (needs a -DRD_PERSIST_PHASES build via RADISH_HIP_LIB)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from radish_pt_amd import api, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
sd = scenes.cornell() if scene == "cornell" else scenes.teapots()
W, H = 1920, 1080
cam = scenes.cornell_camera(W, H) if scene == "cornell" else scenes.teapots_camera(W, H)
ctx = api.Context(0); ctx.upload_scene(sd); ctx.set_camera(cam)
d = torch.zeros(W*H, 3, device='cuda'); i = torch.zeros(W*H, 3, device='cuda')
for it in range(3):
    ctx.path_trace(d, i, 0, it, 8, api.RDH_PT_PERSISTENT)
ctx.synchronize()
print("kernel ms", ctx.last_kernel_ms())
p = ctx.debug_persist_phases().astype(np.float64)
names = ["raygen", "whole-wave rays", "box loop", "leaf tests", "retire", "shading"]
tot = p[6]
print(f"sum of wave lifetimes {tot / 100:.0f} us over all waves ({tot / 100 / 3072:.0f} us per wave if 3072)")
for k, n in enumerate(names):
    print(f"  {n:16s} {100 * p[k] / tot:5.1f} % of wave time")
print(f"box: {p[8]:.3g} wave-steps, {p[9]:.3g} lane-steps, {p[9] / max(p[8], 1):.1f} lanes per step; {p[2] / max(p[8], 1) * 10:.0f} ns per wave-step")
print(f"leaf: {p[10]:.3g} calls, {p[11] / max(p[10], 1):.1f} lanes per call; {p[3] / max(p[10], 1) * 10:.0f} ns per call")
print(f"shade: {p[12]:.3g} calls, {p[13] / max(p[12], 1):.1f} lanes per call; {p[5] / max(p[12], 1) * 10:.0f} ns per call")
print(f"raygen: {p[14]:.3g} calls, {p[15] / max(p[14], 1):.1f} idle lanes per call; {p[0] / max(p[14], 1) * 10:.0f} ns per call")
