#!/usr/bin/env python3
"""Diagnostic: timeline of the persistent kernel's waves (needs a -DRD_PERSIST_STAMPS build via RADISH_HIP_LIB)."""
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
st = ctx.debug_persist_stamps().astype(np.float64)
n = int((st[0] > 0).sum())
t0 = st[0, :n].min()
start, dry, end = [(st[k, :n] - t0) / 100.0 for k in range(3)]
print(f"waves {n}; kernel span {end.max():.1f} us; starts within {start.max():.1f} us")
print("dry  (us): min %.1f  median %.1f  max %.1f" % (dry.min(), np.median(dry), dry.max()))
print("end  (us): p10 %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(np.percentile(end, [10, 50, 90, 99, 100])))
print("mean wave lifetime / span = %.3f" % ((end - start).mean() / end.max()))
for frac in (0.5, 0.25, 0.1, 0.02):
    # time at which only `frac` of the waves are still running
    print(f"  waves still running <= {frac:4.0%} after {np.percentile(end, 100 * (1 - frac)):.1f} us")
