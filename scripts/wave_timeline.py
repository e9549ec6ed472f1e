#!/usr/bin/env python3
"""Diagnostic: timeline of the persistent kernel's waves (needs a -DRD_PERSIST_STAMPS build via RADISH_HIP_LIB)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from radish_pt_amd import api, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # this rank's share of an N-GPU job (rank 0), no collective
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sd = scenes.cornell() if scene == "cornell" else scenes.teapots()
W, H = 1920, 1080
cam = scenes.cornell_camera(W, H) if scene == "cornell" else scenes.teapots_camera(W, H)
ctx = api.Context(0); ctx.upload_scene(sd); ctx.set_camera(cam); ctx.set_partition(rank, world, 64)
npx = W * H if world == 1 else ctx.tiles_per_rank() * 64 * 64
d = torch.zeros(npx, 3, device='cuda'); i = torch.zeros(npx, 3, device='cuda')
print(f"scene {scene}, rank {rank} of {world}: {npx} pixels")
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
lit_us, lit_n = st[3, :n] / 100.0, st[4, :n]
print(f"literal-class rays traced whole: {int(lit_n.sum())} in {int((lit_n > 0).sum())} waves; per wave with one: median {np.median(lit_us[lit_n > 0]) if (lit_n > 0).any() else 0:.1f} us, max {lit_us.max():.1f} us")
order = np.argsort(end)[::-1]
for k in (16, 64, 256):
    top = order[:k]
    print(f"  the {k:3d} waves that end last: end >= {end[top].min():.1f} us; {int((lit_n[top] > 0).sum())} of them traced a literal-class ray (mean {lit_us[top].mean():.1f} us there); whole launch: {100 * (lit_n > 0).mean():.1f} % of waves")
# what the launch would look like if those waves had ended `lit_us` earlier
print(f"  span if literal traces were free: {np.max(end - lit_us):.1f} us (now {end.max():.1f})")
