#!/usr/bin/env python3
"""Diagnostic: how many box steps per second does a WALK-ONLY kernel reach?  rdh_trace_closest (k_trace_closest: 60 VGPRs,
8 waves per SIMD, one lane per ray, no shading in its register allocation) over a large batch of incoherent rays with
class-0 directions only (no whole-wave traces), Cornell and teapots scenes.  Compare with k_pt_persistent's ~167 G steps/s
in the bulk of a frame at 3 waves per SIMD."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from radish_pt_amd import api, scenes
N = 8 << 20
rng = np.random.default_rng(5)
for name in ("cornell", "teapots"):
    sd = scenes.cornell() if name == "cornell" else scenes.teapots()
    lo, hi = sd.vertices.min(0), sd.vertices.max(0)
    o = rng.uniform(lo, hi, (N, 3)).astype(np.float32)
    d = rng.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(np.abs(d) < 1e-3, 1e-3, d); d /= np.linalg.norm(d, axis=1, keepdims=True)  # keep every ray in class 0
    rays = torch.from_numpy(np.concatenate([o, d.astype(np.float32)], 1).astype(np.float32)).cuda()
    hits = torch.zeros(N, 4, dtype=torch.int32, device="cuda")
    ctx = api.Context(0); ctx.upload_scene(sd)
    ctx.counters_reset(); ctx.trace_closest(rays, hits, api.RDH_PT_COUNT); ctx.synchronize()
    c = ctx.counters()
    for flags, kname in ((0, "one lane per ray"), (api.RDH_PT_PERSISTENT, "lane refill (k_walk_persistent)")):
        ts = []
        for r in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.trace_closest(rays, hits, flags); ctx.synchronize(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        print(f"{name} [{kname}]: {N} rays, {c['nodeVisits'] / N:.1f} steps per ray, {t * 1e3:.2f} ms -> {c['nodeVisits'] / t / 1e9:.0f} G box steps/s, "
              f"{N / t / 1e6:.0f} Mrays/s, algorithmic {(40 * N + 32 * c['nodeVisits'] + 36 * c['triTests'] + 64 * c['closestHits']) / t / 1e9:.0f} GB/s", flush=True)
    ctx.close()
