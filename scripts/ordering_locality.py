#!/usr/bin/env python3
"""Diagnostic: does the SIX-fold copy of the node array (one threaded ordering per dominant ray axis and sign) cost the walker
cache capacity?  k_walk_persistent over 8 M incoherent class-0 rays whose directions are (a) unrestricted — all six orderings
in flight at once — and (b) restricted to one dominant axis and sign — one ordering, a sixth of the node bytes."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from radish_pt_amd import api, scenes
N = 8 << 20
rng = np.random.default_rng(5)
for name in ("cornell", "teapots"):
    sd = scenes.cornell() if name == "cornell" else scenes.teapots()
    lo, hi = sd.vertices.min(0), sd.vertices.max(0)
    ctx = api.Context(0); ctx.upload_scene(sd)
    for label in ("all six orderings", "one ordering"):
        o = rng.uniform(lo, hi, (N, 3)).astype(np.float32)
        d = rng.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        if label == "one ordering":  # make -z the dominant component
            a = np.abs(d); k = a.argmax(1)
            idx = np.arange(N)
            big = d[idx, k].copy(); d[idx, k] = d[idx, 2]; d[idx, 2] = -np.abs(big)
        d = np.where(np.abs(d) < 1e-3, 1e-3, d); d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = torch.from_numpy(np.concatenate([o, d.astype(np.float32)], 1).astype(np.float32)).cuda()
        hits = torch.zeros(N, 4, dtype=torch.int32, device="cuda")
        ctx.counters_reset(); ctx.trace_closest(rays, hits, api.RDH_PT_COUNT | api.RDH_PT_PERSISTENT); ctx.synchronize()
        c = ctx.counters()
        ts = []
        for r in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.trace_closest(rays, hits, api.RDH_PT_PERSISTENT); ctx.synchronize(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        print(f"{name} [{label}]: {c['nodeVisits'] / N:.1f} steps per ray, {t * 1e3:.2f} ms -> {c['nodeVisits'] / t / 1e9:.0f} G box steps/s", flush=True)
    ctx.close()
