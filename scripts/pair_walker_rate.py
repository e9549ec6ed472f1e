#!/usr/bin/env python3
"""k_walk_persistent (six threaded node arrays, 192 B per tree node) against k_walk_pair (sibling pairs: ONE shared 64-byte record per
inner node, both children per round trip; round 3 also measured a one-node-per-step walk over a shared tree, profiles/r03_g_*, r03_h_*)
on each scene's own frame rays (rdh_dump_rays: every closest-hit ray
and every occlusion segment of a depth-8 frame).  Records and counters are compared first; then hipEvent time of each, mean of 5.
usage: tree_walker_rate.py [scene ...]   scenes: cornell teapots teasets_1m"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from radish_pt_amd import api, scenes
names = sys.argv[1:] or ["cornell", "teapots", "teasets_1m"]
for name in names:
    if name == "cornell":
        sd, cam, W, H = scenes.cornell(), scenes.cornell_camera, 1920, 1080
    elif name == "teapots":
        sd, cam, W, H = scenes.teapots(), scenes.teapots_camera, 1920, 1080
    else:
        sd, cam, W, H = scenes.teapots(segments=200, bands=156, emissive_grid=(16, 32)), scenes.teapots_camera, 3840, 2160
    ctx = api.Context(0)
    ctx.upload_scene(sd)
    ctx.set_camera(cam(W, H))
    closest, segs = ctx.dump_rays(3, 8)
    nC, nA = closest.shape[0], segs.shape[0]
    res = {}
    for label, fl in (("threaded", api.RDH_PT_PERSISTENT | api.RDH_PT_NO_PAIRS), ("pairs", api.RDH_PT_PERSISTENT | api.RDH_PT_PAIRS)):
        hits = torch.zeros(nC, 4, dtype=torch.int32, device="cuda")
        occ = torch.zeros(max(nA, 1), dtype=torch.int32, device="cuda")
        ctx.counters_reset()
        ctx.trace_closest(closest, hits, fl | api.RDH_PT_COUNT)
        ctx.trace_occluded(segs, occ, fl | api.RDH_PT_COUNT)
        ctx.synchronize()
        c = ctx.counters()
        ta = tb = 0.0
        reps = 5
        for r in range(reps + 1):
            ctx.trace_closest(closest, hits, fl); a = ctx.last_kernel_ms()
            ctx.trace_occluded(segs, occ, fl); b = ctx.last_kernel_ms()
            if r:
                ta += a; tb += b
        ta /= reps; tb /= reps
        res[label] = (hits.clone(), occ.clone(), c, ta, tb)
        print(f"{name} {W}x{H} [{label}]: closest {nC} rays {ta:.3f} ms, any {nA} segs {tb:.3f} ms, total {ta + tb:.3f} ms -> "
              f"{c['nodeVisits'] / ((ta + tb) * 1e-3) / 1e9:.1f} G box steps/s, {(nC + nA) / ((ta + tb) * 1e-3) / 1e6:.0f} Mrays/s", flush=True)
    a = res["threaded"]
    for k in ("pairs",):
        b = res[k]
        same = bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2])
        print(f"{name}: {k}: records and counters equal: {same}; {k} / threaded time = {(b[3] + b[4]) / (a[3] + a[4]):.3f} "
              f"(closest {b[3] / a[3]:.3f}, any {b[4] / a[4]:.3f})", flush=True)
    ctx.close()
