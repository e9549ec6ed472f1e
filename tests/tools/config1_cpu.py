#!/usr/bin/env python3
"""BASELINE config 1: Cornell stand-in, 400x400, 4 bounces, 16 spp (iter = looper = 0..15), `pathTrace` on the single-threaded
CPU oracle — the reference's own CPU-runnable case — and, when a GPU is present, the same 16 accumulated frames through
libradish_hip.so (persistent kernel), compared bit for bit.  One JSON line."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import pyoracle
from radish_pt_amd import scenes

W = H = 400
DEPTH, SPP = 4, 16
sd = scenes.cornell()
cam = scenes.cornell_camera(W, H)
o = pyoracle.OracleScene(sd)
d = np.zeros((W * H, 3), np.float32)
i = np.zeros((W * H, 3), np.float32)
t0 = time.perf_counter()
for it in range(SPP):
    o.path_trace(cam, d, i, it, it, DEPTH)
cpu_s = time.perf_counter() - t0
st = o.stats()
rays = st["closestRays"] + st["anyRays"]
out = {"config": f"cornell stand-in ({sd.num_prims} tris), {W}x{H}, {DEPTH} bounces, {SPP} spp, CPU oracle 1 thread",
       "rays": rays, "cpu_seconds": round(cpu_s, 2), "cpu_mrays_s": round(rays / cpu_s / 1e6, 4),
       "mean_direct": float(d.mean()), "mean_indirect": float(i.mean())}
try:
    import torch
    if torch.cuda.is_available():
        from radish_pt_amd import api
        ctx = api.Context(0); ctx.upload_scene(sd); ctx.set_camera(cam)
        gd = torch.zeros(W * H, 3, device="cuda"); gi = torch.zeros(W * H, 3, device="cuda")
        ctx.counters_reset()
        ctx.synchronize(); t0 = time.perf_counter()
        for it in range(SPP):
            ctx.path_trace(gd, gi, it, it, DEPTH, api.RDH_PT_PERSISTENT | api.RDH_PT_COUNT)
        ctx.synchronize(); gpu_s = time.perf_counter() - t0
        ct = ctx.counters()
        out["gpu_ms_per_spp"] = round(gpu_s / SPP * 1e3, 3)
        out["gpu_bit_exact"] = bool(np.array_equal(gd.cpu().numpy().view(np.uint32), d.view(np.uint32))
                                    and np.array_equal(gi.cpu().numpy().view(np.uint32), i.view(np.uint32)))
        out["gpu_counters_equal"] = all(ct[k] == st[k] for k in ("closestRays", "anyRays", "nodeVisits", "triTests", "closestHits"))
except ImportError:
    pass
print(json.dumps(out))
