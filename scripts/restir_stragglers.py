#!/usr/bin/env python3
"""Analysis: which rays does ReSTIR pass 1 set aside (literal-class, kernels_restir.h), how long are they, and is any long ray
left in the walker?  Config 4 frames on the GPU; the frame's own primary rays / shadow segments are read back
(rdh_restir_read_scratch) and their box-step counts taken from the CPU checker's per-ray histogram hook (analysis only).
usage: python scripts/restir_stragglers.py [frames]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from radish_pt_amd import api, scenes
from oracle import pyoracle

W, H = 1920, 1080
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sd = scenes.teapots(emissive_grid=(16, 32)); cam = scenes.teapots_camera(W, H)
ctx = api.Context(0); ctx.upload_scene(sd); ctx.set_camera(cam)
orc = pyoracle.OracleScene(sd); l = pyoracle.lib()
l.orc_debug_visit_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; l.orc_debug_visit_hist.restype = None
gb = api.GBuffer(); gb.create(W, H, 0)
img = torch.zeros(W * H, 3, device="cuda")
ctx.restir_init()


def lengths(arr, kind):
    """per-bucket counts of box steps (log2 buckets) of rays / segments on the checker"""
    rl = np.zeros(128, np.uint64)
    l.orc_debug_visit_hist(orc.h, None, rl.ctypes.data)
    (orc.trace_occluded if kind else orc.trace_closest)(arr)
    l.orc_debug_visit_hist(orc.h, None, None)
    return rl[kind * 32:kind * 32 + 32], rl[64 + kind * 32:64 + kind * 32 + 32]


for f in range(frames):
    ctx.set_camera(cam)
    ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), 0); ctx.synchronize()
    ctx.restir_direct(img, 0, f, gb.c_struct(cam), 3, num_spatial=5); ctx.synchronize()
    ms = ctx.last_kernel_ms()
    gb.update(cam)
    lists = ctx.restir_read_scratch(2)
    nP, nS = int(lists[0]), int(lists[1])
    segs = ctx.restir_read_scratch(1); rays = ctx.restir_read_scratch(0)
    out = {"frame": f, "ms_restir": round(ms, 3), "set_aside_primary": nP, "set_aside_shadow": nS}
    for name, arr, slots, kind in (("primary", rays, lists[4:4 + min(nP, 256)], 0), ("shadow", segs, lists[260:260 + min(nS, 256)], 1)):
        per = []
        for s in slots:
            cnt, sm = lengths(arr[int(s):int(s) + 1], kind)
            per.append(int(sm.sum()))
        out[name + "_set_aside_steps"] = sorted(per, reverse=True)[:8]
        ok = ~np.isnan(arr[:, 0])
        keep = np.ones(len(arr), bool); keep[np.asarray(slots, np.int64)] = False
        rest = arr[ok & keep]
        cnt, sm = lengths(rest, kind)
        out[name + "_walker_rays"] = int(cnt.sum())
        out[name + "_walker_long"] = {f">={1 << (b - 1)}": int(cnt[b]) for b in range(10, 32) if cnt[b]}
    print(out, flush=True)
