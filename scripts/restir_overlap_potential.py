#!/usr/bin/env python3
"""Analysis: how much would ReSTIR's launches gain from running beside each other?  Config 4 on ONE context (launches back to
back on one stream) against TWO independent contexts on two streams, each rendering its own sequence (not something a renderer can
do with consecutive frames — the temporal reservoirs chain them; this only measures how well the walks, the RIS kernel and the
tails of the launches overlap)."""
import json, sys, time
import torch
sys.path.insert(0, ".")
from radish_pt_amd import api, scenes

W, H = 1920, 1080
sd = scenes.teapots(emissive_grid=(16, 32)); cam = scenes.teapots_camera(W, H)


def make():
    c = api.Context(0, use_torch_stream=False); c.upload_scene(sd); c.set_camera(cam); c.restir_init()
    gb = api.GBuffer(); gb.create(W, H, 0)
    return c, gb, torch.zeros(W * H, 3, device="cuda")


def frame(c, gb, img, f, sync_gbuffer):
    c.set_camera(cam)
    c.gbuffer_render(gb.c_struct(cam_fallback=cam), 0)
    if sync_gbuffer: c.synchronize()
    c.restir_direct(img, 0, f, gb.c_struct(cam), 3, num_spatial=5)
    gb.update(cam)


ctxs = [make(), make()]
K = 16
out = {}
for name, n, sync in (("one context, G-buffer awaited (bench_restir.py)", 1, True), ("one context, no host sync inside the frame", 1, False),
                      ("two contexts on two streams", 2, False)):
    for f in range(3):
        for c, gb, img in ctxs[:n]: frame(c, gb, img, f, sync)
    for c, _, _ in ctxs: c.synchronize()
    t0 = time.perf_counter()
    for f in range(3, 3 + K):
        for c, gb, img in ctxs[:n]: frame(c, gb, img, f, sync)
        if sync:
            ctxs[0][0].synchronize()
    for c, _, _ in ctxs: c.synchronize()
    el = time.perf_counter() - t0
    out[name] = round(el / (K * n) * 1e3, 3)
print(json.dumps({"ms_per_frame": out}))
