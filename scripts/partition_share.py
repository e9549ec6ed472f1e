#!/usr/bin/env python3
"""Experiment: a rank's share of the teapots frame (persistent kernel) with FEWER resident waves (rdh_set_occupancy_share divides the
persistent grid): do fuller, longer-lived waves shorten a small launch?  usage: partition_share.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radish_pt_amd import api
import bench
sd = bench.make_scene("teapots"); cam = bench.make_camera("teapots", 1920, 1080)
dev = torch.device("cuda", 0)
for world in (8, 4, 1):
    for share in (1, 2, 3, 4):
        ctx = api.Context(0); ctx.upload_scene(sd); ctx.set_camera(cam); ctx.set_partition(0, world, 64)
        ctx.set_occupancy_share(share)
        n = 1920 * 1080 if world == 1 else ctx.tiles_per_rank() * 64 * 64
        d = torch.zeros(n, 3, device=dev); i = torch.zeros(n, 3, device=dev)
        for s in range(4): ctx.path_trace(d, i, 0, s, 8, api.RDH_PT_PERSISTENT)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(4, 14): ctx.path_trace(d, i, 0, s, 8, api.RDH_PT_PERSISTENT)
        torch.cuda.synchronize()
        print(json.dumps({"world": world, "grid_divided_by": share, "ms": round((time.perf_counter() - t0) / 10 * 1e3, 4)}), flush=True)
        ctx.close()
