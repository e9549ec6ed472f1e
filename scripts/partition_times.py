#!/usr/bin/env python3
"""What ONE rank of an N-GPU job spends on its share of the frame, measured on one GPU: rdh_set_partition(rank r, world N) and a
timed pathTrace into packed tiles (no collective), for N = 1, 2, 4, 8 and both fast structures.  This is the compute part of a
rank's step in `bench.py --gpus N`; it bounds the strong-scaling curve before any interconnect enters.
usage: python scripts/partition_times.py [scene] [W H]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radish_pt_amd import api
import bench

scene = sys.argv[1] if len(sys.argv) > 1 else "teapots"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
depth, K = 8, 10
sd = bench.make_scene(scene)
cam = bench.make_camera(scene, W, H)
dev = torch.device("cuda", 0)
out = {"scene": scene, "W": W, "H": H, "depth": depth, "rows": []}
for mode in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("wavefront_sort2", "wavefront_sort", "persistent")):
    flags = bench.mode_flags(api, mode)
    for world in (1, 2, 4, 8):
        worst = None
        for rank in sorted({0, world // 2, world - 1}):
            ctx = api.Context(0)
            ctx.upload_scene(sd); ctx.set_camera(cam); ctx.set_partition(rank, world, 64)
            n = W * H if world == 1 else ctx.tiles_per_rank() * 64 * 64
            d = torch.zeros(n, 3, device=dev); i = torch.zeros(n, 3, device=dev)
            for s in range(4):
                ctx.path_trace(d, i, 0, s, depth, flags)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(4, 4 + K):
                ctx.path_trace(d, i, 0, s, depth, flags)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / K * 1e3
            worst = ms if worst is None else max(worst, ms)
            ctx.close()
        out["rows"].append({"mode": mode, "world": world, "ms_rank_share_worst_of_3_ranks": round(worst, 4)})
        print(json.dumps(out["rows"][-1]), flush=True)
base = {r["mode"]: r["ms_rank_share_worst_of_3_ranks"] for r in out["rows"] if r["world"] == 1}
for r in out["rows"]:
    r["speedup_vs_same_mode_1"] = round(base[r["mode"]] / r["ms_rank_share_worst_of_3_ranks"], 3)
print(json.dumps(out))
