#!/usr/bin/env python3
"""bench.py — Mrays/s and ms/frame of the hot path at BASELINE.json's headline configuration.

Workload (config.workload): BASELINE config 2 — Cornell box stand-in (18 444 triangles; the reference's scene assets
are absent), 1920x1080, 8 bounces, 1 spp per frame, `pathTrace` semantics.  A "step" is one frame: step s
renders Sobol row `looper = s` with `iter = 0` (the reference app resets `iteration` every frame), into
device-resident image buffers.

N = 1:  one process, one GPU, frame layout.
Frames in flight (default: one per GPU of the job, so 1 at N = 1): consecutive frames are independent (iter = 0), so with
F > 1 frame s+1 is issued on another HIP stream into its own buffers while frame s is still draining its last long paths
— with the frame cut N ways each GPU's share of one frame is too small to fill it (a frame's latency is its longest path,
not its pixel count).  Every frame is rendered completely; ms_per_step = wall time of the K frames / K.

N > 1:  launched by torch.distributed.run, one rank per GPU.  The frame is cut into 64x64 tiles, tile t → rank t % N
        (strong scaling: the frame is fixed, per-GPU work shrinks).  Each rank traces its tiles into packed tile
        buffers; the finished tiles are exchanged with ONE RCCL all-gather per image per frame (all_gather_into_tensor
        over xGMI) and re-assembled with rdh_untile — all inside the timed region.

value = (closest-hit + any-hit rays actually traced in the K timed frames, all ranks) / (max-over-ranks wall time).
Ray counts are exact device counters taken in an untimed pass over the same Sobol rows (the counters cost atomics,
so the timed pass runs without them; the rays traced are identical).

roofline: the dominant kernel is the one that traverses (k_pt_persistent by default; k_wf_trace in wavefront mode,
k_path_trace_mega in megakernel mode).
`achieved` = algorithmic bytes ÷ the hipEvent-measured duration of its launches during the timed steps, with
  B = 40·closestRays + 28·anyRays + 32·nodeVisits + 36·triTests + 64·closestHits          
(ray in 24 B + hit record 16 B / occlusion flag 4 B; 32 B per box step; 36 B per triangle test; 64 B of normals, uvs and
material id per found hit), against the 8.0 TB/s HBM3E peak.

cpu_baseline: the oracle (CPU restatement, single thread) timed on one whole frame of the same workload, rank 0, N = 1
only; its image doubles as a full-frame bit-exact parity check of the timed configuration (`parity_check`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(c):
    return (40 * c["closestRays"] + 28 * c["anyRays"] + 32 * c["nodeVisits"] + 36 * c["triTests"] + 64 * c["closestHits"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--mode", default=os.environ.get("RADISH_BENCH_MODE", "persistent"),
                    choices=["mega", "wavefront", "wavefront_sort", "persistent"])
    ap.add_argument("--scene", default="cornell", choices=["cornell", "cornell_small", "teapots", "teapots_lights", "teasets_1m"])
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=int(os.environ.get("RADISH_FRAMES_IN_FLIGHT", "0")),
                    help="frames rendered concurrently, each on its own HIP stream into its own image buffers "
                         "(0 = one per GPU of the job, at most 8)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from radish_pt_amd import api, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # Rehearsal hooks (one-GPU boxes): RADISH_FORCE_DEVICE puts every rank on one device, RADISH_DIST_BACKEND=gloo
        # replaces RCCL, which refuses two ranks on one GPU.  The driver's N>1 runs use neither.
        dev_index = int(os.environ.get("RADISH_FORCE_DEVICE", local_rank))
        backend = os.environ.get("RADISH_DIST_BACKEND", "nccl")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    W, H, depth = args.width, args.height, args.depth
    sd = {"cornell": scenes.cornell, "cornell_small": lambda: scenes.cornell(segments=16, bands=12),
          "teapots": scenes.teapots, "teapots_lights": lambda: scenes.teapots(emissive_grid=(16, 32)),
          # stand-in for BASELINE config 5's "camera and tea sets" (asset absent): the teapots scene re-tessellated to ~1.0 M tris
          "teasets_1m": lambda: scenes.teapots(segments=200, bands=156, emissive_grid=(16, 32))}[args.scene]()
    cam = scenes.cornell_camera(W, H) if args.scene.startswith("cornell") else scenes.teapots_camera(W, H)
    flags = {"mega": api.RDH_PT_MEGAKERNEL, "wavefront": api.RDH_PT_WAVEFRONT,
             "wavefront_sort": api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL,
             "persistent": api.RDH_PT_PERSISTENT}[args.mode]

    # One "slot" per frame in flight: its own context (stream, persistent-kernel workspace) and its own image buffers,
    # so consecutive frames are independent (iter = 0: each frame overwrites its images) and can overlap on the GPU —
    # the tail of frame s, where a few long paths are still running, is filled by the start of frame s+1.
    F = args.frames_in_flight if args.frames_in_flight > 0 else min(8, max(1, world))

    class Slot:
        pass

    slots = []
    for f in range(F):
        sl = Slot()
        sl.stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(sl.stream):
            sl.ctx = api.Context(dev.index)  # binds to the current torch stream = sl.stream
            sl.ctx.upload_scene(sd)
            sl.ctx.set_camera(cam)
            sl.ctx.set_partition(rank, world, args.tile)
            if world == 1:
                sl.direct = torch.zeros(W * H, 3, device=dev)
                sl.indirect = torch.zeros(W * H, 3, device=dev)
            else:
                tpr = sl.ctx.tiles_per_rank()
                shard = tpr * args.tile * args.tile
                sl.direct = torch.zeros(shard, 3, device=dev)
                sl.indirect = torch.zeros(shard, 3, device=dev)
                sl.gath_d = torch.zeros(world * shard, 3, device=dev)
                sl.gath_i = torch.zeros(world * shard, 3, device=dev)
                sl.frame_d = torch.zeros(W * H, 3, device=dev)
                sl.frame_i = torch.zeros(W * H, 3, device=dev)
        slots.append(sl)
    torch.cuda.synchronize()
    ctx = slots[0].ctx
    direct, indirect = slots[0].direct, slots[0].indirect

    def step(s, f):
        sl = slots[s % F]
        with torch.cuda.stream(sl.stream):
            sl.ctx.path_trace(sl.direct, sl.indirect, 0, s % api.SOBOL_SAMPLE_NUM, depth, f)
            if world > 1:
                dist.all_gather_into_tensor(sl.gath_d, sl.direct)
                dist.all_gather_into_tensor(sl.gath_i, sl.indirect)
                sl.ctx.untile(sl.gath_d, sl.frame_d)
                sl.ctx.untile(sl.gath_i, sl.frame_i)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    K, Wm = args.steps, args.warmup
    # untimed: warmup, then the exact work counters of the K frames that will be timed
    for s in range(Wm):
        step(s, flags)
    torch.cuda.synchronize()
    for sl in slots:
        sl.ctx.counters_reset()
    for s in range(Wm, Wm + K):
        sl = slots[s % F]
        with torch.cuda.stream(sl.stream):
            sl.ctx.path_trace(sl.direct, sl.indirect, 0, s % api.SOBOL_SAMPLE_NUM, depth, flags | api.RDH_PT_COUNT)
    torch.cuda.synchronize()
    counters = {}
    for sl in slots:
        for key, val in sl.ctx.counters().items():
            counters[key] = counters.get(key, 0) + val

    for sl in slots:
        sl.ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for s in range(Wm, Wm + K):
        step(s, flags | api.RDH_PT_PROFILE)
    barrier()
    elapsed = time.perf_counter() - t0
    trace_ms, trace_launches = 0.0, 0
    for sl in slots:
        ms, n = sl.ctx.profile_read()
        trace_ms += ms
        trace_launches += n

    rays_local = counters["closestRays"] + counters["anyRays"]
    stats = torch.tensor([elapsed, float(rays_local), float(algorithmic_bytes(counters)), trace_ms, float(trace_launches)],
                         dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        rays_total = float(sm[1])
    else:
        rays_total = float(rays_local)

    if rank == 0:
        mrays = rays_total / elapsed / 1e6
        launches = max(trace_launches, 1)
        alg_bytes = algorithmic_bytes(counters)  # rank 0's launches
        achieved = (alg_bytes / launches) / (trace_ms / launches * 1e-3) / 1e9 if trace_ms > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(pmc_path) and world == 1 and F == 1:  # the PMC figure is per launch of the one-GPU workload
            try:
                with open(pmc_path) as fh:
                    traffic = json.load(fh).get(f"{args.scene}_{args.mode}_{W}x{H}_d{depth}_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s", "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "strong",  # the frame (total work) is fixed; per-GPU work shrinks with N
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} stand-in ({sd.num_prims} tris), {W}x{H}, {depth} bounces, 1 spp/frame, "
                                   f"pathTrace ({args.mode}), tile-partitioned x{world}",
                       "rays_per_frame": rays_total / K, "mode": args.mode, "parallelism": f"tile{args.tile}x{world}",
                       "frames_in_flight": F},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": ("k_wf_trace" if flags & api.RDH_PT_WAVEFRONT else
                                    "k_pt_persistent" if flags & api.RDH_PT_PERSISTENT else "k_path_trace_mega"),
                         "launches": trace_launches, "avg_launch_ms": round(trace_ms / launches, 5),
                         "algorithmic_bytes_per_launch": alg_bytes / launches},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle

            o = pyoracle.OracleScene(sd)
            ref_d = np.zeros((W * H, 3), np.float32)
            ref_i = np.zeros((W * H, 3), np.float32)
            stride = 1  # the whole frame: ~6 s of single-thread CPU work, and a full-frame parity check for free
            full_affinity = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
            if full_affinity is not None:
                try:
                    os.sched_setaffinity(0, {sorted(full_affinity)[0]})
                except OSError:
                    pass
            tc = time.perf_counter()
            o.path_trace(cam, ref_d, ref_i, 0, Wm, depth, pix=(0, W * H, stride))
            cpu_s = time.perf_counter() - tc
            st = o.stats()
            cpu_rays = st["closestRays"] + st["anyRays"]
            out["cpu_baseline"] = {
                "value": round(cpu_rays / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
                "sample": f"oracle pathTrace on every pixel (stride {stride}) of the same {W}x{H} depth-{depth} frame "
                          f"(looper {Wm}): {cpu_rays} rays in {cpu_s:.1f} s",
            }
            # secondary figure: the same frame on all host cores of this process's share, one oracle handle per
            # thread, pixels dealt round-robin (the ctypes call releases the GIL)
            if full_affinity is not None:
                try:
                    os.sched_setaffinity(0, full_affinity)
                except OSError:
                    pass
            import threading

            nthreads = max(1, min(16, len(full_affinity) if full_affinity else (os.cpu_count() or 1)))
            if nthreads > 1:
                handles = [pyoracle.OracleScene(sd) for _ in range(nthreads)]
                td, ti = np.zeros((W * H, 3), np.float32), np.zeros((W * H, 3), np.float32)
                threads = [threading.Thread(target=handles[t].path_trace, args=(cam, td, ti, 0, Wm, depth),
                                            kwargs={"pix": (t, W * H, nthreads)}) for t in range(nthreads)]
                tc = time.perf_counter()
                for th in threads:
                    th.start()
                for th in threads:
                    th.join()
                par_s = time.perf_counter() - tc
                par_rays = sum(h.stats()["closestRays"] + h.stats()["anyRays"] for h in handles)
                out["cpu_baseline"]["all_cores"] = {"value": round(par_rays / par_s / 1e6, 4), "unit": "Mrays/s", "cores": nthreads,
                                                    "bit_equal_to_one_thread": bool(np.array_equal(td.view(np.uint32), ref_d.view(np.uint32))
                                                                                    and np.array_equal(ti.view(np.uint32), ref_i.view(np.uint32)))}
            # parity spot check on the timed configuration: the sampled pixels must equal the GPU frame bit for bit
            with torch.cuda.stream(slots[0].stream):
                ctx.path_trace(direct, indirect, 0, Wm, depth, flags)
            ctx.synchronize()
            idx = np.arange(0, W * H, stride)
            g_d, g_i = direct.cpu().numpy(), indirect.cpu().numpy()
            ok = (np.array_equal(g_d[idx].view(np.uint32), ref_d[idx].view(np.uint32))
                  and np.array_equal(g_i[idx].view(np.uint32), ref_i[idx].view(np.uint32)))
            out["parity_check"] = {"pixels": int(len(idx)), "bit_exact": bool(ok)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for sl in slots:
        sl.ctx.close()


if __name__ == "__main__":
    main()
