#!/usr/bin/env python3
"""bench.py — Mrays/s and ms/frame of the hot path at BASELINE.json's headline configuration.

Workload (config.workload): BASELINE config 2 — Cornell box stand-in (18 444 triangles; the reference's scene assets
are absent), 1920x1080, 8 bounces, 1 spp per frame, `pathTrace` semantics.  A "step" is one frame: step s
renders Sobol row `looper = s` with `iter = 0` (the reference app resets `iteration` every frame), into
device-resident image buffers.

N = 1:  one process, one GPU, frame layout.
N > 1:  one rank per GPU over RCCL.  `python bench.py --gpus N` with no WORLD_SIZE in the environment starts the ranks itself:
        before torch or HIP is touched it runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
        process, relays its output and exits with its code; launched by torch.distributed.run it is simply a rank.
        The frame is cut into 64x64 tiles, tile t → rank t % N (strong scaling: the frame is fixed, per-GPU work shrinks).
        Each rank traces its tiles into packed tile buffers; the finished tiles are exchanged with ONE RCCL all-gather per
        image per frame (all_gather_into_tensor over xGMI) and re-assembled with rdh_untile — all inside the timed region.

Frames in flight: `value` and `ms_per_step` are ALWAYS measured with ONE frame in flight (F = 1: frame s+1 is issued when frame
s has been enqueued on the same stream; ms_per_step is then a frame latency as well as a rate), at every N, so the scaling
curve compares like with like.  A second, labelled figure `pipelined` re-times the same K frames with F = max(3, N) frames in
flight (each on its own stream and buffers, persistent grids divided by F so the slots together fill the GPU once; on one GPU
F = 3 measured best: 2.66 ms against 2.80 at F = 2 and 3.04 at F = 4): that one is throughput only.
`--frames-in-flight F` overrides the headline's F (then `config.frames_in_flight` says so).
Frames in flight need HARDWARE queues of their own: ROCm gives a process 4 by default and deals further streams onto them round
robin, so two contexts (three streams each) can land on one queue and run one after the other (measured: F = 2 at 4.71 ms per
frame instead of 2.80).  bench.py therefore sets GPU_MAX_HW_QUEUES=16 before the HIP runtime starts, unless the caller has set it.

value = (closest-hit + any-hit rays actually traced in the K timed frames, all ranks) / (max-over-ranks wall time).
Ray counts are exact device counters taken in an untimed pass over the same Sobol rows (the counters cost atomics,
so the timed pass runs without them; the rays traced are identical).

roofline: the dominant kernel is the one that traverses (k_pt_persistent by default; k_wf_trace in wavefront mode,
k_path_trace_mega in megakernel mode).
`achieved` = algorithmic bytes ÷ the hipEvent-measured duration of its launches during the timed steps, with
  B = 40·closestRays + 28·anyRays + 32·nodeVisits + 36·triTests + 64·closestHits          
(ray in 24 B + hit record 16 B / occlusion flag 4 B; 32 B per box step; 36 B per triangle test; 64 B of normals, uvs and
material id per found hit), against the 8.0 TB/s HBM3E peak.
`roofline.traversal_only` (N = 1): north_star's 70 % target is "during BVH traversal", so the walk-only kernel
(k_walk_persistent, device/kernels_walk.h) is timed in this same run on the ray lists of the warm-up frame itself (every
closest-hit ray and every occlusion segment of frame looper = warmup, dumped once, untimed), with the algorithmic bytes of
exactly those rays.
`roofline.traffic`: HBM-side bytes per launch from PMC counters cannot be read inside the run; when profiles/hbm_traffic.json
holds the figure of a `rocprofv3 --pmc` pass of this exact workload it is carried with `traffic_source: "replayed ..."`.

cpu_baseline (rank 0, N = 1 only): §8d's denominator — the oracle's DevScene::intersect / testOcclusion (restating
src/intersections.h + src/scene.h:262-334) over a strided sample of the SAME dumped ray lists, one thread pinned to a core,
best of 3 → `value`; beside it the oracle's whole pathTrace of the frame, one thread (`whole_path_trace`) and all cores
(`all_cores`), whose image is a full-frame bit-exact parity check of the timed configuration (`parity_check`).  A parity
failure makes bench.py exit non-zero WITHOUT printing a performance line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(c):
    return (40 * c["closestRays"] + 28 * c["anyRays"] + 32 * c["nodeVisits"] + 36 * c["triTests"] + 64 * c["closestHits"])


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--mode", default=os.environ.get("RADISH_BENCH_MODE", "persistent"),
                    choices=["mega", "wavefront", "wavefront_sort", "wavefront2", "wavefront_sort2", "persistent"],
                    help="wavefront2 / wavefront_sort2: the wavefront pipeline as three sub-frames on three streams (RDH_PT_WF_SUBFRAMES)")
    ap.add_argument("--scene", default="cornell", choices=["cornell", "cornell_small", "teapots", "teapots_lights", "teasets_1m"])
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traversal-only", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=int(os.environ.get("RADISH_FRAMES_IN_FLIGHT", "1")),
                    help="frames in flight of the HEADLINE measurement (default 1 at every N)")
    ap.add_argument("--master-port", type=int, default=int(os.environ.get("RADISH_MASTER_PORT", "29533")))
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as a child process — this process has
    not imported torch nor touched HIP, and it never exec()s — relay the child's output and return its exit code."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    args = parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime starts (children of self_launch inherit it)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import numpy as np
    import torch

    from radish_pt_amd import api, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
             "wavefront2": api.RDH_PT_WAVEFRONT | api.RDH_PT_WF_SUBFRAMES,
             "wavefront_sort2": api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL | api.RDH_PT_WF_SUBFRAMES,
             "persistent": api.RDH_PT_PERSISTENT}[args.mode]
    K, Wm = args.steps, args.warmup

    class Slot:
        """One frame in flight: its own context (stream, persistent-kernel workspace) and its own image buffers, so consecutive
        frames are independent (iter = 0: each frame overwrites its images)."""

        def __init__(self, share):
            self.stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self.stream):
                self.ctx = api.Context(dev.index)  # binds to the current torch stream = self.stream
                self.ctx.upload_scene(sd)
                self.ctx.set_camera(cam)
                self.ctx.set_partition(rank, world, args.tile)
                self.ctx.set_occupancy_share(share)
                if world == 1:
                    self.direct = torch.zeros(W * H, 3, device=dev)
                    self.indirect = torch.zeros(W * H, 3, device=dev)
                else:
                    shard = self.ctx.tiles_per_rank() * args.tile * args.tile
                    self.direct = torch.zeros(shard, 3, device=dev)
                    self.indirect = torch.zeros(shard, 3, device=dev)
                    self.gath_d = torch.zeros(world * shard, 3, device=dev)
                    self.gath_i = torch.zeros(world * shard, 3, device=dev)
                    self.frame_d = torch.zeros(W * H, 3, device=dev)
                    self.frame_i = torch.zeros(W * H, 3, device=dev)

        def step(self, s, f, comm_stream=None):
            """One frame.  comm_stream None (the headline, F = 1): render, all-gather and un-tile on this slot's stream.
            With several frames in flight every slot's collectives are issued on ONE shared stream, in frame order — the same
            order on every rank, exactly as in the headline — and only the rendering overlaps."""
            with torch.cuda.stream(self.stream):
                self.ctx.path_trace(self.direct, self.indirect, 0, s % api.SOBOL_SAMPLE_NUM, depth, f)
                if world > 1 and comm_stream is None:
                    dist.all_gather_into_tensor(self.gath_d, self.direct)
                    dist.all_gather_into_tensor(self.gath_i, self.indirect)
                elif world > 1:
                    rendered = torch.cuda.Event()
                    rendered.record(self.stream)
                    comm_stream.wait_event(rendered)
                    with torch.cuda.stream(comm_stream):
                        dist.all_gather_into_tensor(self.gath_d, self.direct)
                        dist.all_gather_into_tensor(self.gath_i, self.indirect)
                        gathered = torch.cuda.Event()
                        gathered.record(comm_stream)
                    self.stream.wait_event(gathered)  # also orders this slot's NEXT render after the gather has read its tiles
                if world > 1:
                    self.ctx.untile(self.gath_d, self.frame_d)
                    self.ctx.untile(self.gath_i, self.frame_i)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(F):
        """Warm up, count, then time exactly K frames with F frames in flight.  Returns a dict of this rank's figures."""
        slots = [Slot(F) for _ in range(F)]
        comm_stream = torch.cuda.Stream(device=dev) if (F > 1 and world > 1) else None
        torch.cuda.synchronize()
        for s in range(Wm):
            slots[s % F].step(s, flags, comm_stream)
        torch.cuda.synchronize()
        for sl in slots:
            sl.ctx.counters_reset()
        for s in range(Wm, Wm + K):  # untimed: the exact work counters of the K frames that will be timed
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
            slots[s % F].step(s, flags | api.RDH_PT_PROFILE, comm_stream)
        barrier()
        elapsed = time.perf_counter() - t0
        trace_ms, trace_launches = 0.0, 0
        for sl in slots:
            ms, n = sl.ctx.profile_read()
            trace_ms += ms
            trace_launches += n
        rays_local = counters["closestRays"] + counters["anyRays"]
        if world > 1:
            st = torch.tensor([elapsed, float(rays_local)], dtype=torch.float64, device=dev)
            mx, sm = st.clone(), st.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            elapsed, rays_total = float(mx[0]), float(sm[1])
        else:
            rays_total = float(rays_local)
        return {"slots": slots, "F": F, "elapsed": elapsed, "rays_total": rays_total, "counters": counters, "trace_ms": trace_ms,
                "trace_launches": trace_launches}

    F = max(1, args.frames_in_flight)
    m = measure(F)
    pipelined = None
    if not args.no_pipelined and F == 1:
        Fp = min(8, max(3, world))
        mp_ = measure(Fp)
        pipelined = {"frames_in_flight": Fp, "value": round(mp_["rays_total"] / mp_["elapsed"] / 1e6, 3), "unit": "Mrays/s",
                     "ms_per_step": round(mp_["elapsed"] / K * 1e3, 4),
                     "note": "throughput with several frames in flight per GPU (each on its own stream, persistent grids divided "
                             "by F); not a frame latency, not the headline"}
        for sl in mp_["slots"]:
            sl.ctx.close()
    slots = m["slots"]
    ctx = slots[0].ctx
    direct, indirect = slots[0].direct, slots[0].indirect
    elapsed, rays_total, counters = m["elapsed"], m["rays_total"], m["counters"]
    trace_ms, trace_launches = m["trace_ms"], m["trace_launches"]

    rc = 0
    if rank == 0:
        mrays = rays_total / elapsed / 1e6
        launches = max(trace_launches, 1)
        alg_bytes = algorithmic_bytes(counters)  # rank 0's launches
        # per-launch figures only mean something when launches do not overlap (F = 1)
        # (… and the two-pipeline wavefront modes are not profiled per launch: their launches overlap by design)
        achieved = (alg_bytes / launches) / (trace_ms / launches * 1e-3) / 1e9 if (trace_ms > 0 and F == 1) else None
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(pmc_path) and world == 1 and F == 1:  # the PMC figure is per launch of the one-GPU workload
            try:
                with open(pmc_path) as fh:
                    pj = json.load(fh)
                traffic = pj.get(f"{args.scene}_{args.mode}_{W}x{H}_d{depth}_bytes_per_launch")
                if traffic is not None:
                    traffic_source = ("replayed from profiles/hbm_traffic.json — " + str(pj.get("source", "rocprofv3 --pmc pass of this workload"))
                                      + "; PMC counters cannot be read inside this run")
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
                       "frames_in_flight": F,
                       "value_is": f"F = {F} frame(s) in flight: every frame is enqueued after the previous one on one stream per GPU"},
            "roofline": {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "bound_note": "algorithmic bytes per second against the HBM3E peak; the scene is L2 / Infinity-Cache resident, "
                                       "so the binding units are the L1 request rate and VALU issue (DESIGN.md §6-7)",
                         "kernel": ("k_wf_trace" if flags & api.RDH_PT_WAVEFRONT else
                                    "k_pt_persistent" if flags & api.RDH_PT_PERSISTENT else "k_path_trace_mega"),
                         "launches": trace_launches, "avg_launch_ms": round(trace_ms / launches, 5),
                         "algorithmic_bytes_per_launch": alg_bytes / launches},
        }
        if pipelined is not None:
            out["pipelined"] = pipelined

        closest = segs = None
        if world == 1 and not (args.no_traversal_only and args.no_cpu_baseline):
            with torch.cuda.stream(slots[0].stream):
                closest, segs = ctx.dump_rays(Wm, depth)  # the warm-up frame's own rays, untimed
        if world == 1 and not args.no_traversal_only:
            # ---- traversal only: k_walk_persistent over the frame's own ray lists ----
            with torch.cuda.stream(slots[0].stream):
                hits = torch.zeros(closest.shape[0], 4, dtype=torch.int32, device=dev)
                occ = torch.zeros(max(segs.shape[0], 1), dtype=torch.int32, device=dev)
                ctx.counters_reset()
                ctx.trace_closest(closest, hits, api.RDH_PT_PERSISTENT | api.RDH_PT_COUNT)
                if segs.shape[0]:
                    ctx.trace_occluded(segs, occ, api.RDH_PT_PERSISTENT | api.RDH_PT_COUNT)
                wc = ctx.counters()
                t_ms, reps = 0.0, 5
                for r in range(reps + 1):
                    ctx.trace_closest(closest, hits, api.RDH_PT_PERSISTENT)
                    a = ctx.last_kernel_ms()
                    b = 0.0
                    if segs.shape[0]:
                        ctx.trace_occluded(segs, occ, api.RDH_PT_PERSISTENT)
                        b = ctx.last_kernel_ms()
                    if r > 0:  # first repetition warms the instruction cache
                        t_ms += a + b
                t_ms /= reps
            wbytes = algorithmic_bytes(wc)
            wach = wbytes / (t_ms * 1e-3) / 1e9
            out["roofline"]["traversal_only"] = {
                "kernel": "k_walk_persistent (closest-hit list, then any-hit list)", "rays": int(closest.shape[0] + segs.shape[0]),
                "ms": round(t_ms, 4), "algorithmic_bytes": wbytes, "achieved": round(wach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(wach / HBM_PEAK_GBS, 4), "mrays_per_s": round((closest.shape[0] + segs.shape[0]) / (t_ms * 1e-3) / 1e6, 1),
                "box_steps_per_s_G": round(wc["nodeVisits"] / (t_ms * 1e-3) / 1e9, 1),
                "sample": f"every ray of frame looper={Wm} of this workload (dumped untimed), hipEvents on the context's stream, mean of {reps}",
            }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle

            o = pyoracle.OracleScene(sd)
            full_affinity = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None

            def pin(cpus):
                if full_affinity is not None:
                    try:
                        os.sched_setaffinity(0, cpus)
                    except OSError:
                        pass

            pin({sorted(full_affinity)[0]} if full_affinity else None)
            # ---- §8d denominator: traversal only, one pinned thread, best of 3, on a strided sample of the frame's ray lists ----
            stride_r = max(1, int(closest.shape[0] + segs.shape[0]) // 1_500_000)
            h_closest = closest[::stride_r].contiguous().cpu().numpy()
            h_segs = segs[::stride_r].contiguous().cpu().numpy()
            best = None
            for _ in range(3):
                tc = time.perf_counter()
                o.trace_closest(h_closest)
                if len(h_segs):
                    o.trace_occluded(h_segs)
                dt = time.perf_counter() - tc
                best = dt if best is None else min(best, dt)
            n_sample = len(h_closest) + len(h_segs)
            out["cpu_baseline"] = {
                "value": round(n_sample / best / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
                "sample": f"oracle DevScene::intersect / testOcclusion (traversal only) over every {stride_r}-th ray of the frame's own "
                          f"ray lists (looper {Wm}): {n_sample} rays, one pinned thread, best of 3 = {best:.2f} s",
            }
            # ---- the oracle's whole pathTrace of the same frame (shading included), one thread: also the parity reference ----
            o.reset_stats()
            ref_d = np.zeros((W * H, 3), np.float32)
            ref_i = np.zeros((W * H, 3), np.float32)
            tc = time.perf_counter()
            o.path_trace(cam, ref_d, ref_i, 0, Wm, depth)
            cpu_s = time.perf_counter() - tc
            st = o.stats()
            cpu_rays = st["closestRays"] + st["anyRays"]
            out["cpu_baseline"]["whole_path_trace"] = {
                "value": round(cpu_rays / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                "sample": f"oracle pathTrace on every pixel of the same {W}x{H} depth-{depth} frame (looper {Wm}): {cpu_rays} rays in {cpu_s:.1f} s"}
            pin(full_affinity)
            import threading

            nthreads = max(1, min(16, len(full_affinity) if full_affinity else (os.cpu_count() or 1)))
            all_equal = True
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
                all_equal = bool(np.array_equal(td.view(np.uint32), ref_d.view(np.uint32)) and np.array_equal(ti.view(np.uint32), ref_i.view(np.uint32)))
                out["cpu_baseline"]["all_cores"] = {"value": round(par_rays / par_s / 1e6, 4), "unit": "Mrays/s", "cores": nthreads,
                                                    "bit_equal_to_one_thread": all_equal}
            # parity check on the timed configuration: the whole GPU frame must equal the oracle's bit for bit
            with torch.cuda.stream(slots[0].stream):
                ctx.path_trace(direct, indirect, 0, Wm, depth, flags)
            ctx.synchronize()
            g_d, g_i = direct.cpu().numpy(), indirect.cpu().numpy()
            bad = np.argwhere((g_d.view(np.uint32) != ref_d.view(np.uint32)) | (g_i.view(np.uint32) != ref_i.view(np.uint32)))
            out["parity_check"] = {"pixels": int(W * H), "bit_exact": bool(len(bad) == 0)}
            if len(bad) or not all_equal:
                # a performance number for wrong pixels is worthless: report and fail, print no metric line
                p = int(bad[0][0]) if len(bad) else -1
                print(f"bench.py: PARITY FAILURE — {len(bad)} floats differ from the oracle; first at pixel {p} "
                      f"(x={p % W}, y={p // W}): gpu direct {g_d[p] if p >= 0 else None} indirect {g_i[p] if p >= 0 else None} vs oracle "
                      f"{ref_d[p] if p >= 0 else None} {ref_i[p] if p >= 0 else None}; all-cores oracle equal to one thread: {all_equal}",
                      file=sys.stderr, flush=True)
                rc = 3
        elif world > 1:
            out["cpu_baseline"] = None
            # Parity of the PARTITIONED frame (no oracle at N > 1: it needs minutes per frame): rank 0 renders the last timed frame
            # once more alone — one rank, whole frame, same Sobol row, same kernel structure — and compares it bit for bit with what
            # the all-gathers and rdh_untile left on this rank.  (The one-rank frame itself is checked against the oracle at N = 1.)
            s_last = Wm + K - 1
            last = slots[s_last % F]
            torch.cuda.synchronize()
            with torch.cuda.stream(last.stream):
                solo = api.Context(dev.index)
                solo.upload_scene(sd)
                solo.set_camera(cam)
                solo.set_partition(0, 1, args.tile)
                solo_d = torch.zeros(W * H, 3, device=dev)
                solo_i = torch.zeros(W * H, 3, device=dev)
                solo.path_trace(solo_d, solo_i, 0, s_last % api.SOBOL_SAMPLE_NUM, depth, flags)
                solo.synchronize()
            diff = (solo_d.view(torch.int32) != last.frame_d.view(torch.int32)) | (solo_i.view(torch.int32) != last.frame_i.view(torch.int32))
            n_bad = int(diff.sum())
            out["parity_check"] = {"pixels": int(W * H), "bit_exact": n_bad == 0,
                                   "against": "the same frame rendered by rank 0 alone (one rank, whole frame); that frame is checked "
                                              "against the CPU oracle by the N = 1 run"}
            solo.close()
            if n_bad:
                p = int(torch.nonzero(diff.any(dim=1))[0])
                print(f"bench.py: PARITY FAILURE at {world} ranks — {n_bad} floats of the gathered frame differ from the one-rank frame; "
                      f"first at pixel {p} (x={p % W}, y={p // W}, tile {(p // W) // args.tile * ((W + args.tile - 1) // args.tile) + (p % W) // args.tile})",
                      file=sys.stderr, flush=True)
                rc = 3
        elif args.no_cpu_baseline:
            out["cpu_baseline"] = None
        if rc == 0:
            print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for sl in slots:
        sl.ctx.close()
    sys.exit(rc)


if __name__ == "__main__":
    main()
