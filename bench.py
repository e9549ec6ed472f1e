#!/usr/bin/env python3
"""bench.py — Mrays/s and ms/frame of the hot path on the scene BOTH north_star targets name: BASELINE config 3.

Default workload (`config.workload`): teapots stand-in (100 364 triangles; the reference's assets are absent),
1920x1080, 8 bounces, 1 spp per frame, `pathTrace` semantics, in the structure config 3 names — the wavefront pipeline with
stream compaction and the material sort (`--mode wavefront_sort2`: three sub-frame pipelines on three streams).
It replaces the `printf("PT runtime ...")` of /root/reference/src/pathtrace.cu:352-377.  A "step" is one frame: step s renders
Sobol row `looper = s` with `iter = 0` (the reference app resets `iteration` every frame) into device-resident
images.  The SAME default runs at every N, so the driver's BENCH line and the N = 1 point of its SCALE curve are one workload.

N = 1:  one process, one GPU, frame layout.
N > 1:  one rank per GPU over RCCL.  `python bench.py --gpus N` with no WORLD_SIZE in the environment starts the ranks itself:
        before torch or HIP is touched it runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
        process, relays its output and exits with its code; launched by torch.distributed.run it is simply a rank.
        The frame is cut into 64x64 tiles, tile t → rank t % N (strong scaling: the frame is fixed, per-GPU work shrinks).
        The headline runs the PRODUCT's collective: `rdh_path_trace_gathered` (library RCCL — ncclAllGather on the context's
        stream, then the un-tile kernel: what a C++ Radish host calls), whole-frame images on every rank.
        `collective_cross_check` re-times the same frames with torch.distributed's all_gather_into_tensor (same RCCL underneath).
        If the library's communicator cannot be created (e.g. the gloo rehearsal on a one-GPU box) the torch path IS the
        headline and `config.collective` says so with the reason — never silently.

Frames in flight: `value` / `ms_per_step` are ALWAYS one frame in flight (F = 1), at every N.  `pipelined` re-times the same K
frames with F = max(3, N) frames in flight (throughput only, labelled).  GPU_MAX_HW_QUEUES=16 is set before HIP starts
(contexts on separate streams need hardware queues of their own).

value = (closest-hit + any-hit rays actually traced in the K timed frames, all ranks) / (max-over-ranks wall time).
Ray counts are exact device counters taken in an untimed pass over the same Sobol rows.

roofline: B = 40·closestRays + 28·anyRays + 32·nodeVisits + 36·triTests + 64·closestHits algorithmic bytes.
  * persistent / mega / one-pipeline wavefront: `achieved` = B per launch ÷ hipEvent-measured launch duration (RDH_PT_PROFILE).
  * wavefront with sub-frames (the default): the three pipelines' launches overlap BY DESIGN, so no single launch time means
    anything; `achieved` = B per FRAME ÷ the hipEvent span of the frame (one event pair, before the fork → after the join, on the
    context's stream), dominant kernel named k_wf_trace, `launches` = frames, `span` = "frame".
  * `traversal_only` (N = 1): the walk-only kernel (k_walk_pair: sibling-pair walks) timed on the warm-up frame's own
    ray lists; next to its §8(d) fraction it carries `l1_request_frac` = box steps/s ÷ the measured ceiling of the record layout
    (four dwordx4 requests per lane for the two children of a node = two L1 requests per visit, fully divergent lanes: 306 G box
    steps/s, scripts/micro/gather_modes mode 6, profiles/r01_c_micro_gather_modes.txt) — the honest ceiling: these scenes live
    in L2 / Infinity Cache, HBM-side traffic is a small fraction of B (`traffic`, replayed from the committed PMC passes).
cpu_baseline (rank 0, N = 1): §8d's denominator — the oracle's intersect / testOcclusion over a strided sample of the SAME ray
lists, one pinned thread, best of 3; beside it the oracle's whole pathTrace (all cores: the full frame = the parity reference of
`parity_check`; one thread: every 8th pixel).  A parity failure exits non-zero WITHOUT printing a performance line.
`configs` (N = 1): short sub-records of BASELINE configs 2, 4 and 5's scene — {ms_per_step, mrays_s, frac, parity_sample_ok},
each checked against the oracle on >= 20 000 pixels.

`--workload restir`: ReSTIR DI (config 4 at N = 1; config 5 — the 1-M-triangle scene at 4K, 128-px ownership tiles, partitioned
G-buffer — at N > 1) through the same self-launch, with the library's gathered entries.
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
L1_BOX_STEP_CEILING_G = 306.0  # box steps/s the vector L1s serve at 2 requests per lane and visit, fully divergent lanes, chip-wide
                               # (gather_modes: 53.9 ns per wave-step for 2 x dwordx4 = 304 G; 107.1 ns for the 4 x dwordx4 of a pair = 306 G)
RESTIR_PX_BYTES = 2384  # per ReSTIR pixel, both passes, excluding its two rays


def algorithmic_bytes(c):
    return (40 * c["closestRays"] + 28 * c["anyRays"] + 32 * c["nodeVisits"] + 36 * c["triTests"] + 64 * c["closestHits"])


MODES = ["mega", "wavefront", "wavefront_sort", "wavefront2", "wavefront_sort2", "persistent"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("RADISH_BENCH_WORKLOAD", "pathtrace"), choices=["pathtrace", "restir"])
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--mode", default=os.environ.get("RADISH_BENCH_MODE", "auto"), choices=["auto"] + MODES,
                    help="auto = config 3's named structure, wavefront_sort2 (wavefront + compaction + material sort, three sub-frames)")
    ap.add_argument("--scene", default=None, choices=["cornell", "cornell_small", "teapots", "teapots_lights", "teasets_1m"])
    ap.add_argument("--tile", type=int, default=None, help="ownership tile edge (default 64; restir at N > 1: 128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traversal-only", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the sub-records of configs 2, 4 and 5")
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


def make_scene(name):
    from radish_pt_amd import scenes

    return {"cornell": scenes.cornell, "cornell_small": lambda: scenes.cornell(segments=16, bands=12),
            "teapots": scenes.teapots, "teapots_lights": lambda: scenes.teapots(emissive_grid=(16, 32)),
            # stand-in for BASELINE config 5's "camera and tea sets" (asset absent): the teapots scene re-tessellated to ~1.0 M tris
            "teasets_1m": lambda: scenes.teapots(segments=200, bands=156, emissive_grid=(16, 32))}[name]()


def make_camera(name, W, H):
    from radish_pt_amd import scenes

    return scenes.cornell_camera(W, H) if name.startswith("cornell") else scenes.teapots_camera(W, H)


def mode_flags(api, mode):
    return {"mega": api.RDH_PT_MEGAKERNEL, "wavefront": api.RDH_PT_WAVEFRONT,
            "wavefront_sort": api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL,
            "wavefront2": api.RDH_PT_WAVEFRONT | api.RDH_PT_WF_SUBFRAMES,
            "wavefront_sort2": api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL | api.RDH_PT_WF_SUBFRAMES,
            "persistent": api.RDH_PT_PERSISTENT}[mode]


def dominant_kernel(api, flags):
    return ("k_wf_trace" if flags & api.RDH_PT_WAVEFRONT else "k_pt_persistent" if flags & api.RDH_PT_PERSISTENT else "k_path_trace_mega")


def bit_equal(a, b):
    import numpy as np

    return bool(np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)))


# ---------------------------------------------------------------------------------------------------------------------------
# sub-records of the other BASELINE configs (N = 1 only): short, each with an oracle sample of >= 20 000 pixels
# ---------------------------------------------------------------------------------------------------------------------------
def sub_path_trace(torch, np, api, dev, scene_name, W, H, depth, mode, K, warm, stride, label):
    """configs 2 / 5: pathTrace of `scene_name` in structure `mode`; parity on every `stride`-th pixel of the first timed frame."""
    from oracle import pyoracle

    sd = make_scene(scene_name)
    cam = make_camera(scene_name, W, H)
    flags = mode_flags(api, mode)
    ctx = api.Context(dev.index)
    ctx.upload_scene(sd)
    ctx.set_camera(cam)
    d = torch.zeros(W * H, 3, device=dev)
    i = torch.zeros(W * H, 3, device=dev)
    for s in range(warm):
        ctx.path_trace(d, i, 0, s, depth, flags)
    ctx.counters_reset()
    for s in range(warm, warm + K):
        ctx.path_trace(d, i, 0, s, depth, flags | api.RDH_PT_COUNT)
    c = ctx.counters()
    ctx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(warm, warm + K):
        ctx.path_trace(d, i, 0, s, depth, flags | api.RDH_PT_PROFILE)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    k_ms, k_n = ctx.profile_read()
    rays = c["closestRays"] + c["anyRays"]
    alg = algorithmic_bytes(c)
    frac = alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None
    # parity sample: the oracle on every stride-th pixel of frame looper = warm
    ctx.path_trace(d, i, 0, warm, depth, flags)
    ctx.synchronize()
    g_d, g_i = d.cpu().numpy(), i.cpu().numpy()
    r_d, r_i = np.zeros((W * H, 3), np.float32), np.zeros((W * H, 3), np.float32)
    n_t = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    import threading

    hs = [pyoracle.OracleScene(sd) for _ in range(n_t)]
    th = [threading.Thread(target=hs[t].path_trace, args=(cam, r_d, r_i, 0, warm, depth), kwargs={"pix": (t * stride, W * H, n_t * stride)})
          for t in range(n_t)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sel = np.arange(0, W * H, stride)
    ok = bit_equal(g_d[sel], r_d[sel]) and bit_equal(g_i[sel], r_i[sel])
    ctx.close()
    return {"workload": label, "mode": mode, "ms_per_step": round(el / K * 1e3, 4), "mrays_s": round(rays / el / 1e6, 1),
            "frac": None if frac is None else round(frac, 4), "frac_is": "algorithmic bytes / hipEvent time of %s (%d launches) / 8 TB/s" % (dominant_kernel(api, flags), k_n),
            "rays_per_frame": rays / K, "parity_sample_ok": ok, "parity_sample_pixels": int(len(sel))}


def sub_restir(torch, np, api, dev, W, H, K, num_spatial=5):
    """config 4: teapots + 1 024 emissive triangles, ReSTIR DI M = 32, temporal + spatial, split pass 1; a frame = G-buffer + ReSTIR,
    timed twice: frames enqueued back to back, and blocking after each call as in the reference's runCuda (main.cpp:183-200).  Parity: the same scene and settings at 208x112 (23 296
    px), two frames (temporal reuse active in the second), image bit-equal to the oracle's."""
    from oracle import pyoracle
    from radish_pt_amd import layouts as L

    sd = make_scene("teapots_lights")
    cam = make_camera("teapots_lights", W, H)
    ctx = api.Context(dev.index)
    ctx.upload_scene(sd)
    ctx.set_camera(cam)
    gb = api.GBuffer()
    gb.create(W, H, dev.index)
    img = torch.zeros(W * H, 3, device=dev)
    ctx.restir_init()

    def frame(f, flags=0, block=True):
        ctx.set_camera(cam)
        ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), 0)
        t_g = t_r = 0.0
        if block:
            ctx.synchronize()
            t_g = ctx.last_kernel_ms()
        ctx.restir_direct(img, 0, f, gb.c_struct(cam), 3, num_spatial=num_spatial, flags=flags)
        if block:
            ctx.synchronize()
            t_r = ctx.last_kernel_ms()
        gb.update(cam)
        return t_g, t_r

    for f in range(3):
        frame(f)
    ctx.counters_reset()
    frame(3, api.RDH_PT_COUNT)
    c = ctx.counters()
    # (1) the host blocks after each of the two calls, as the reference's runCuda does with ERRORCHECK on (main.cpp:183-200,
    # cudaUtil.h:10-18): two host round trips per frame, and the per-pass hipEvent times
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tg = tr = 0.0
    for f in range(4, 4 + K):
        a, b = frame(f)
        tg += a
        tr += b
    torch.cuda.synchronize()
    el_block = time.perf_counter() - t0
    # (2) K frames enqueued back to back on the context's stream, one barrier at each end — how the headline's frames are timed
    t0 = time.perf_counter()
    for f in range(4 + K, 4 + 2 * K):
        frame(f, block=False)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rays = c["closestRays"] + c["anyRays"]
    ray_bytes = algorithmic_bytes(c)
    ctx.restir_free()
    ctx.close()
    # parity at 208x112
    w2, h2 = 208, 112
    cam2 = make_camera("teapots_lights", w2, h2)
    ctx2 = api.Context(dev.index)
    ctx2.upload_scene(sd)
    ctx2.set_camera(cam2)
    o = pyoracle.OracleScene(sd)
    gbo = pyoracle.GBufferHost(w2, h2)
    gbg = api.GBuffer()
    gbg.create(w2, h2, dev.index)
    res = [np.zeros(w2 * h2, L.RESERVOIR_DTYPE) for _ in range(3)]
    img2 = torch.zeros(w2 * h2, 3, device=dev)
    ctx2.restir_init()
    ok = True
    for f in range(2):
        o.gbuffer_render(cam2, gbo)
        ref = np.zeros((w2 * h2, 3), np.float32)
        o.restir_direct(cam2, ref, 0, f, res[0], res[1], res[2], gbo, f == 0, 3, 1, num_spatial, 32)
        res[0], res[1] = res[1], res[0]  # restir.cu:221
        gbo.update(cam2)
        ctx2.gbuffer_render(gbg.c_struct(cam_fallback=cam2), 0)
        ctx2.restir_direct(img2, 0, f, gbg.c_struct(cam2), 3, num_spatial=num_spatial)
        ctx2.synchronize()
        gbg.update(cam2)
        ok = ok and bit_equal(img2.cpu().numpy(), ref)
    ctx2.restir_free()
    ctx2.close()
    k_s = tr / K * 1e-3
    return {"workload": f"teapots + 1024 emissive tris ({sd.num_prims} tris), {W}x{H}, ReSTIR DI M=32, temporal + {num_spatial} spatial, split pass 1",
            "ms_per_step": round(el / K * 1e3, 4), "ms_per_step_host_blocking": round(el_block / K * 1e3, 4),
            "timing": "ms_per_step: K frames (G-buffer + ReSTIRDirect) enqueued back to back, one barrier at each end, like the headline's "
                      "frames; ms_per_step_host_blocking: the host waits after each of the two calls (the reference's ERRORCHECK build, "
                      "cudaUtil.h:10-18) — two host round trips per frame; the per-pass kernel times come from that loop",
            "ms_gbuffer_kernels": round(tg / K, 4), "ms_restir_kernels": round(tr / K, 4),
            "mrays_s": round(rays / (el / K) / 1e6, 1), "rays_per_frame": rays,
            "frac": round((ray_bytes + RESTIR_PX_BYTES * W * H) / k_s / 1e9 / HBM_PEAK_GBS, 4),
            "frac_rays_only": round(ray_bytes / k_s / 1e9 / HBM_PEAK_GBS, 4),
            "frac_is": "SURVEY 8d: (ray bytes + 2 384 B per pixel) / hipEvent time of the ReSTIR kernels / 8 TB/s; the 1 920 B/px of light-table "
                       "reads come from LDS in this design, so frac_rays_only (the two walks' bytes alone over the same time) is the physical figure",
            "parity_sample_ok": ok, "parity_sample_pixels": w2 * h2 * 2,
            "parity_sample": f"same scene and settings at {w2}x{h2}, frames 0 and 1, image bit-equal to the oracle"}


# ---------------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime starts (children of self_launch inherit it)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import numpy as np
    import torch

    from radish_pt_amd import api

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    dist = None
    backend = None
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

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def agree(ok):
        """True only if every rank says True (the ranks must take the same branch around collectives)."""
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t[0]))

    def share_unique_id():
        """rank 0's ncclUniqueId for the library's communicator, handed to every rank over torch.distributed's store"""
        box = [api.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    if args.workload == "restir":
        rc = run_restir(args, torch, np, api, dist, dev, world, rank, backend, barrier, agree, share_unique_id)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(rc)

    W, H, depth = args.width or 1920, args.height or 1080, args.depth
    scene_name = args.scene or "teapots"
    tile = args.tile or 64
    mode = args.mode
    mode_choice = None
    if mode == "auto" and world <= 2:
        mode = "wavefront_sort2"
        mode_choice = ("auto, N <= 2 -> wavefront_sort2: BASELINE config 3's named structure (material-sorted wavefront + stream compaction; three "
                       "sub-frame pipelines, sibling-pair walks); on this scene it is also the fastest structure but for its own unsorted form: "
                       "7.88 ms against 7.77 unsorted, 8.66 one pipeline, 9.34 persistent (profiles/r03_m_*, DESIGN §6).  It is what the "
                       "library's own RDH_PT_AUTO picks for this tree and this share of the frame")
    elif mode == "auto":
        mode = "persistent"
        mode_choice = ("auto, N >= 4 -> persistent (the library's RDH_PT_AUTO rule: a big tree AND at least half a 1080p frame -> wavefront, else "
                       "persistent): a rank's share of the frame is small, and the wavefront pipeline ends nine stages per frame on their longest "
                       "ray where the persistent kernel ends once on its longest path — one rank's share of this frame measured on one GPU "
                       "(scripts/partition_times.py, profiles/r03_o_partition_times_teapots.txt): 5.45 / 3.37 / 2.60 ms persistent against "
                       "5.16 / 4.03 / 3.56 ms wavefront_sort2 at 2 / 4 / 8 ranks (N = 1: 9.37 against 7.91).  Same pixels, bit for bit")
    sd = make_scene(scene_name)
    cam = make_camera(scene_name, W, H)
    flags = flags_main = mode_flags(api, mode)
    subframes = bool(flags & api.RDH_PT_WF_SUBFRAMES)
    K, Wm = args.steps, args.warmup

    # ---- the library's own communicator (headline collective at N > 1) ----
    lib_comm_reason = None

    class Slot:
        """One frame in flight: its own context (stream, persistent-kernel workspace) and its own image buffers, so consecutive
        frames are independent (iter = 0: each frame overwrites its images)."""

        def __init__(self, share, lib_comm):
            nonlocal lib_comm_reason
            self.stream = torch.cuda.Stream(device=dev)
            self.lib_comm = False
            with torch.cuda.stream(self.stream):
                self.ctx = api.Context(dev.index)  # binds to the current torch stream = self.stream
                self.ctx.upload_scene(sd)
                self.ctx.set_camera(cam)
                self.ctx.set_partition(rank, world, tile)
                self.ctx.set_occupancy_share(share)
                if world == 1:
                    self.direct = torch.zeros(W * H, 3, device=dev)
                    self.indirect = torch.zeros(W * H, 3, device=dev)
                else:
                    shard = self.ctx.tiles_per_rank() * tile * tile
                    self.direct = torch.zeros(shard, 3, device=dev)
                    self.indirect = torch.zeros(shard, 3, device=dev)
                    self.gath_d = torch.zeros(world * shard, 3, device=dev)
                    self.gath_i = torch.zeros(world * shard, 3, device=dev)
                    self.frame_d = torch.zeros(W * H, 3, device=dev)
                    self.frame_i = torch.zeros(W * H, 3, device=dev)
            if world > 1 and lib_comm:
                ok, why = True, None
                try:
                    uid = share_unique_id()
                    with torch.cuda.stream(self.stream):
                        self.ctx.comm_init(uid, rank, world)
                except Exception as e:  # RCCL refuses two ranks on one device (the gloo rehearsal), or is not loadable
                    ok, why = False, f"{type(e).__name__}: {e}"
                if agree(ok):
                    self.lib_comm = True
                else:
                    lib_comm_reason = why or "another rank could not create the library's communicator"
                    if ok:
                        self.ctx.comm_destroy()
                        self.ctx.set_partition(rank, world, tile)

        def step(self, s, f, comm_stream=None, use_lib=True):
            """One frame.  Library collective (the headline): rdh_path_trace_gathered — pack, render this rank's tiles, two
            ncclAllGather on the context's stream, un-tile.  torch collective: render into packed tiles, two all_gather_into_tensor,
            rdh_untile; with several frames in flight every slot's torch collectives are issued on ONE shared stream, in frame
            order — the same order on every rank — and only the rendering overlaps."""
            looper = s % api.SOBOL_SAMPLE_NUM
            with torch.cuda.stream(self.stream):
                if world > 1 and self.lib_comm and use_lib and comm_stream is None:
                    self.ctx.path_trace_gathered(self.frame_d, self.frame_i, 0, looper, depth, f)
                    return
                self.ctx.path_trace(self.direct, self.indirect, 0, looper, depth, f)
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

    def measure(F, lib_comm, slots=None, use_lib=True, flags=None):
        """Warm up, count, then time exactly K frames with F frames in flight.  Returns a dict of this rank's figures."""
        flags = flags_main if flags is None else flags
        if slots is None:
            slots = [Slot(F, lib_comm) for _ in range(F)]
        comm_stream = torch.cuda.Stream(device=dev) if (F > 1 and world > 1) else None
        torch.cuda.synchronize()
        for s in range(Wm):
            slots[s % F].step(s, flags, comm_stream, use_lib)
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
            slots[s % F].step(s, flags | api.RDH_PT_PROFILE, comm_stream, use_lib)
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
    m = measure(F, lib_comm=(F == 1))
    lib_headline = world > 1 and all(sl.lib_comm for sl in m["slots"])
    cross = None
    if lib_headline:  # the same frames through torch.distributed's collectives, on the same contexts: a cross-check of the headline
        mc = measure(F, False, slots=m["slots"], use_lib=False)
        cross = {"collective": "torch.distributed all_gather_into_tensor + rdh_untile", "value": round(mc["rays_total"] / mc["elapsed"] / 1e6, 3),
                 "unit": "Mrays/s", "ms_per_step": round(mc["elapsed"] / K * 1e3, 4)}
    pipelined = None
    if not args.no_pipelined and F == 1:
        Fp = 3  # the same at every N, so that this figure has a like-for-like scaling curve of its own (3 measured best on one GPU)
        # which structure: with a big share of the frame per rank, ONE wavefront pipeline per frame (three frames = three pipelines on
        # three streams, like the headline's three sub-frames, but each stage three times as large: 6.7 ms per frame against 7.85,
        # profiles/r03_t_*); with a small share, the persistent kernel (as the headline)
        p_mode = "wavefront_sort" if world <= 2 else "persistent"
        mp_ = measure(Fp, False, flags=mode_flags(api, p_mode))
        pipelined = {"frames_in_flight": Fp, "mode": p_mode, "value": round(mp_["rays_total"] / mp_["elapsed"] / 1e6, 3), "unit": "Mrays/s",
                     "ms_per_step": round(mp_["elapsed"] / K * 1e3, 4),
                     "note": "throughput with 3 frames in flight per GPU at every N (each on its own stream(s), persistent grids divided by 3"
                             + ("; torch.distributed collectives on one shared stream" if world > 1 else "") + "): the tails of one frame's "
                             "launches — a launch lasts at least as long as its longest path or ray, which does not shrink with the work it holds "
                             "(DESIGN §7, §8) — are filled by the other frames; not a frame latency, not the headline"}
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
        alg_bytes = algorithmic_bytes(counters)  # rank 0's launches
        # per-launch (or, with sub-frames, per-frame-span) figures only mean something when FRAMES do not overlap (F = 1)
        have = trace_ms > 0 and trace_launches > 0 and F == 1
        achieved = alg_bytes / (trace_ms * 1e-3) / 1e9 if have else None
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(pmc_path) and world == 1 and F == 1:  # the PMC figure is per launch / per frame of the one-GPU workload
            try:
                with open(pmc_path) as fh:
                    pj = json.load(fh)
                traffic = pj.get(f"{scene_name}_{mode}_{W}x{H}_d{depth}_bytes_per_launch")
                if traffic is not None:
                    traffic_source = ("replayed from profiles/hbm_traffic.json — " + str(pj.get(f"{scene_name}_{mode}_source", pj.get("source", "rocprofv3 --pmc pass of this workload")))
                                      + "; PMC counters cannot be read inside this run")
            except Exception:
                traffic = None
        kern = dominant_kernel(api, flags)
        roof = {"bound": "hbm", "binds": "l1-request/latency",
                "achieved": None if achieved is None else round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "bound_note": "the contract's figure: algorithmic bytes per second against the HBM3E peak.  The scene is L2 / Infinity-Cache "
                              "resident (traffic << algorithmic bytes), so what physically binds is the vector L1 (request rate and miss "
                              "handling) together with VALU issue: see traversal_only.l1_request_frac (DESIGN.md §6-7)",
                "kernel": kern}
        if subframes:
            roof.update({"span": "frame", "launches": trace_launches, "avg_launch_ms": round(trace_ms / max(trace_launches, 1), 5),
                         "algorithmic_bytes_per_launch": alg_bytes / max(trace_launches, 1) if have else None,
                         "span_note": "three sub-frame pipelines on three streams: their launches overlap by design, so the timed unit is the FRAME "
                                      "(one hipEvent pair on the context's stream, before the fork -> after the join); `launches` counts frames, "
                                      "`algorithmic_bytes_per_launch` is per frame; rocprofv3's per-kernel table of the same command is in profiles/"})
        else:
            roof.update({"span": "launch", "launches": trace_launches, "avg_launch_ms": round(trace_ms / max(trace_launches, 1), 5),
                         "algorithmic_bytes_per_launch": alg_bytes / trace_launches if have else None})
        out = {
            "metric": "Mrays/s", "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "strong",  # the frame (total work) is fixed; per-GPU work shrinks with N
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{scene_name} stand-in ({sd.num_prims} tris), {W}x{H}, {depth} bounces, 1 spp/frame, "
                                   f"pathTrace ({mode}), tile-partitioned x{world}",
                       "baseline_config": 3 if scene_name == "teapots" else None,
                       "rays_per_frame": rays_total / K, "mode": mode, "mode_choice": mode_choice, "parallelism": f"tile{tile}x{world}",
                       "frames_in_flight": F,
                       "collective": (None if world == 1 else
                                      "library RCCL: rdh_path_trace_gathered (ncclAllGather on the context's stream + k_untile)" if lib_headline else
                                      f"torch.distributed all_gather_into_tensor + rdh_untile — the library's communicator was not used: {lib_comm_reason or 'frames in flight > 1'}"),
                       "value_is": f"F = {F} frame(s) in flight: every frame is enqueued after the previous one on one stream per GPU"},
            "roofline": roof,
        }
        if cross is not None:
            out["collective_cross_check"] = cross
        if pipelined is not None:
            out["pipelined"] = pipelined

        closest = segs = None
        if world == 1 and not (args.no_traversal_only and args.no_cpu_baseline):
            with torch.cuda.stream(slots[0].stream):
                closest, segs = ctx.dump_rays(Wm, depth)  # the warm-up frame's own rays, untimed
        if world == 1 and not args.no_traversal_only:
            # ---- traversal only: the walk-only kernel over the frame's own ray lists ----
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
            steps_g = wc["nodeVisits"] / (t_ms * 1e-3) / 1e9
            out["roofline"]["traversal_only"] = {
                "kernel": "k_walk_pair (closest-hit list, then any-hit list)", "rays": int(closest.shape[0] + segs.shape[0]),
                "ms": round(t_ms, 4), "algorithmic_bytes": wbytes, "achieved": round(wach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(wach / HBM_PEAK_GBS, 4), "mrays_per_s": round((closest.shape[0] + segs.shape[0]) / (t_ms * 1e-3) / 1e6, 1),
                "box_steps_per_s_G": round(steps_g, 1),
                "l1_request_frac": round(steps_g / L1_BOX_STEP_CEILING_G, 4), "l1_request_ceiling_G": L1_BOX_STEP_CEILING_G,
                "l1_note": "box steps/s against what the 256 vector L1s can serve for this record layout with fully divergent lanes (the two "
                           "children of a node are one 64-byte record = four dwordx4 requests per lane, i.e. two requests per visit; 107 ns per "
                           "wave-step and CU: 306 G box steps/s, scripts/micro/gather_modes mode 6).  Lanes of a wave that meet at the top of the "
                           "tree share requests, so this fraction can pass 1 (the 1-M-triangle scene: 1.09); the §8(d) byte figure has no "
                           "ceiling at all while the nodes are served from cache",
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
            # ---- the oracle's whole pathTrace of every 8th pixel of the same frame (shading included), one pinned thread ----
            o.reset_stats()
            one_d = np.zeros((W * H, 3), np.float32)
            one_i = np.zeros((W * H, 3), np.float32)
            one_stride = 8
            tc = time.perf_counter()
            o.path_trace(cam, one_d, one_i, 0, Wm, depth, pix=(0, W * H, one_stride))
            cpu_s = time.perf_counter() - tc
            st = o.stats()
            cpu_rays = st["closestRays"] + st["anyRays"]
            out["cpu_baseline"]["whole_path_trace"] = {
                "value": round(cpu_rays / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                "sample": f"oracle pathTrace on every {one_stride}-th pixel of the same {W}x{H} depth-{depth} frame (looper {Wm}): {cpu_rays} rays in {cpu_s:.1f} s"}
            pin(full_affinity)
            # ---- the whole frame on all cores: the parity reference ----
            import threading

            nthreads = max(1, min(16, len(full_affinity) if full_affinity else (os.cpu_count() or 1)))
            handles = [pyoracle.OracleScene(sd) for _ in range(nthreads)]
            ref_d, ref_i = np.zeros((W * H, 3), np.float32), np.zeros((W * H, 3), np.float32)
            threads = [threading.Thread(target=handles[t].path_trace, args=(cam, ref_d, ref_i, 0, Wm, depth),
                                        kwargs={"pix": (t, W * H, nthreads)}) for t in range(nthreads)]
            tc = time.perf_counter()
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            par_s = time.perf_counter() - tc
            par_rays = sum(h.stats()["closestRays"] + h.stats()["anyRays"] for h in handles)
            sel = np.arange(0, W * H, one_stride)
            all_equal = bit_equal(one_d[sel], ref_d[sel]) and bit_equal(one_i[sel], ref_i[sel])
            out["cpu_baseline"]["all_cores"] = {"value": round(par_rays / par_s / 1e6, 4), "unit": "Mrays/s", "cores": nthreads,
                                                "sample": f"the whole frame: {par_rays} rays in {par_s:.1f} s",
                                                "bit_equal_to_one_thread": all_equal}
            # parity check on the timed configuration: the whole GPU frame must equal the oracle's bit for bit
            with torch.cuda.stream(slots[0].stream):
                ctx.path_trace(direct, indirect, 0, Wm, depth, flags)
            ctx.synchronize()
            g_d, g_i = direct.cpu().numpy(), indirect.cpu().numpy()
            bad = np.argwhere((g_d.view(np.uint32) != ref_d.view(np.uint32)) | (g_i.view(np.uint32) != ref_i.view(np.uint32)))
            out["parity_check"] = {"pixels": int(W * H), "bit_exact": bool(len(bad) == 0), "against": "the CPU oracle's frame (all cores; one thread agrees on every 8th pixel)"}
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
                solo.set_partition(0, 1, tile)
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
                      f"first at pixel {p} (x={p % W}, y={p // W}, tile {(p // W) // tile * ((W + tile - 1) // tile) + (p % W) // tile})",
                      file=sys.stderr, flush=True)
                rc = 3
        elif args.no_cpu_baseline:
            out["cpu_baseline"] = None

        # ---- sub-records of the other BASELINE configs (N = 1) ----
        if world == 1 and not args.no_configs and rc == 0:
            for sl in slots:
                sl.ctx.close()
            slots = []
            del direct, indirect, closest, segs
            torch.cuda.empty_cache()
            cfgs = {}
            try:
                cfgs["2"] = sub_path_trace(torch, np, api, dev, "cornell", 1920, 1080, 8, "persistent", 10, 3, 100,
                                           "BASELINE config 2: Cornell stand-in (18 444 tris), 1920x1080, 8 bounces, one persistent launch per frame")
                cfgs["4"] = sub_restir(torch, np, api, dev, 1920, 1080, 8)
                cfgs["5_scene_one_gpu"] = sub_path_trace(torch, np, api, dev, "teasets_1m", 3840, 2160, 8, "wavefront2", 3, 3, 400,
                                                         "BASELINE config 5's scene on ONE GPU: teapots re-tessellated to 999 436 tris, 3840x2160, 8 bounces, pathTrace")
            except Exception as e:  # a sub-record must not cost the headline; say what happened
                cfgs["error"] = f"{type(e).__name__}: {e}"
            out["configs"] = cfgs
            if any(isinstance(v, dict) and v.get("parity_sample_ok") is False for v in cfgs.values()):
                print("bench.py: PARITY FAILURE in a config sub-record: " + json.dumps(cfgs), file=sys.stderr, flush=True)
                rc = 3
        if rc == 0:
            print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for sl in slots:
        sl.ctx.close()
    sys.exit(rc)


# ---------------------------------------------------------------------------------------------------------------------------
# --workload restir: ReSTIR DI through the library's gathered entries (config 4 at N = 1; config 5 at N > 1)
# ---------------------------------------------------------------------------------------------------------------------------
def run_restir(args, torch, np, api, dist, dev, world, rank, backend, barrier, agree, share_unique_id):
    multi = world > 1
    scene_name = args.scene or ("teasets_1m" if multi else "teapots_lights")
    W = args.width or (3840 if multi else 1920)
    H = args.height or (2160 if multi else 1080)
    # ReSTIR's pass 1 over-computes an 8-px apron around a rank's tiles (spatial radius 5): (tile + 16)^2 / tile^2 pixels — +56 % at
    # 64-px tiles, +27 % at 128; ownership granularity for ReSTIR is therefore 128 px (pathTrace keeps 64)
    tile = args.tile or (128 if multi else 64)
    K, Wm = args.steps, max(args.warmup, 2)
    sd = make_scene(scene_name)
    cam = make_camera(scene_name, W, H)
    ctx = api.Context(dev.index)
    ctx.upload_scene(sd)
    ctx.set_camera(cam)
    ctx.set_partition(rank, world, tile)
    lib_comm, why = False, None
    if multi:
        ok = True
        try:
            ctx.comm_init(share_unique_id(), rank, world)
        except Exception as e:
            ok, why = False, f"{type(e).__name__}: {e}"
        lib_comm = agree(ok)
        if not lib_comm and ok:
            ctx.comm_destroy()
            ctx.set_partition(rank, world, tile)
    gb = api.GBuffer()
    gb.create(W, H, dev.index)
    frame_img = torch.zeros(W * H, 3, device=dev)
    n_local = W * H if not multi else ctx.tiles_per_rank() * tile * tile
    img = torch.zeros(n_local, 3, device=dev)
    packed9 = torch.zeros(n_local, 9, device=dev)
    ctx.restir_init()

    def gather(t):
        if backend == "gloo":  # rehearsal: gloo moves host memory
            h = t.cpu()
            o = torch.empty(world * h.shape[0], *h.shape[1:], dtype=h.dtype)
            dist.all_gather_into_tensor(o, h)
            return o.to(dev)
        o = torch.empty(world * t.shape[0], *t.shape[1:], dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(o, t)
        return o

    def frame(f, flags=0):
        """runCuda's sequence (main.cpp:183-200): G-buffer, ReSTIRDirect, G-buffer update — on N ranks with the exchanges."""
        ctx.set_camera(cam)
        g = gb.c_struct(cam_fallback=cam)
        if not multi:
            ctx.gbuffer_render(g, 0)
            ctx.restir_direct(frame_img, 0, f, gb.c_struct(cam), 3, flags=flags)
        elif lib_comm:
            ctx.gbuffer_render(g, api.RDH_PT_PARTITION_GBUFFER)
            ctx.gbuffer_exchange(g)
            ctx.restir_direct_gathered(frame_img, 0, f, gb.c_struct(cam), 3, flags=flags)
        else:
            ctx.gbuffer_render(g, api.RDH_PT_PARTITION_GBUFFER)
            ctx.gbuffer_exchange_pack(g, packed9)
            ctx.synchronize()
            ctx.gbuffer_exchange_unpack(g, gather(packed9))
            ctx.restir_direct(img, 0, f, gb.c_struct(cam), 3, flags=flags)
            ctx.synchronize()
            ctx.untile(gather(img), frame_img)
            ctx.restir_exchange_pack(packed9)
            ctx.synchronize()
            ctx.restir_exchange_unpack(gather(packed9))
        if multi and not lib_comm:  # torch's collectives run on torch's stream: the host orders them against the context's stream.
            ctx.synchronize()       # Otherwise frames are stream-ordered (N = 1 and the library's exchanges alike: one curve), so that
            #                         the reservoir gather runs beside the next frame's G-buffer pass and pass 1
        gb.update(cam)

    for f in range(Wm):
        frame(f)
    ctx.counters_reset()
    frame(Wm, api.RDH_PT_COUNT)
    c = ctx.counters()
    barrier()
    t0 = time.perf_counter()
    for f in range(Wm + 1, Wm + 1 + K):
        frame(f)
    barrier()
    el = time.perf_counter() - t0
    rays_local = float(c["closestRays"] + c["anyRays"])
    if multi:
        st = torch.tensor([el, rays_local], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        mx, sm = st.clone(), st.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        el, rays = float(mx[0]), float(sm[1])
    else:
        rays = rays_local
    rc = 0
    parity = None
    if multi:  # the partitioned frame against one rank rendering the same sequence alone (bit-equal image)
        solo = api.Context(dev.index)
        solo.upload_scene(sd)
        solo.set_camera(cam)
        gbs = api.GBuffer()
        gbs.create(W, H, dev.index)
        simg = torch.zeros(W * H, 3, device=dev)
        solo.restir_init()
        for f in range(Wm + 1 + K):
            solo.set_camera(cam)
            solo.gbuffer_render(gbs.c_struct(cam_fallback=cam), 0)
            solo.restir_direct(simg, 0, f, gbs.c_struct(cam), 3)
            solo.synchronize()
            gbs.update(cam)
        n_bad = int((simg.view(torch.int32) != frame_img.view(torch.int32)).sum())
        parity = {"pixels": W * H, "bit_exact": n_bad == 0, "against": f"the same {Wm + 1 + K}-frame sequence rendered by this rank alone (whole frame)"}
        solo.restir_free()
        solo.close()
        if n_bad:
            print(f"bench.py: PARITY FAILURE (restir, {world} ranks): {n_bad} floats differ from the one-rank frame", file=sys.stderr, flush=True)
            rc = 3
    if rank == 0 and rc == 0:
        out = {"metric": "Mrays/s", "value": round(rays / (el / K) / 1e6, 3), "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": Wm,
               "ms_per_step": round(el / K * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": f"{scene_name} stand-in ({sd.num_prims} tris, {sd.num_lights} emissive), {W}x{H}, ReSTIR DI M=32, temporal "
                                      f"(clamp 20) + 5 spatial, G-buffer + split pass 1 + pass 2 per frame, tile-partitioned x{world}",
                          "baseline_config": 5 if multi else 4, "rays_per_frame": rays, "parallelism": f"tile{tile}x{world}",
                          "apron": None if not multi else f"pass 1 over-computes an 8-px apron: +{((tile + 16) ** 2 / tile ** 2 - 1) * 100:.0f} % pixels at {tile}-px tiles",
                          "collective": None if not multi else ("library RCCL: rdh_gbuffer_exchange + rdh_restir_direct_gathered" if lib_comm else
                                                                f"torch.distributed (library communicator not used: {why})")},
               "parity_check": parity, "cpu_baseline": None,
               "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                            "note": "per-kernel figures of this workload: `configs.4` of the default bench line and profiles/"}}
        print(json.dumps(out), flush=True)
    ctx.restir_free()
    ctx.close()
    return rc


if __name__ == "__main__":
    main()
