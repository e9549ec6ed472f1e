"""The library's own collectives (RCCL bound at run time inside libradish_hip.so: rdh_comm_*, rdh_allgather_tiles,
rdh_path_trace_gathered, rdh_restir_direct_gathered, rdh_restir_exchange, rdh_gbuffer_exchange) and the partitioned G-buffer.

A one-GPU box can run RCCL with ONE rank only (RCCL refuses two ranks on one device), so here the collectives run at
world = 1 — the same code path as at world = N: packed tile buffers, ncclAllGather on the context's stream, un-tile — and the
N-rank data movement is covered by virtual ranks (one rdh_ctx per rank on this GPU, the all-gather replaced by a concatenation,
which is what an all-gather delivers).  The 2-rank RCCL test at the end runs only where two GPUs are visible.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import assert_bit_equal

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torch():
    import torch

    return torch


def test_rccl_one_rank_path_trace_gathered(cornell_small):
    """rdh_comm_init with world = 1, then the multi-GPU pathTrace entry (whole-frame images in and out): three accumulated
    frames through packed tiles + ncclAllGather + un-tile equal the plain single-GPU call bit for bit; frame size not a
    multiple of the tile size."""
    from radish_pt_amd import api, scenes

    torch = _torch()
    W, H, depth = 200, 120, 4
    cam = scenes.cornell_camera(W, H)
    ref_ctx, ctx = api.Context(0), api.Context(0)
    try:
        for c in (ref_ctx, ctx):
            c.upload_scene(cornell_small)
            c.set_camera(cam)
        ctx.set_partition(0, 1, 32)
        ctx.comm_init(api.Context.comm_unique_id(), 0, 1)
        rd, ri = torch.zeros(W * H, 3, device="cuda"), torch.zeros(W * H, 3, device="cuda")
        gd, gi = torch.zeros(W * H, 3, device="cuda"), torch.zeros(W * H, 3, device="cuda")
        for it, flags in enumerate((api.RDH_PT_PERSISTENT, api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL, api.RDH_PT_MEGAKERNEL)):
            ref_ctx.path_trace(rd, ri, it, 30 + it, depth, api.RDH_PT_PERSISTENT)
            ctx.path_trace_gathered(gd, gi, it, 30 + it, depth, flags)
        ctx.synchronize()
        ref_ctx.synchronize()
        assert_bit_equal(gd.cpu().numpy(), rd.cpu().numpy(), "gathered direct")
        assert_bit_equal(gi.cpu().numpy(), ri.cpu().numpy(), "gathered indirect")
        assert float(ri.max()) > 0
        # rdh_allgather_tiles on its own: a packed render of the one rank's tiles -> frame
        ctx.set_partition(0, 1, 32)
        tpr = ctx.tiles_per_rank()
        # with world == 1 the plain entry renders in frame layout; pack it by hand through the partition helper
        from radish_pt_amd import partition

        order = torch.from_numpy(partition.untile_indices(W, H, 1, 32)).cuda()  # frame pixel -> index into the gathered buffer
        packed = torch.zeros(tpr * 32 * 32, 3, device="cuda")
        packed[order.long()] = rd
        out = torch.zeros(W * H, 3, device="cuda")
        ctx.allgather_tiles(packed.contiguous(), out)
        ctx.synchronize()
        assert_bit_equal(out.cpu().numpy(), rd.cpu().numpy(), "rdh_allgather_tiles")
        ctx.comm_destroy()
        with pytest.raises(api.RadishError):
            ctx.path_trace_gathered(gd, gi, 0, 0, depth)  # no communicator
    finally:
        ctx.close()
        ref_ctx.close()


def test_one_process_group_init_and_gathered_all(cornell_small):
    """The ONE-process host model (the reference is one process with one frame loop, main.cpp:163-202):
    rdh_comm_init_all creates the communicators of n contexts inside one RCCL group and rdh_path_trace_gathered_all renders and
    gathers for all of them from one thread.  One GPU here, so n = 1 (two contexts on one device must be refused — RCCL wants a
    device per rank); the n = 2 form runs in test_one_process_two_gpus where two devices exist."""
    from radish_pt_amd import api, scenes

    torch = _torch()
    W, H, depth = 200, 120, 4
    cam = scenes.cornell_camera(W, H)
    ref_ctx, ctx, ctx_b = api.Context(0), api.Context(0), api.Context(0)
    try:
        for c in (ref_ctx, ctx, ctx_b):
            c.upload_scene(cornell_small)
            c.set_camera(cam)
        ctx.set_partition(0, 1, 32)
        with pytest.raises(api.RadishError):
            api.Context.comm_init_all([ctx, ctx_b])  # same device twice
        rd, ri = torch.zeros(W * H, 3, device="cuda"), torch.zeros(W * H, 3, device="cuda")
        gd, gi = torch.zeros(W * H, 3, device="cuda"), torch.zeros(W * H, 3, device="cuda")
        with pytest.raises(api.RadishError):
            api.Context.path_trace_gathered_all([ctx], [gd], [gi], 0, 0, depth)  # no communicator yet
        api.Context.comm_init_all([ctx])
        assert (ctx.rank, ctx.world) == (0, 1)
        for it, flags in enumerate((api.RDH_PT_PERSISTENT, api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL)):
            ref_ctx.path_trace(rd, ri, it, 40 + it, depth, api.RDH_PT_PERSISTENT)
            api.Context.path_trace_gathered_all([ctx], [gd], [gi], it, 40 + it, depth, flags)
        ctx.synchronize()
        ref_ctx.synchronize()
        assert_bit_equal(gd.cpu().numpy(), rd.cpu().numpy(), "gathered_all direct")
        assert_bit_equal(gi.cpu().numpy(), ri.cpu().numpy(), "gathered_all indirect")
        assert float(ri.max()) > 0
    finally:
        for c in (ctx, ctx_b, ref_ctx):
            c.close()


@pytest.mark.skipif("__import__('torch').cuda.device_count() < 2", reason="needs two GPUs")
def test_one_process_two_gpus(cornell_small):
    """n = 2 contexts of ONE process on two devices: grouped communicator creation, grouped all-gathers; both devices end up with
    the whole frame, equal to the single-GPU render."""
    from radish_pt_amd import api, scenes

    torch = _torch()
    W, H, depth = 200, 120, 4
    cam = scenes.cornell_camera(W, H)
    ref_ctx = api.Context(0)
    ctxs = [api.Context(0, use_torch_stream=False), api.Context(1, use_torch_stream=False)]
    try:
        for c in [ref_ctx] + ctxs:
            c.upload_scene(cornell_small)
            c.set_camera(cam)
        for c in ctxs:
            c.set_partition(0, 1, 32)
        api.Context.comm_init_all(ctxs)
        rd, ri = torch.zeros(W * H, 3, device="cuda:0"), torch.zeros(W * H, 3, device="cuda:0")
        gd = [torch.zeros(W * H, 3, device=f"cuda:{k}") for k in range(2)]
        gi = [torch.zeros(W * H, 3, device=f"cuda:{k}") for k in range(2)]
        torch.cuda.synchronize(0)
        torch.cuda.synchronize(1)
        for it in range(2):
            ref_ctx.path_trace(rd, ri, it, 50 + it, depth, api.RDH_PT_PERSISTENT)
            api.Context.path_trace_gathered_all(ctxs, gd, gi, it, 50 + it, depth, api.RDH_PT_PERSISTENT)
        for c in [ref_ctx] + ctxs:
            c.synchronize()
        for k in range(2):
            assert_bit_equal(gd[k].cpu().numpy(), rd.cpu().numpy(), f"device {k} direct")
            assert_bit_equal(gi[k].cpu().numpy(), ri.cpu().numpy(), f"device {k} indirect")
    finally:
        for c in ctxs + [ref_ctx]:
            c.close()


@pytest.mark.parametrize("mode", ["overlap", "render_stream", "overlap_no_host_sync", "one_process_all"])
def test_rccl_one_rank_restir_gathered(mode):
    """ReSTIRDirect for N GPUs at N = 1: partitioned G-buffer + rdh_gbuffer_exchange, rdh_restir_direct_gathered (image
    all-gather + reservoir exchange inside), frames with a moving camera: G-buffer planes, images and reservoirs equal the plain
    single-GPU calls.  overlap (default): the exchanges run on the context's communication stream beside the rendering;
    render_stream: rdh_comm_set_overlap(0); overlap_no_host_sync: four frames enqueued back to back with no host synchronisation
    in between, so that only the events order the reservoir / G-buffer exchanges against the next frame's kernels;
    one_process_all: the same through rdh_comm_init_all + the *_all entries (one RCCL group per collective)."""
    from radish_pt_amd import api, hostlib, scenes

    torch = _torch()
    sd = scenes.teapots(segments=12, bands=8, grid=2, emissive_grid=(4, 8))
    W, H = 150, 90
    n = W * H
    nframes = 4 if mode == "overlap_no_host_sync" else 3
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.08 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0) for f in range(nframes)]
    ref_ctx, ctx = api.Context(0), api.Context(0)
    try:
        for c in (ref_ctx, ctx):
            c.upload_scene(sd)
            c.set_camera(cams[0])
            c.restir_init()
        ctx.set_partition(0, 1, 32)
        if mode == "one_process_all":
            api.Context.comm_init_all([ctx])
        else:
            ctx.comm_init(api.Context.comm_unique_id(), 0, 1)
        if mode == "render_stream":
            ctx.comm_set_overlap(False)
        gb_ref, gb = api.GBuffer(), api.GBuffer()
        gb_ref.create(W, H)
        gb.create(W, H)
        img_ref, img = torch.zeros(n, 3, device="cuda"), torch.zeros(n, 3, device="cuda")
        for f, cam in enumerate(cams):
            ref_ctx.set_camera(cam)
            ctx.set_camera(cam)
            ref_ctx.gbuffer_render(gb_ref.c_struct(cam_fallback=cam), 0)
            ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), api.RDH_PT_PARTITION_GBUFFER)
            if mode == "one_process_all":
                api.Context.gbuffer_exchange_all([ctx], [gb.c_struct(cam_fallback=cam)])
            else:
                ctx.gbuffer_exchange(gb.c_struct(cam_fallback=cam))
            if mode != "overlap_no_host_sync":
                ctx.synchronize()
                ref_ctx.synchronize()
                k = gb.frameIdx
                for a, b, plane in ((gb.albedo, gb_ref.albedo, "albedo"), (gb.normal[k], gb_ref.normal[k], "normal"),
                                    (gb.depth[k], gb_ref.depth[k], "depth"), (gb.primId[k], gb_ref.primId[k], "primId"),
                                    (gb.motion, gb_ref.motion, "motion")):
                    assert np.array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32)), f"frame {f}: {plane}"
            ref_ctx.restir_direct(img_ref, 0, 40 + f, gb_ref.c_struct(cam), 3)
            if mode == "one_process_all":
                api.Context.restir_direct_gathered_all([ctx], [img], 0, 40 + f, [gb.c_struct(cam)], 3)
            else:
                ctx.restir_direct_gathered(img, 0, 40 + f, gb.c_struct(cam), 3)
            if mode != "overlap_no_host_sync":
                ctx.synchronize()
                assert_bit_equal(img.cpu().numpy(), img_ref.cpu().numpy(), f"ReSTIR gathered, frame {f}")
                assert ctx.restir_read(1).tobytes() == ref_ctx.restir_read(1).tobytes(), f"reservoirs, frame {f}"
            gb_ref.update(cam)
            gb.update(cam)
        ctx.synchronize()
        ref_ctx.synchronize()
        assert_bit_equal(img.cpu().numpy(), img_ref.cpu().numpy(), "ReSTIR gathered, last frame")
        assert ctx.restir_read(1).tobytes() == ref_ctx.restir_read(1).tobytes(), "reservoirs, last frame"
        assert float(img_ref.max()) > 0
    finally:
        ctx.close()
        ref_ctx.close()


@pytest.mark.parametrize("axis_aligned", [False, True])
def test_gbuffer_partition_exchange_virtual_ranks(gpu_ctx, axis_aligned):
    """The G-buffer without replicated work: every (virtual) rank renders the records of ITS tiles only
    (RDH_PT_PARTITION_GBUFFER), packs them (36 B per pixel), the packs are concatenated as an all-gather would, every rank
    unpacks: each rank's planes equal the single-GPU G-buffer, and the ranks together trace exactly W*H primary rays.  The
    axis-aligned camera makes a pixel row and column literal-class rays (more than the workgroup-per-ray cap)."""
    from radish_pt_amd import api, hostlib, scenes

    torch = _torch()
    sd = scenes.teapots(segments=12, bands=8, grid=2, emissive_grid=(4, 8))
    W, H = 301, 145
    if axis_aligned:
        cams = [hostlib.make_camera(W, H, eye=(0.0, 1.0 + 0.1 * f, 9.0), rotation=(-90.0, 0.0, 0.0), fovy=19.0) for f in range(2)]
    else:
        cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.08 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0) for f in range(2)]
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    ref = api.GBuffer()
    ref.create(W, H)
    ref_planes, ref_counts = [], []
    for cam in cams:
        gpu_ctx.set_camera(cam)
        gpu_ctx.counters_reset()
        gpu_ctx.gbuffer_render(ref.c_struct(cam_fallback=cam), api.RDH_PT_COUNT)
        gpu_ctx.synchronize()
        k = ref.frameIdx
        ref_planes.append([t.cpu().numpy().copy() for t in (ref.albedo, ref.normal[k], ref.depth[k], ref.primId[k], ref.motion)])
        ref_counts.append(gpu_ctx.counters())
        ref.update(cam)
    for world, tile in ((2, 64), (3, 32), (8, 16)):
        ctxs, gbs = [], []
        for rank in range(world):
            c = api.Context(0)
            c.upload_scene(sd)
            c.set_camera(cams[0])
            c.set_partition(rank, world, tile)
            g = api.GBuffer()
            g.create(W, H)
            ctxs.append(c)
            gbs.append(g)
        tpr = ctxs[0].tiles_per_rank()
        for f, cam in enumerate(cams):
            packs, total = [], {}
            for c, g in zip(ctxs, gbs):
                c.set_camera(cam)
                c.counters_reset()
                c.gbuffer_render(g.c_struct(cam_fallback=cam), api.RDH_PT_PARTITION_GBUFFER | api.RDH_PT_COUNT)
                pk = torch.zeros(tpr * tile * tile, 9, device="cuda")
                c.gbuffer_exchange_pack(g.c_struct(cam_fallback=cam), pk)
                c.synchronize()
                packs.append(pk)
                for key, v in c.counters().items():
                    total[key] = total.get(key, 0) + v
            gathered = torch.cat(packs).contiguous()
            assert total == ref_counts[f], f"world={world} frame {f}: the ranks' work is not the frame's work"
            assert total["closestRays"] == W * H
            for r, (c, g) in enumerate(zip(ctxs, gbs)):
                c.gbuffer_exchange_unpack(g.c_struct(cam_fallback=cam), gathered)
                c.synchronize()
                k = g.frameIdx
                for t, want, plane in zip((g.albedo, g.normal[k], g.depth[k], g.primId[k], g.motion), ref_planes[f],
                                          ("albedo", "normal", "depth", "primId", "motion")):
                    assert np.array_equal(t.cpu().numpy().view(np.uint32), want.view(np.uint32)), f"world={world} rank {r} frame {f}: {plane}"
                g.update(cam)
        for c in ctxs:
            c.close()
    gpu_ctx.set_partition(0, 1, 64)


def test_stream_rebind_between_launches(cornell_small):
    """rdh_set_stream while launches are in flight: the persistent kernel's block reservation counter and cost /
    order double buffers are per context, so a rebind must drain the old stream first.  Frames rendered while hopping between
    two streams every call equal frames rendered on one stream."""
    import ctypes as C

    from radish_pt_amd import api, scenes

    torch = _torch()
    W, H, depth = 320, 200, 6
    cam = scenes.cornell_camera(W, H)
    ctx = api.Context(0)
    try:
        ctx.upload_scene(cornell_small)
        ctx.set_camera(cam)
        ref = []
        d, i = torch.zeros(W * H, 3, device="cuda"), torch.zeros(W * H, 3, device="cuda")
        for f in range(6):
            ctx.path_trace(d, i, 0, f, depth)
            ctx.synchronize()
            ref.append((d.cpu().numpy().copy(), i.cpu().numpy().copy()))
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [(torch.zeros(W * H, 3, device="cuda"), torch.zeros(W * H, 3, device="cuda")) for _ in range(6)]
        for f in range(6):
            ctx.check(api.lib().rdh_set_stream(ctx.h, C.c_void_p(streams[f & 1].cuda_stream)))
            ctx.path_trace(outs[f][0], outs[f][1], 0, f, depth)
        torch.cuda.synchronize()
        for f in range(6):
            assert_bit_equal(outs[f][0].cpu().numpy(), ref[f][0], f"frame {f} direct after stream hops")
            assert_bit_equal(outs[f][1].cpu().numpy(), ref[f][1], f"frame {f} indirect after stream hops")
    finally:
        ctx.close()


def test_api_rejects_wrong_tensors(gpu_ctx, cornell_small):
    """The Python mirror checks what the C ABI cannot: dtype, device, contiguity and size of every image argument."""
    from radish_pt_amd import api, scenes

    torch = _torch()
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(cornell_small)
    gpu_ctx.set_camera(scenes.cornell_camera(32, 16))
    good = torch.zeros(32 * 16, 3, device="cuda")
    for bad in (torch.zeros(32 * 16 - 1, 3, device="cuda"), torch.zeros(32 * 16, 3, device="cuda", dtype=torch.float64),
                torch.zeros(32 * 16, 3), torch.zeros(32 * 16, 6, device="cuda")[:, ::2]):
        with pytest.raises(api.RadishError):
            gpu_ctx.path_trace(bad, good, 0, 0, 2)
        with pytest.raises(api.RadishError):
            gpu_ctx.path_trace(good, bad, 0, 0, 2)
        with pytest.raises(api.RadishError):
            gpu_ctx.path_trace_direct(bad, 0, 0)
    gpu_ctx.path_trace(good, good.clone(), 0, 0, 2)
    gpu_ctx.synchronize()


def test_two_ranks_over_rccl():
    """Two real ranks over RCCL (needs two visible GPUs; skipped on a one-GPU box): scripts/multi_rank_check.py drives bench.py's
    step sequence (torch all_gather_into_tensor + rdh_untile) AND the library's own rdh_path_trace_gathered /
    rdh_restir_direct_gathered, and compares every frame bit for bit with the one-GPU frame."""
    torch = _torch()
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "scripts", "multi_rank_check.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "multi_rank_check ok" in r.stdout
