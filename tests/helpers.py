import numpy as np


def random_rays(n, seed, lo=(-1.2, -0.2, -1.2), hi=(1.2, 2.2, 4.5)):
    """Origins in a box around the Cornell scene, unit directions; a few axis-parallel and tiny-component
    directions so every branch of AABB::intersect is exercised."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    k = max(n // 16, 1)
    for axis in range(3):  # exactly axis-parallel
        idx = rng.choice(n, k, replace=False)
        d[idx] = 0
        d[idx, axis] = rng.choice([-1.0, 1.0], k)
    idx = rng.choice(n, k, replace=False)  # one tiny component
    d[idx, rng.integers(0, 3, k)] = 1e-7
    idx = rng.choice(n, k, replace=False)  # one exactly-zero component
    d[idx, rng.integers(0, 3, k)] = 0.0
    return np.concatenate([o, d], axis=1).astype(np.float32)


def random_segments(n, seed, lo=(-1.0, 0.0, -1.0), hi=(1.0, 2.0, 1.0)):
    rng = np.random.default_rng(seed)
    x = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    y = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    return np.concatenate([x, y], axis=1).astype(np.float32)


def bits(a):
    """View float arrays as uint32 so comparisons are bit-exact (NaN == NaN, +0 != -0)."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    ba, bb = bits(a), bits(b)
    if not np.array_equal(ba, bb):
        bad = np.argwhere(ba != bb)
        first = tuple(bad[0])
        raise AssertionError(
            f"{what}: {len(bad)} of {ba.size} floats differ bitwise; first at {first}: "
            f"{np.asarray(a, np.float32)[first]!r} vs {np.asarray(b, np.float32)[first]!r}"
        )


def restir_partition_run(gpu_ctx, sd, W, H, cams, world, tile, reuse):
    """ReSTIR frames on `world` virtual ranks (one rdh_ctx each, all on this GPU): whole-frame G-buffer, rdh_restir_direct per
    rank into its packed tiles, reservoir exchange (pack -> simulated all-gather -> unpack), rdh_untile.  Returns the frames and
    every rank's `last` reservoirs after each frame."""
    from radish_pt_amd import api

    import torch
    n = W * H
    dev = api.DevScene()
    ctxs = []
    for rank in range(world):
        c = gpu_ctx if world == 1 else api.Context(0)
        c.upload_scene(sd)
        c.set_partition(rank, world, tile)
        c.set_camera(cams[0])
        c.restir_init()
        ctxs.append(c)
    gb = api.GBuffer()
    gb.create(W, H)
    frames, resv = [], []
    tpr = ctxs[0].tiles_per_rank()
    imgs = [torch.zeros(n if world == 1 else tpr * tile * tile, 3, device="cuda") for _ in ctxs]
    for f, cam in enumerate(cams):
        dev.ctx = ctxs[0]
        gb.render(dev, cam)  # whole frame regardless of the partition
        for c, img in zip(ctxs, imgs):
            c.set_camera(cam)
            c.restir_direct(img, 0, 40 + f, gb.c_struct(cam), reuse)
        if world == 1:
            frames.append(imgs[0].cpu().numpy().copy())
        else:
            packs = []
            for c in ctxs:
                pk = torch.zeros(tpr * tile * tile, 9, device="cuda")
                c.restir_exchange_pack(pk)
                packs.append(pk)
            gathered = torch.cat(packs).contiguous()
            for c in ctxs:
                c.restir_exchange_unpack(gathered)
            frame = torch.zeros(n, 3, device="cuda")
            ctxs[0].untile(torch.cat(imgs).contiguous(), frame)
            ctxs[0].synchronize()
            frames.append(frame.cpu().numpy().copy())
        resv.append([c.restir_read(1).tobytes() for c in ctxs])
        gb.update(cam)
    for c in ctxs:
        c.restir_free()
        if c is not gpu_ctx:
            c.close()
    return frames, resv


def oracle_path_trace_mt(sd, cam, direct, indirect, iter, looper, depth, stride=1, threads=None):
    """The oracle's pathTrace on every `stride`-th pixel of the frame, spread over host threads (one oracle handle per thread,
    pixels dealt round-robin; the ctypes call releases the GIL).  Fills the sampled rows of direct / indirect in place and
    returns (pixel indices, summed work counters)."""
    import os
    import threading

    from oracle import pyoracle

    w, h = (int(v) for v in cam["resolution"])
    n = w * h
    if threads is None:
        threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    handles = [pyoracle.OracleScene(sd) for _ in range(threads)]
    ts = [threading.Thread(target=handles[t].path_trace, args=(cam, direct, indirect, iter, looper, depth),
                           kwargs={"pix": (t * stride, n, threads * stride)}) for t in range(threads)]
    for th in ts:
        th.start()
    for th in ts:
        th.join()
    stats = {}
    for hd in handles:
        for k, v in hd.stats().items():
            stats[k] = stats.get(k, 0) + v
    return np.arange(0, n, stride), stats
