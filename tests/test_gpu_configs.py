"""The five BASELINE.json configurations, each at its own size, under `pytest -m gpu`.

Config 2 (Cornell 1080p, 8 bounces) is `test_gpu_parity.py::test_1080p_variants_agree` plus the full-frame `parity_check` of
bench.py; config 4 at full size is `test_gpu_parity.py::test_restir_1080p_partition_and_determinism`.  This file adds the
three the round-1 review found untested — 1 (400x400, depth 4, 16 spp), 3 (teapots, 1080p, wavefront + compaction + material
sort) and 5 (~1 M triangles, 4K, pathTrace and ReSTIR DI on an 8-way tile split) — and an oracle check of config 4's scene.

Bar: bit-exact against the CPU oracle wherever the oracle finishes in seconds (whole frames at 400x400 and below, strided
samples of >= 20 000 pixels at 1080p / 4K), and bit-exact agreement between the kernel structures, with equal work counters,
on the whole frame.  The scenes are procedural stand-ins of the BASELINE triangle counts (the reference ships no assets,
SURVEY F4).
"""
import numpy as np
import pytest

from helpers import assert_bit_equal, oracle_path_trace_mt, restir_partition_run

pytestmark = pytest.mark.gpu

COUNTER_KEYS = ("closestRays", "anyRays", "nodeVisits", "triTests", "closestHits")


def _torch():
    import torch

    return torch


def test_config1_cornell_400x400_depth4_16spp(gpu_ctx, cornell_full):
    """BASELINE config 1 — the reference's own CPU-runnable case: Cornell stand-in (18 444 tris), 400x400, `Depth 4`, 16
    accumulated frames (iter = looper = 0..15, the running mean of pathtrace.cu:287-290).  The whole frame of every kernel
    structure equals the oracle bit for bit after the 16th frame, and so do the work counters of all 16 frames."""
    from radish_pt_amd import api, scenes

    torch = _torch()
    W = H = 400
    depth, spp = 4, 16
    cam = scenes.cornell_camera(W, H)
    ref_d = np.zeros((W * H, 3), np.float32)
    ref_i = np.zeros((W * H, 3), np.float32)
    stats = {k: 0 for k in COUNTER_KEYS}
    for it in range(spp):
        _, st = oracle_path_trace_mt(cornell_full, cam, ref_d, ref_i, it, it, depth)
        for k in COUNTER_KEYS:
            stats[k] += st[k]
    assert ref_d.max() > 0 and ref_i.max() > 0 and np.isfinite(ref_i).all()
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(cornell_full)
    gpu_ctx.set_camera(cam)
    for name, flags in (("persistent", api.RDH_PT_PERSISTENT), ("megakernel", api.RDH_PT_MEGAKERNEL),
                        ("wavefront+sort", api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL),
                        ("wavefront+sort, sub-frames", api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL | api.RDH_PT_WF_SUBFRAMES),
                        # the per-stage lists of literal-class rays cut to 4 entries: what does not fit stays in the ordinary queues
                        ("wavefront, 4-entry first lists", api.RDH_PT_WAVEFRONT | api.RDH_PT_WF_SMALL_LISTS),
                        ("wavefront+sort, 4-entry first lists", api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL | api.RDH_PT_WF_SMALL_LISTS)):
        d = torch.zeros(W * H, 3, device="cuda")
        i = torch.zeros(W * H, 3, device="cuda")
        gpu_ctx.counters_reset()
        for it in range(spp):
            gpu_ctx.path_trace(d, i, it, it, depth, flags | api.RDH_PT_COUNT)
        assert_bit_equal(d.cpu().numpy(), ref_d, f"config 1 {name}: directIllum after {spp} spp")
        assert_bit_equal(i.cpu().numpy(), ref_i, f"config 1 {name}: indirectIllum after {spp} spp")
        ct = gpu_ctx.counters()
        for k in COUNTER_KEYS:
            assert ct[k] == stats[k], (name, k)


def test_config3_teapots_1080p_wavefront_sort(gpu_ctx):
    """BASELINE config 3: teapots stand-in (100 364 tris), 1920x1080, 8 bounces, the material-sorted wavefront pipeline with
    stream compaction.  The sorted pipeline, the unsorted one, the persistent kernel and the megakernel write identical
    frames and count identical work (22.8 M rays); a strided sample of > 20 000 pixels equals the oracle bit for bit."""
    from radish_pt_amd import api, scenes

    torch = _torch()
    W, H, depth = 1920, 1080, 8
    sd = scenes.teapots()
    assert sd.num_prims == 100364
    cam = scenes.teapots_camera(W, H)
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    gpu_ctx.set_camera(cam)
    out = {}
    for name, flags in (("sort", api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL), ("wave", api.RDH_PT_WAVEFRONT),
                        ("sort2", api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL | api.RDH_PT_WF_SUBFRAMES),
                        ("wave2", api.RDH_PT_WAVEFRONT | api.RDH_PT_WF_SUBFRAMES),
                        ("persist", api.RDH_PT_PERSISTENT), ("mega", api.RDH_PT_MEGAKERNEL)):
        d = torch.zeros(W * H, 3, device="cuda")
        i = torch.zeros(W * H, 3, device="cuda")
        gpu_ctx.counters_reset()
        gpu_ctx.path_trace(d, i, 0, 3, depth, flags | api.RDH_PT_COUNT)
        out[name] = (d.cpu().numpy(), i.cpu().numpy(), gpu_ctx.counters())
    for name in ("wave", "sort2", "wave2", "persist", "mega"):
        assert_bit_equal(out[name][0], out["sort"][0], f"config 3: {name} vs sorted wavefront, direct")
        assert_bit_equal(out[name][1], out["sort"][1], f"config 3: {name} vs sorted wavefront, indirect")
        assert out[name][2] == out["sort"][2], name
    c = out["sort"][2]
    assert c["closestRays"] + c["anyRays"] > 20_000_000 and np.isfinite(out["sort"][1]).all()
    ref_d = np.zeros((W * H, 3), np.float32)
    ref_i = np.zeros((W * H, 3), np.float32)
    stride = 97  # 21 378 pixels, co-prime with the row length: scattered over the whole frame
    idx, _ = oracle_path_trace_mt(sd, cam, ref_d, ref_i, 0, 3, depth, stride=stride)
    assert len(idx) >= 20000
    assert_bit_equal(out["sort"][0][idx], ref_d[idx], "config 3: sorted wavefront direct vs oracle sample")
    assert_bit_equal(out["sort"][1][idx], ref_i[idx], "config 3: sorted wavefront indirect vs oracle sample")
    assert ref_i[idx].max() > 0


def test_config4_scene_restir_vs_oracle(gpu_ctx):
    """Config 4's scene (teapots + 1 024 emissive triangles) against the ORACLE: G-buffer + ReSTIR DI (M = 32, temporal +
    5 spatial), three frames with a moving camera at 320x180 (the oracle's ReSTIR is whole-frame only), images and
    reservoirs bit for bit.  The full-size run of this config is checked through its properties in test_gpu_parity.py."""
    from oracle import pyoracle
    from radish_pt_amd import api, hostlib, layouts as L, scenes

    torch = _torch()
    sd = scenes.teapots(emissive_grid=(16, 32))
    assert sd.num_lights == 1026
    W, H = 320, 180
    n = W * H
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.03 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0) for f in range(3)]
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    gpu_ctx.set_camera(cams[0])
    o = pyoracle.OracleScene(sd)
    gb_ref = pyoracle.GBufferHost(W, H)
    gb = api.GBuffer()
    gb.create(W, H)
    dev = api.DevScene()
    dev.ctx = gpu_ctx
    res = [np.zeros(n, L.RESERVOIR_DTYPE) for _ in range(3)]
    ref = np.zeros((n, 3), np.float32)
    img = torch.zeros(n, 3, device="cuda")
    gpu_ctx.restir_init()
    for f, cam in enumerate(cams):
        o.gbuffer_render(cam, gb_ref)
        gb.render(dev, cam)
        o.restir_direct(cam, ref, 0, 7 + f, res[0], res[1], res[2], gb_ref, f == 0, 3, 1)
        res[0], res[1] = res[1], res[0]
        gpu_ctx.set_camera(cam)
        gpu_ctx.restir_direct(img, 0, 7 + f, gb.c_struct(cam), 3)
        assert_bit_equal(img.cpu().numpy(), ref, f"config 4 scene, ReSTIR frame {f}")
        assert gpu_ctx.restir_read(1).tobytes() == res[1].tobytes(), f"reservoirs frame {f}"
        gb_ref.update(cam)
        gb.update(cam)
    assert ref.mean() > 0.02
    gpu_ctx.restir_free()


@pytest.fixture(scope="module")
def teasets_1m():
    from radish_pt_amd import scenes

    sd = scenes.teapots(segments=200, bands=156, emissive_grid=(16, 32))
    assert 990_000 < sd.num_prims < 1_010_000
    return sd


def test_config5_teasets_4k_path_trace(gpu_ctx, teasets_1m):
    """BASELINE config 5's scene and size on one GPU: ~1.0 M triangles (stand-in for the absent "camera and tea sets" asset),
    3840x2160, 8 bounces.  Persistent kernel: > 20 000 scattered pixels equal the oracle bit for bit; an 8-way tile
    partition (config 5's split, virtual ranks on this GPU) re-assembles to the same frame."""
    from radish_pt_amd import api, scenes

    torch = _torch()
    W, H, depth = 3840, 2160, 8
    sd = teasets_1m
    cam = scenes.teapots_camera(W, H)
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    gpu_ctx.set_camera(cam)
    d = torch.zeros(W * H, 3, device="cuda")
    i = torch.zeros(W * H, 3, device="cuda")
    gpu_ctx.counters_reset()
    gpu_ctx.path_trace(d, i, 0, 5, depth, api.RDH_PT_PERSISTENT | api.RDH_PT_COUNT)
    got_d, got_i, ct = d.cpu().numpy(), i.cpu().numpy(), gpu_ctx.counters()
    assert ct["closestRays"] >= W * H and np.isfinite(got_i).all()
    ref_d = np.zeros((W * H, 3), np.float32)
    ref_i = np.zeros((W * H, 3), np.float32)
    stride = 401  # 20 685 pixels
    idx, _ = oracle_path_trace_mt(sd, cam, ref_d, ref_i, 0, 5, depth, stride=stride)
    assert len(idx) >= 20000
    assert_bit_equal(got_d[idx], ref_d[idx], "config 5: 4K direct vs oracle sample")
    assert_bit_equal(got_i[idx], ref_i[idx], "config 5: 4K indirect vs oracle sample")
    # the 8-way split: every rank's packed tiles, concatenated as the all-gather would, untiled
    world, tile = 8, 64
    shards_d, shards_i = [], []
    total = {k: 0 for k in COUNTER_KEYS}
    for rank in range(world):
        gpu_ctx.set_partition(rank, world, tile)
        tpr = gpu_ctx.tiles_per_rank()
        sd_, si_ = torch.zeros(tpr * tile * tile, 3, device="cuda"), torch.zeros(tpr * tile * tile, 3, device="cuda")
        gpu_ctx.counters_reset()
        gpu_ctx.path_trace(sd_, si_, 0, 5, depth, api.RDH_PT_PERSISTENT | api.RDH_PT_COUNT)
        for k, v in gpu_ctx.counters().items():
            total[k] += v
        shards_d.append(sd_)
        shards_i.append(si_)
    frame_d = torch.zeros(W * H, 3, device="cuda")
    frame_i = torch.zeros(W * H, 3, device="cuda")
    gpu_ctx.untile(torch.cat(shards_d).contiguous(), frame_d)
    gpu_ctx.untile(torch.cat(shards_i).contiguous(), frame_i)
    assert_bit_equal(frame_d.cpu().numpy(), got_d, "config 5: 8 ranks direct")
    assert_bit_equal(frame_i.cpu().numpy(), got_i, "config 5: 8 ranks indirect")
    assert total == ct  # the ranks together traced exactly the single-GPU frame's rays, box tests and triangle tests
    gpu_ctx.set_partition(0, 1, 64)


def test_config5_teasets_4k_restir_8_ranks(gpu_ctx, teasets_1m):
    """BASELINE config 5 proper: ReSTIR DI (M = 32, temporal + spatial) at 3840x2160 on the ~1 M-triangle scene, two frames
    with a moving camera, on 8 virtual ranks (one rdh_ctx each, reservoir exchange between the frames), with ReSTIR's ownership
    granularity of 128x128 pixels dealt round-robin (pass 1 over-computes an 8-pixel apron around a rank's tiles: +27 % pixels at
    128, +56 % at 64 — DESIGN §8; the 64-pixel form is covered at smaller sizes by test_restir_tile_partition_matches_frame):
    frames and every rank's reservoirs equal the single-context run bit for bit."""
    from radish_pt_amd import hostlib

    W, H = 3840, 2160
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.01 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0) for f in range(2)]
    ref_frames, ref_resv = restir_partition_run(gpu_ctx, teasets_1m, W, H, cams, 1, 64, 3)
    frames, resv = restir_partition_run(gpu_ctx, teasets_1m, W, H, cams, 8, 128, 3)
    for f in range(len(cams)):
        assert np.isfinite(ref_frames[f]).all() and ref_frames[f].mean() > 0.02
        assert_bit_equal(frames[f], ref_frames[f], f"config 5 ReSTIR, 8 ranks, frame {f}")
        for r in range(8):
            assert resv[f][r] == ref_resv[f][0], f"config 5 ReSTIR reservoirs, rank {r}, frame {f}"
    gpu_ctx.set_partition(0, 1, 64)


def test_config5_scene_restir_vs_oracle(gpu_ctx, teasets_1m):
    """Config 5's scene against the ORACLE at a size its whole-frame ReSTIR finishes in seconds (480x270): G-buffer planes,
    image and reservoirs of two frames bit for bit."""
    from oracle import pyoracle
    from radish_pt_amd import api, hostlib, layouts as L

    torch = _torch()
    sd = teasets_1m
    W, H = 480, 270
    n = W * H
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.02 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0) for f in range(2)]
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    gpu_ctx.set_camera(cams[0])
    o = pyoracle.OracleScene(sd)
    gb_ref = pyoracle.GBufferHost(W, H)
    gb = api.GBuffer()
    gb.create(W, H)
    dev = api.DevScene()
    dev.ctx = gpu_ctx
    res = [np.zeros(n, L.RESERVOIR_DTYPE) for _ in range(3)]
    ref = np.zeros((n, 3), np.float32)
    img = torch.zeros(n, 3, device="cuda")
    gpu_ctx.restir_init()
    for f, cam in enumerate(cams):
        o.gbuffer_render(cam, gb_ref)
        gb.render(dev, cam)
        cur = gb.frameIdx
        assert np.array_equal(gb.primId[cur].cpu().numpy(), gb_ref.primId[cur])
        assert_bit_equal(gb.depth[cur].cpu().numpy(), gb_ref.depth[cur], "config 5 scene: G-buffer depth")
        o.restir_direct(cam, ref, 0, 11 + f, res[0], res[1], res[2], gb_ref, f == 0, 3, 1)
        res[0], res[1] = res[1], res[0]
        gpu_ctx.set_camera(cam)
        gpu_ctx.restir_direct(img, 0, 11 + f, gb.c_struct(cam), 3)
        assert_bit_equal(img.cpu().numpy(), ref, f"config 5 scene, ReSTIR frame {f}")
        assert gpu_ctx.restir_read(1).tobytes() == res[1].tobytes()
        gb_ref.update(cam)
        gb.update(cam)
    gpu_ctx.restir_free()


@pytest.mark.parametrize("reuse", [3, 0])
def test_config4_restir_1080p_split_equals_fused(gpu_ctx, reuse):
    """BASELINE config 4 at full size (teapots + 1 024 emissive triangles, 1920x1080, M = 32): the split pass 1 (raygen / walk /
    RIS from the LDS light table / walk / resolve — the default) and round 1's fused one-lane-per-pixel kernel write identical
    images and reservoirs over three frames with a moving camera, and count identical work."""
    from radish_pt_amd import api, hostlib, scenes

    torch = _torch()
    sd = scenes.teapots(emissive_grid=(16, 32))
    W, H = 1920, 1080
    n = W * H
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.02 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0) for f in range(3)]
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    gpu_ctx.set_camera(cams[0])
    other, third = api.Context(0), api.Context(0)
    for c in (other, third):
        c.upload_scene(sd)
        c.set_camera(cams[0])
    gb = api.GBuffer()
    gb.create(W, H)
    dev = api.DevScene()
    dev.ctx = gpu_ctx
    img_a, img_b, img_c = (torch.zeros(n, 3, device="cuda") for _ in range(3))
    gpu_ctx.restir_init()
    other.restir_init()
    third.restir_init()
    try:
        for f, cam in enumerate(cams):
            gb.render(dev, cam)
            other.set_camera(cam)
            gpu_ctx.counters_reset()
            other.counters_reset()
            gpu_ctx.restir_direct(img_a, 0, 60 + f, gb.c_struct(cam), reuse, flags=api.RDH_PT_COUNT)
            other.restir_direct(img_b, 0, 60 + f, gb.c_struct(cam), reuse, flags=api.RDH_PT_COUNT | api.RDH_PT_RESTIR_FUSED)
            # the split pass with its literal-class rays traced in place by the walker instead of one per workgroup
            third.set_camera(cam)
            third.counters_reset()
            third.restir_direct(img_c, 0, 60 + f, gb.c_struct(cam), reuse, flags=api.RDH_PT_COUNT | api.RDH_PT_NO_DEFER)
            gpu_ctx.synchronize()
            other.synchronize()
            third.synchronize()
            assert_bit_equal(img_c.cpu().numpy(), img_b.cpu().numpy(), f"split (no defer) vs fused, frame {f}, reuse {reuse}")
            assert third.counters() == other.counters()
            assert_bit_equal(img_a.cpu().numpy(), img_b.cpu().numpy(), f"split vs fused, frame {f}, reuse {reuse}")
            assert gpu_ctx.restir_read(1).tobytes() == other.restir_read(1).tobytes(), f"reservoirs, frame {f}"
            ca, cb = gpu_ctx.counters(), other.counters()
            assert ca == cb and ca["closestRays"] == n and ca["anyRays"] > 0.3 * n, (ca, cb)
            # the rays set aside for the workgroup-per-ray launches are the few finite literal-class ones: the zero-length shadow
            # segments of empty reservoirs (NaN direction, ~5 % of the frame) must not be listed, or the lists overflow their 256
            # entries and every literal-class ray stays in the walker (rdh_restir_read_scratch)
            lists = gpu_ctx.restir_read_scratch(2)
            assert 0 < lists[0] <= 256 and 0 <= lists[1] <= 256, lists[:4]
            segs = gpu_ctx.restir_read_scratch(1)
            assert segs.shape[0] >= n and segs.shape[1] == 6  # one slot per lane of every 8x8 block, NaN where there is no segment
            with np.errstate(invalid="ignore"):
                degenerate = int((np.abs(segs[:, 0:3] - segs[:, 3:6]).max(axis=1) == 0).sum())
            assert degenerate > 256, "this frame is expected to hold zero-length segments (the case the lists must skip)"
            listed = lists[260:260 + int(lists[1])]
            assert not (np.abs(segs[listed, 0:3] - segs[listed, 3:6]).max(axis=1) == 0).any()
            gb.update(cam)
        assert float(img_a.mean()) > 0.05
    finally:
        gpu_ctx.restir_free()
        other.close()
        third.close()
