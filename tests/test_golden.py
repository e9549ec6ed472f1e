"""Committed golden vectors (tests/golden/cornell_small.npz, made by tests/golden/make_golden.py with the CPU oracle).

CPU test: the oracle still reproduces them (regression pin; the reference itself ships no vectors — parity unpinned).
GPU test: the HIP path reproduces the same bytes from the same stored inputs, through the C ABI."""
import hashlib
import os

import numpy as np
import pytest

from helpers import assert_bit_equal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_small.npz")
W, H = 48, 36


@pytest.fixture(scope="module")
def golden():
    from radish_pt_amd import layouts as L, scenes

    g = np.load(GOLDEN, allow_pickle=False)
    mats = np.frombuffer(g["materials"].tobytes(), L.MATERIAL_DTYPE)
    sd = scenes.SceneData("golden", g["vertices"], g["normals"], g["texcoords"], g["material_ids"], mats)
    cam = np.frombuffer(g["camera"].tobytes(), L.CAMERA_DTYPE)[0]
    return g, sd, cam


def test_host_builders_reproduce_golden_bvh_and_alias(golden):
    g, sd, _ = golden
    digest = hashlib.sha256(sd.boxes.tobytes() + b"".join(a.tobytes() for a in sd.nodes)).digest()
    assert digest == g["bvh_sha256"].tobytes()
    assert sd.light_sampler.tobytes() == g["light_sampler"].tobytes()


def test_oracle_reproduces_golden(golden):
    from oracle import pyoracle
    from radish_pt_amd import layouts as L

    g, sd, cam = golden
    o = pyoracle.OracleScene(sd)
    assert o.trace_closest(g["rays"]).tobytes() == g["hits"].tobytes()
    assert np.array_equal(o.trace_occluded(g["segments"]), g["occluded"])
    n = W * H
    d, i = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    for it in range(4):
        o.path_trace(cam, d, i, it, it, 4)
    assert_bit_equal(d, g["pt_direct"], "pathTrace direct")
    assert_bit_equal(i, g["pt_indirect"], "pathTrace indirect")
    dd = np.zeros((n, 3), np.float32)
    o.path_trace_direct(cam, dd, 0, 9)
    assert_bit_equal(dd, g["ptd_direct"], "pathTraceDirect")
    gb = pyoracle.GBufferHost(W, H)
    res = [np.zeros(n, L.RESERVOIR_DTYPE) for _ in range(3)]
    img = np.zeros((n, 3), np.float32)
    for f in range(3):
        o.gbuffer_render(cam, gb)
        o.restir_direct(cam, img, 0, 30 + f, res[0], res[1], res[2], gb, f == 0, 3, 1)
        res[0], res[1] = res[1], res[0]
        gb.update(cam)
    assert_bit_equal(img, g["restir_direct"], "ReSTIR image")
    assert res[1].tobytes() == g["restir_reservoirs"].tobytes()
    cur = gb.frameIdx ^ 1
    assert_bit_equal(gb.albedo, g["gb_albedo"], "albedo")
    assert_bit_equal(gb.normal[cur], g["gb_normal"], "normal")
    assert_bit_equal(gb.depth[cur], g["gb_depth"], "depth")
    assert np.array_equal(gb.primId[cur], g["gb_primId"]) and np.array_equal(gb.motion, g["gb_motion"])


@pytest.mark.gpu
def test_hip_reproduces_golden(golden, gpu_ctx):
    import torch

    from radish_pt_amd import api, layouts as L

    g, sd, cam = golden
    ctx = gpu_ctx
    ctx.set_partition(0, 1, 64)
    ctx.upload_scene(sd)
    ctx.set_camera(cam)
    n = W * H
    rays = torch.from_numpy(g["rays"]).cuda()
    hits = torch.zeros(len(g["rays"]), 4, dtype=torch.int32, device="cuda")
    ctx.trace_closest(rays, hits)
    assert hits.cpu().numpy().tobytes() == g["hits"].tobytes()
    occ = torch.zeros(len(g["segments"]), dtype=torch.int32, device="cuda")
    ctx.trace_occluded(torch.from_numpy(g["segments"]).cuda(), occ)
    assert np.array_equal(occ.cpu().numpy(), g["occluded"])
    for flags in (api.RDH_PT_MEGAKERNEL, api.RDH_PT_WAVEFRONT, api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL,
                  api.RDH_PT_PERSISTENT):
        d = torch.zeros(n, 3, device="cuda")
        i = torch.zeros(n, 3, device="cuda")
        for it in range(4):
            ctx.path_trace(d, i, it, it, 4, flags)
        assert_bit_equal(d.cpu().numpy(), g["pt_direct"], f"pathTrace direct flags={flags}")
        assert_bit_equal(i.cpu().numpy(), g["pt_indirect"], f"pathTrace indirect flags={flags}")
    dd = torch.zeros(n, 3, device="cuda")
    ctx.path_trace_direct(dd, 0, 9)
    assert_bit_equal(dd.cpu().numpy(), g["ptd_direct"], "pathTraceDirect")
    gb = api.GBuffer()
    gb.create(W, H)
    dev = api.DevScene()
    dev.ctx = ctx
    ctx.restir_init()
    img = torch.zeros(n, 3, device="cuda")
    for f in range(3):
        gb.render(dev, cam)
        ctx.restir_direct(img, 0, 30 + f, gb.c_struct(cam), 3)
        gb.update(cam)
    assert_bit_equal(img.cpu().numpy(), g["restir_direct"], "ReSTIR image")
    assert ctx.restir_read(1).tobytes() == g["restir_reservoirs"].tobytes()
    cur = gb.frameIdx ^ 1
    assert_bit_equal(gb.albedo.cpu().numpy(), g["gb_albedo"], "albedo")
    assert_bit_equal(gb.normal[cur].cpu().numpy(), g["gb_normal"], "normal")
    assert_bit_equal(gb.depth[cur].cpu().numpy(), g["gb_depth"], "depth")
    assert np.array_equal(gb.primId[cur].cpu().numpy(), g["gb_primId"])
    assert np.array_equal(gb.motion.cpu().numpy(), g["gb_motion"])
    ctx.restir_free()


# ---------------------------------------------------------------------------------------------------------------------
# tests/golden/post_small.npz (made by tests/golden/make_golden_post.py): display and denoisers on stored inputs
# ---------------------------------------------------------------------------------------------------------------------
POST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "post_small.npz")
PW, PH = 40, 30


def _post_gbuffers(g):
    """The oracle-side G-buffer object for frames 0 and 1 rebuilt from the stored planes."""
    from oracle import pyoracle
    from radish_pt_amd import layouts as L

    gb = pyoracle.GBufferHost(PW, PH)
    for f in range(2):
        cur = gb.frameIdx
        gb.albedo[:] = g[f"gb{f}_albedo"]
        gb.normal[cur][:] = g[f"gb{f}_normal"]
        gb.depth[cur][:] = g[f"gb{f}_depth"]
        gb.primId[cur][:] = g[f"gb{f}_primId"]
        gb.motion[:] = g[f"gb{f}_motion"]
        cam = np.frombuffer(g[f"camera{f}"].tobytes(), L.CAMERA_DTYPE)[0]
        yield f, gb, cam
        gb.update(cam)


def test_oracle_reproduces_post_golden():
    from oracle import pyoracle

    g = np.load(POST, allow_pickle=False)
    n = PW * PH
    accC = [np.zeros((n, 3), np.float32) for _ in range(2)]
    accM = [np.zeros((n, 3), np.float32) for _ in range(2)]
    for f, gb, cam in _post_gbuffers(g):
        noisy = g[f"noisy{f}"]
        ref = pyoracle.denoise_eaw(noisy, gb, cam, 64.0, 0.2, 1.0, 0)
        for lv in (1, 2, 3, 4):
            ref = pyoracle.denoise_eaw(ref, gb, cam, 64.0, 0.2, 1.0, lv)
        assert_bit_equal(ref, g[f"eaw{f}"], f"EAW frame {f}")
        accC[f], accM[f] = pyoracle.denoise_temporal_accumulate(accC[f ^ 1], accM[f ^ 1], noisy, gb, f == 0)
        assert_bit_equal(accC[f], g[f"accum_color{f}"], "accumColor")
        assert_bit_equal(accM[f], g[f"accum_moment{f}"], "accumMoment")
        var = pyoracle.denoise_estimate_variance(accM[f], PW, PH)
        fvar = pyoracle.denoise_filter_variance(var, PW, PH)
        assert_bit_equal(var, g[f"variance{f}"], "variance")
        assert_bit_equal(fvar, g[f"filtered_variance{f}"], "filtered variance")
        col, var2 = pyoracle.denoise_svgf(accC[f], var, fvar, gb, cam, 4.0, 128.0, 1.0, 0)
        assert_bit_equal(col, g[f"svgf_color{f}"], "SVGF colour")
        assert_bit_equal(var2, g[f"svgf_variance{f}"], "SVGF variance")
        mod = pyoracle.denoise_modulate(ref, gb)
        assert_bit_equal(mod, g[f"modulated{f}"], "modulated")
        for tone in (0, 1, 2):
            assert np.array_equal(pyoracle.copy_image_to_pbo(mod, PW, PH, 0, tone, 0.8), g[f"pbo{f}_tone{tone}"])
        assert np.array_equal(pyoracle.copy_image_to_pbo(g[f"gb{f}_motion"], PW, PH, 3), g[f"pbo{f}_motion"])
        assert np.array_equal(pyoracle.copy_image_to_pbo(g[f"gb{f}_depth"] * np.float32(0.2), PW, PH, 2), g[f"pbo{f}_depth"])


@pytest.mark.gpu
def test_gpu_reproduces_post_golden(gpu_ctx):
    """The HIP display and denoiser kernels reproduce the stored bytes from the stored inputs, through the C ABI."""
    import torch

    from radish_pt_amd import api, layouts as L

    g = np.load(POST, allow_pickle=False)
    n = PW * PH
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    gb = api.GBuffer()
    gb.create(PW, PH)
    accC = [torch.zeros(n, 3, device="cuda") for _ in range(2)]
    accM = [torch.zeros(n, 3, device="cuda") for _ in range(2)]
    for f in range(2):
        cam = np.frombuffer(g[f"camera{f}"].tobytes(), L.CAMERA_DTYPE)[0]
        cur = gb.frameIdx
        gb.albedo.copy_(dev(g[f"gb{f}_albedo"]))
        gb.normal[cur].copy_(dev(g[f"gb{f}_normal"]))
        gb.depth[cur].copy_(dev(g[f"gb{f}_depth"]))
        gb.primId[cur].copy_(dev(g[f"gb{f}_primId"]))
        gb.motion.copy_(dev(g[f"gb{f}_motion"]))
        gc = gb.c_struct(cam)
        noisy = dev(g[f"noisy{f}"])
        a, b = torch.zeros(n, 3, device="cuda"), torch.zeros(n, 3, device="cuda")
        gpu_ctx.denoise_eaw(a, noisy, gc, cam, 64.0, 0.2, 1.0, 0)
        for lv in (1, 2, 3, 4):
            gpu_ctx.denoise_eaw(b, a, gc, cam, 64.0, 0.2, 1.0, lv)
            a, b = b, a
        assert_bit_equal(a.cpu().numpy(), g[f"eaw{f}"], f"EAW frame {f}")
        gpu_ctx.denoise_temporal_accumulate(accC[f], accC[f ^ 1], accM[f], accM[f ^ 1], noisy, gc, f == 0)
        assert_bit_equal(accC[f].cpu().numpy(), g[f"accum_color{f}"], "accumColor")
        assert_bit_equal(accM[f].cpu().numpy(), g[f"accum_moment{f}"], "accumMoment")
        var, fvar = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        gpu_ctx.denoise_estimate_variance(var, accM[f], PW, PH)
        gpu_ctx.denoise_filter_variance(fvar, var, PW, PH)
        assert_bit_equal(var.cpu().numpy(), g[f"variance{f}"], "variance")
        assert_bit_equal(fvar.cpu().numpy(), g[f"filtered_variance{f}"], "filtered variance")
        col, var2 = torch.zeros(n, 3, device="cuda"), torch.zeros(n, device="cuda")
        gpu_ctx.denoise_svgf(col, accC[f], var2, var, fvar, gc, cam, 4.0, 128.0, 1.0, 0)
        assert_bit_equal(col.cpu().numpy(), g[f"svgf_color{f}"], "SVGF colour")
        assert_bit_equal(var2.cpu().numpy(), g[f"svgf_variance{f}"], "SVGF variance")
        gpu_ctx.denoise_modulate(a, gc)
        assert_bit_equal(a.cpu().numpy(), g[f"modulated{f}"], "modulated")
        pbo = torch.zeros(n, 4, dtype=torch.uint8, device="cuda")
        for tone in (0, 1, 2):
            gpu_ctx.copy_image_to_pbo(pbo, a, PW, PH, 0, tone, 0.8)
            assert np.array_equal(pbo.cpu().numpy(), g[f"pbo{f}_tone{tone}"])
        gpu_ctx.copy_image_to_pbo(pbo, gb.motion, PW, PH, 3)
        assert np.array_equal(pbo.cpu().numpy(), g[f"pbo{f}_motion"])
        gpu_ctx.copy_image_to_pbo(pbo, dev(g[f"gb{f}_depth"] * np.float32(0.2)), PW, PH, 2)
        assert np.array_equal(pbo.cpu().numpy(), g[f"pbo{f}_depth"])
        gb.update(cam)
