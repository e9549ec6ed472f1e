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
