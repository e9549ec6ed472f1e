"""GPU parity: the HIP path (through the C ABI of libradish_hip.so) against the CPU oracle on identical inputs.

Bar: BIT-EXACT.  The pixel values are float32, and north_star asks for 1e-4 relative agreement with a CPU reference on
identical Sobol sequences; path tracing is chaotic (a 1-ulp difference at a triangle edge changes the whole path), so
the only robust way to meet 1e-4 is to execute the same IEEE-754 operation sequence on both sides and require equality
of every bit.  Each test also reports the fraction of pixels outside 1e-4 relative (must be 0).
"""
import numpy as np
import pytest

from helpers import assert_bit_equal, random_rays, random_segments, restir_partition_run as _restir_partition_run

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4  # north_star: "pixel values within 1e-4 relative of a CPU reference tracer"


def _torch():
    import torch

    return torch


def _frac_outside(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    denom = np.maximum(np.abs(b), 1e-12)
    return float(np.mean(np.abs(a - b) / denom > REL_TOL))


def _oracle(sd):
    from oracle import pyoracle

    return pyoracle.OracleScene(sd)


def _dev(a):
    torch = _torch()
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def cornell_gpu(gpu_ctx, cornell_small):
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(cornell_small)
    return gpu_ctx


# ---------------------------------------------------------------------------------------------------------------------
# traversal
# ---------------------------------------------------------------------------------------------------------------------
def _walker_flags(api, kernel):
    return {"one_lane_per_ray": 0, "persistent": api.RDH_PT_PERSISTENT | api.RDH_PT_NO_PAIRS, "sibling_pairs": api.RDH_PT_PERSISTENT | api.RDH_PT_PAIRS}[kernel]


@pytest.mark.parametrize("kernel", ["one_lane_per_ray", "persistent", "sibling_pairs"])
def test_trace_closest_bit_exact(cornell_gpu, cornell_small, kernel):
    from radish_pt_amd import api, layouts as L

    torch = _torch()
    rays = random_rays(8192, seed=11)
    o = _oracle(cornell_small)
    ref = o.trace_closest(rays)
    d_rays = _dev(rays)
    d_hits = torch.zeros(len(rays), 4, dtype=torch.int32, device="cuda")
    cornell_gpu.counters_reset()
    cornell_gpu.trace_closest(d_rays, d_hits, api.RDH_PT_COUNT | _walker_flags(api, kernel))
    got = d_hits.cpu().numpy().view(L.HIT_DTYPE).reshape(-1)
    assert np.array_equal(got["primId"], ref["primId"])
    for f in ("u", "v", "t"):
        assert_bit_equal(got[f], ref[f], f"hit.{f}")
    assert (ref["primId"] >= 0).mean() > 0.3  # the batch really hits things
    # the device's work counters equal the oracle's: same visiting order, not just same answers
    st, ct = o.stats(), cornell_gpu.counters()
    assert ct["closestRays"] == st["closestRays"] == len(rays)
    assert ct["nodeVisits"] == st["nodeVisits"]
    assert ct["triTests"] == st["triTests"]
    assert ct["closestHits"] == st["closestHits"]


def test_nan_origin_ray_gets_a_record_in_every_walker(cornell_gpu, cornell_small):
    """A caller's ray whose origin.x is NaN is a ray like any other for the public ray-batch entries: DevScene::intersect /
    testOcclusion fail the root's box test and return a miss / not-occluded record, and the ray is counted.  The lane-refill
    walker (RDH_PT_PERSISTENT) uses a NaN first float as the EMPTY-SLOT mark of ReSTIR's per-slot lists — that reading must stay
    inside ReSTIR (round-2 advisor finding: such a ray left hits[i] / occluded[i] unwritten and uncounted)."""
    from radish_pt_amd import api, layouts as L

    torch = _torch()
    rays = random_rays(512, seed=23)
    seg = random_segments(512, seed=29)
    for k in (0, 63, 64, 200, 511):
        rays[k, 0] = np.nan
        seg[k, 0] = np.nan
    rays[17, 4] = np.nan  # a NaN direction component as well
    o = _oracle(cornell_small)
    ref_h, st_h = o.trace_closest(rays), None
    st_h = o.stats()
    o.reset_stats()
    ref_o = o.trace_occluded(seg)
    st_o = o.stats()
    for flags in (0, api.RDH_PT_PERSISTENT):
        d_hits = torch.full((len(rays), 4), 0x7fffffff, dtype=torch.int32, device="cuda")  # poison: an unwritten record shows
        cornell_gpu.counters_reset()
        cornell_gpu.trace_closest(_dev(rays), d_hits, api.RDH_PT_COUNT | flags)
        got = d_hits.cpu().numpy().view(L.HIT_DTYPE).reshape(-1)
        assert np.array_equal(got["primId"], ref_h["primId"]), flags
        for f in ("u", "v", "t"):
            assert_bit_equal(got[f], ref_h[f], f"hit.{f} flags={flags}")
        assert got["primId"][[0, 63, 64, 200, 511]].tolist() == [-1] * 5
        ct = cornell_gpu.counters()
        assert ct["closestRays"] == st_h["closestRays"] == len(rays)
        assert ct["nodeVisits"] == st_h["nodeVisits"] and ct["triTests"] == st_h["triTests"]
        d_occ = torch.full((len(seg),), -7, dtype=torch.int32, device="cuda")
        cornell_gpu.counters_reset()
        cornell_gpu.trace_occluded(_dev(seg), d_occ, api.RDH_PT_COUNT | flags)
        assert np.array_equal(d_occ.cpu().numpy(), ref_o), flags
        ct = cornell_gpu.counters()
        assert ct["anyRays"] == st_o["anyRays"] == len(seg)
        assert ct["nodeVisits"] == st_o["nodeVisits"] and ct["triTests"] == st_o["triTests"]


@pytest.mark.parametrize("kernel", ["one_lane_per_ray", "persistent", "sibling_pairs"])
def test_trace_occluded_exact(cornell_gpu, cornell_small, kernel):
    from radish_pt_amd import api

    torch = _torch()
    seg = random_segments(8192, seed=5)
    o = _oracle(cornell_small)
    ref = o.trace_occluded(seg)
    d_out = torch.full((len(seg),), -1, dtype=torch.int32, device="cuda")
    cornell_gpu.counters_reset()
    cornell_gpu.trace_occluded(_dev(seg), d_out, api.RDH_PT_COUNT | _walker_flags(api, kernel))
    assert np.array_equal(d_out.cpu().numpy(), ref)
    assert 0.05 < ref.mean() < 0.95
    st, ct = o.stats(), cornell_gpu.counters()
    assert ct["anyRays"] == st["anyRays"] == len(seg)
    assert ct["nodeVisits"] == st["nodeVisits"] and ct["triTests"] == st["triTests"]


def test_trace_empty_and_ragged(cornell_gpu, cornell_small):
    torch = _torch()
    cornell_gpu.trace_closest(torch.zeros(0, 6, device="cuda"), torch.zeros(0, 4, dtype=torch.int32, device="cuda"))
    rays = random_rays(257, seed=3)  # not a multiple of the 256-lane workgroup
    ref = _oracle(cornell_small).trace_closest(rays)
    from radish_pt_amd import api
    for flags in (0, api.RDH_PT_PERSISTENT):
        d_hits = torch.full((257, 4), 7, dtype=torch.int32, device="cuda")
        cornell_gpu.trace_closest(_dev(rays), d_hits, flags)
        assert np.array_equal(d_hits.cpu().numpy()[:, 0], ref["primId"])
    # axis-parallel directions (the box test's special cases, bvh.h:138-148) through the lane-refill walker's whole-wave trace
    rays[::5, 3:] = np.array([0.0, 0.0, -1.0], np.float32)
    rays[1::7, 3:] = np.array([1.0, 0.0, 0.0], np.float32)
    ref = _oracle(cornell_small).trace_closest(rays)
    d_hits = torch.full((257, 4), 7, dtype=torch.int32, device="cuda")
    cornell_gpu.trace_closest(_dev(rays), d_hits, api.RDH_PT_PERSISTENT)
    assert np.array_equal(d_hits.cpu().numpy()[:, 0], ref["primId"])


def test_workgroup_per_ray_trace_bit_exact(cornell_gpu, cornell_small):
    """wgTraceWhole (device/wg_trace.h), the routine behind k_gbuffer_literal, on every kind of ray: one
    1 024-thread workgroup per ray, closest-hit and any-hit, random rays and axis-parallel / tiny-component ones (the box
    test's special cases, bvh.h:138-148).  Records and work counters equal the oracle's sequential walk."""
    from radish_pt_amd import api, layouts as L

    torch = _torch()
    o = _oracle(cornell_small)
    rays = random_rays(1500, seed=21)
    rays[::4, 3:] = np.array([0.0, 0.0, -1.0], np.float32)
    rays[1::9, 3:] = np.array([0.0, 1.0, 0.0], np.float32)
    d = rays[2::11, 3:].copy()
    d[:, 0] = 3e-7
    rays[2::11, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    before = o.stats()
    ref = o.trace_closest(rays)
    after = o.stats()
    d_hits = torch.zeros(len(rays), 4, dtype=torch.int32, device="cuda")
    cornell_gpu.counters_reset()
    cornell_gpu.trace_closest(_dev(rays), d_hits, api.RDH_PT_COUNT | api.RDH_PT_WG_PER_RAY)
    got = d_hits.cpu().numpy().view(L.HIT_DTYPE).reshape(-1)
    assert np.array_equal(got["primId"], ref["primId"])
    for f in ("u", "v", "t"):
        assert_bit_equal(got[f], ref[f], f"hit.{f}")
    ct = cornell_gpu.counters()
    for k in ("closestRays", "nodeVisits", "triTests", "closestHits"):
        assert ct[k] == after[k] - before[k], k
    assert (ref["primId"] >= 0).mean() > 0.3

    seg = random_segments(1500, seed=8)
    seg[::4, 3:5] = seg[::4, 0:2]              # along z
    seg[1::7, 4:6] = seg[1::7, 1:3]            # along x
    before = o.stats()
    ref_o = o.trace_occluded(seg)
    after = o.stats()
    d_out = torch.full((len(seg),), -1, dtype=torch.int32, device="cuda")
    cornell_gpu.counters_reset()
    cornell_gpu.trace_occluded(_dev(seg), d_out, api.RDH_PT_COUNT | api.RDH_PT_WG_PER_RAY)
    assert np.array_equal(d_out.cpu().numpy(), ref_o)
    ct = cornell_gpu.counters()
    for k in ("anyRays", "nodeVisits", "triTests"):
        assert ct[k] == after[k] - before[k], k
    assert 0.05 < ref_o.mean() < 0.95

# ---------------------------------------------------------------------------------------------------------------------
# pathTrace: megakernel and wavefront, several frames of accumulation
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("flags_name", ["mega", "wavefront", "wavefront_sort", "persistent", "persistent_threaded", "wavefront_threaded", "auto"])
@pytest.mark.parametrize("size,depth", [((64, 48), 4), ((37, 29), 8)])
def test_path_trace_bit_exact(cornell_gpu, cornell_small, flags_name, size, depth):
    from radish_pt_amd import api, scenes

    torch = _torch()
    flags = {"mega": api.RDH_PT_MEGAKERNEL, "wavefront": api.RDH_PT_WAVEFRONT,
             "wavefront_sort": api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL,
             "persistent": api.RDH_PT_PERSISTENT,
             # the walks over the six threaded arrays (the default since round 3 is the sibling pairs) and the library's own choice
             "persistent_threaded": api.RDH_PT_PERSISTENT | api.RDH_PT_NO_PAIRS,
             "wavefront_threaded": api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL | api.RDH_PT_NO_PAIRS,
             "auto": api.RDH_PT_AUTO}[flags_name] | api.RDH_PT_COUNT
    W, H = size
    cam = scenes.cornell_camera(W, H)
    o = _oracle(cornell_small)
    ref_d = np.zeros((W * H, 3), np.float32)
    ref_i = np.zeros((W * H, 3), np.float32)
    d = torch.zeros(W * H, 3, device="cuda")
    i = torch.zeros(W * H, 3, device="cuda")
    cornell_gpu.set_camera(cam)
    cornell_gpu.counters_reset()
    for it in range(3):
        looper = 17 + it
        o.path_trace(cam, ref_d, ref_i, it, looper, depth)
        cornell_gpu.path_trace(d, i, it, looper, depth, flags)
    got_d, got_i = d.cpu().numpy(), i.cpu().numpy()
    assert _frac_outside(got_d, ref_d) == 0.0 and _frac_outside(got_i, ref_i) == 0.0
    assert_bit_equal(got_d, ref_d, "directIllum")
    assert_bit_equal(got_i, ref_i, "indirectIllum")
    assert ref_i.max() > 0 and ref_d.max() > 0
    st, ct = o.stats(), cornell_gpu.counters()
    for k in ("closestRays", "anyRays", "nodeVisits", "triTests", "closestHits"):
        assert ct[k] == st[k], k


def test_dump_rays_reproduces_the_frames_work(cornell_gpu, cornell_small):
    """rdh_dump_rays (the ray lists behind bench.py's roofline.traversal_only and the CPU traversal denominator): the lists hold
    exactly the frame's rays — walking them costs the frame's node visits, triangle tests and hits, on the GPU (walk-only kernel)
    and on the oracle; their hit records and occlusion flags equal the oracle's."""
    from radish_pt_amd import api, layouts as L, scenes

    torch = _torch()
    W, H, depth, looper = 96, 64, 5, 12
    cam = scenes.cornell_camera(W, H)
    cornell_gpu.set_camera(cam)
    d, i = torch.zeros(W * H, 3, device="cuda"), torch.zeros(W * H, 3, device="cuda")
    cornell_gpu.counters_reset()
    cornell_gpu.path_trace(d, i, 0, looper, depth, api.RDH_PT_PERSISTENT | api.RDH_PT_COUNT)
    frame = cornell_gpu.counters()
    closest, segs = cornell_gpu.dump_rays(looper, depth)
    assert closest.shape[0] == frame["closestRays"] and segs.shape[0] == frame["anyRays"]
    hits = torch.zeros(closest.shape[0], 4, dtype=torch.int32, device="cuda")
    occ = torch.zeros(segs.shape[0], dtype=torch.int32, device="cuda")
    cornell_gpu.counters_reset()
    cornell_gpu.trace_closest(closest.contiguous(), hits, api.RDH_PT_PERSISTENT | api.RDH_PT_COUNT)
    cornell_gpu.trace_occluded(segs.contiguous(), occ, api.RDH_PT_PERSISTENT | api.RDH_PT_COUNT)
    assert cornell_gpu.counters() == frame
    o = _oracle(cornell_small)
    ref_h = o.trace_closest(closest.cpu().numpy())
    ref_o = o.trace_occluded(segs.cpu().numpy())
    assert o.stats() == frame
    got = hits.cpu().numpy().view(L.HIT_DTYPE).reshape(-1)
    assert np.array_equal(got["primId"], ref_h["primId"]) and np.array_equal(occ.cpu().numpy(), ref_o)
    assert_bit_equal(got["t"], ref_h["t"], "hit.t of the dumped rays")


def test_path_trace_depth_zero_and_one(cornell_gpu, cornell_small):
    from radish_pt_amd import api, scenes

    torch = _torch()
    W, H = 32, 24
    cam = scenes.cornell_camera(W, H)
    cornell_gpu.set_camera(cam)
    o = _oracle(cornell_small)
    for depth in (0, 1):
        for flags in (api.RDH_PT_MEGAKERNEL, api.RDH_PT_WAVEFRONT, api.RDH_PT_PERSISTENT):
            ref_d = np.zeros((W * H, 3), np.float32)
            ref_i = np.zeros((W * H, 3), np.float32)
            o.path_trace(cam, ref_d, ref_i, 0, 3, depth)
            d = torch.zeros(W * H, 3, device="cuda")
            i = torch.zeros(W * H, 3, device="cuda")
            cornell_gpu.path_trace(d, i, 0, 3, depth, flags)
            assert_bit_equal(d.cpu().numpy(), ref_d, f"direct depth={depth} flags={flags}")
            assert_bit_equal(i.cpu().numpy(), ref_i, f"indirect depth={depth} flags={flags}")


def test_path_trace_direct_bit_exact(cornell_gpu, cornell_small):
    from radish_pt_amd import scenes

    torch = _torch()
    W, H = 56, 40
    cam = scenes.cornell_camera(W, H)
    cornell_gpu.set_camera(cam)
    o = _oracle(cornell_small)
    ref = np.zeros((W * H, 3), np.float32)
    d = torch.zeros(W * H, 3, device="cuda")
    for it in range(2):
        o.path_trace_direct(cam, ref, it, 100 + it)
        cornell_gpu.path_trace_direct(d, it, 100 + it)
    assert_bit_equal(d.cpu().numpy(), ref, "pathTraceDirect")
    assert ref.max() > 0


# ---------------------------------------------------------------------------------------------------------------------
# G-buffer + ReSTIR, three frames with a moving camera (temporal reuse across different motion vectors)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("reuse", [0, 1, 2, 3])
@pytest.mark.parametrize("faithful", [1, 0])
def test_restir_sequence_bit_exact(gpu_ctx, reuse, faithful):
    """(Pass 1's primary rays: walked as packets — the default — when `faithful`, lane by lane otherwise, so both walkers of the
    jittered primary ray go through every reuse mode.)"""
    from oracle import pyoracle
    from radish_pt_amd import api, hostlib, layouts as L, scenes

    torch = _torch()
    sd = scenes.teapots(segments=12, bands=8, grid=2, emissive_grid=(4, 8))
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    o = _oracle(sd)
    W, H = 48, 40
    n = W * H
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.05 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0)
            for f in range(3)]
    gb_ref = pyoracle.GBufferHost(W, H)
    gb = api.GBuffer()
    gb.create(W, H)
    dev = api.DevScene()
    dev.ctx = gpu_ctx
    res = [np.zeros(n, L.RESERVOIR_DTYPE) for _ in range(3)]  # cur, last, temp (restir.cu:4-7)
    ref_img = np.zeros((n, 3), np.float32)
    img = torch.zeros(n, 3, device="cuda")
    gpu_ctx.set_camera(cams[0])
    gpu_ctx.restir_init()
    for f, cam in enumerate(cams):
        o.gbuffer_render(cam, gb_ref)
        gb.render(dev, cam)
        cur = gb.frameIdx
        assert np.array_equal(gb.primId[cur].cpu().numpy(), gb_ref.primId[cur]), "gbuffer id"
        assert np.array_equal(gb.motion.cpu().numpy(), gb_ref.motion), "gbuffer motion"
        assert_bit_equal(gb.albedo.cpu().numpy(), gb_ref.albedo, "gbuffer albedo")
        assert_bit_equal(gb.normal[cur].cpu().numpy(), gb_ref.normal[cur], "gbuffer normal")
        assert_bit_equal(gb.depth[cur].cpu().numpy(), gb_ref.depth[cur], "gbuffer depth")
        before = o.stats()
        o.restir_direct(cam, ref_img, 0, 40 + f, res[0], res[1], res[2], gb_ref, f == 0, reuse, faithful)
        after = o.stats()
        res[0], res[1] = res[1], res[0]  # std::swap(directReservoir, lastDirectReservoir)
        gpu_ctx.set_camera(cam)
        gpu_ctx.counters_reset()
        gpu_ctx.restir_direct(img, 0, 40 + f, gb.c_struct(cam), reuse, faithful_ris=faithful,
                              flags=api.RDH_PT_COUNT | (0 if faithful else api.RDH_PT_NO_PACKETS)
                              | (api.RDH_PT_WF_SMALL_LISTS if (faithful and reuse == 3) else 0))  # one case: packets of 4 visits
        got = img.cpu().numpy()
        assert _frac_outside(got, ref_img) == 0.0
        assert_bit_equal(got, ref_img, f"ReSTIR frame {f} reuse={reuse}")
        last = gpu_ctx.restir_read(1)  # what this frame wrote is now `last`
        assert last.tobytes() == res[1].tobytes(), f"reservoirs frame {f}"
        ct = gpu_ctx.counters()
        for k in ("closestRays", "anyRays", "nodeVisits", "triTests", "closestHits"):
            assert ct[k] == after[k] - before[k], k
        gb_ref.update(cam)
        gb.update(cam)
    assert ref_img.max() > 0
    gpu_ctx.restir_free()


@pytest.mark.parametrize("kernel", ["packets", "packets_budget_4", "persistent", "one_lane_per_pixel"])
def test_gbuffer_kernels_bit_exact(gpu_ctx, kernel):
    """renderGBuffer (gBuffer.cu:3-76): the packet kernel (one 8x8 block per wave, its rays walked together: the default), the
    persistent lane-refill kernel and the one-lane-per-pixel kernel all equal the oracle plane by plane, with equal work
    counters, at a size that is not a multiple of the 8x8 block."""
    from oracle import pyoracle
    from radish_pt_amd import api, hostlib, scenes

    sd = scenes.teapots(segments=12, bands=8, grid=2, emissive_grid=(4, 8))
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    o = _oracle(sd)
    W, H = 150, 91
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.1 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0) for f in range(2)]
    gb_ref = pyoracle.GBufferHost(W, H)
    gb = api.GBuffer()
    gb.create(W, H)
    # packets_budget_4: a wave walks its block as a packet for 4 visits only, then every lane goes on by itself (every block takes the
    # hand-over of traverse.h's packetWalk; 256 visits normally, which only blocks on silhouettes of large scenes exceed)
    flags = api.RDH_PT_COUNT | {"one_lane_per_pixel": api.RDH_PT_MEGA_GBUFFER, "persistent": api.RDH_PT_NO_PACKETS, "packets": 0,
                                "packets_budget_4": api.RDH_PT_WF_SMALL_LISTS}[kernel]
    for cam in cams:
        o.stats_reset() if hasattr(o, "stats_reset") else None
        before = o.stats()
        o.gbuffer_render(cam, gb_ref)
        after = o.stats()
        gpu_ctx.set_camera(cam)
        gpu_ctx.counters_reset()
        gpu_ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), flags)
        gpu_ctx.synchronize()
        cur = gb.frameIdx
        assert np.array_equal(gb.primId[cur].cpu().numpy(), gb_ref.primId[cur])
        assert np.array_equal(gb.motion.cpu().numpy(), gb_ref.motion)
        assert_bit_equal(gb.albedo.cpu().numpy(), gb_ref.albedo, "albedo")
        assert_bit_equal(gb.normal[cur].cpu().numpy(), gb_ref.normal[cur], "normal")
        assert_bit_equal(gb.depth[cur].cpu().numpy(), gb_ref.depth[cur], "depth")
        ct = gpu_ctx.counters()
        for k in ("closestRays", "nodeVisits", "triTests", "closestHits"):
            assert ct[k] == after[k] - before[k], k
        gb_ref.update(cam)
        gb.update(cam)
    assert (gb_ref.primId[0] >= 0).mean() > 0.3


def _count_literal_primary_rays(cam):
    """How many un-jittered centre rays of `cam` have a direction component below the box test's Eps (bvh.h:138-148)."""
    W, H = int(cam["resolution"][0]), int(cam["resolution"][1])
    x, y = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    ru = 1.0 - (x + 0.5) / W * 2.0
    rv = 1.0 - (y + 0.5) / H * 2.0
    t = float(cam["tanFovY"])
    d = (ru * (W / H) * t)[..., None] * cam["right"].astype(np.float64) + (rv * t)[..., None] * cam["up"].astype(np.float64) \
        + cam["view"].astype(np.float64)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return int((np.abs(d) < 5e-7).any(axis=-1).sum())


@pytest.mark.parametrize("size", [(151, 91), (301, 45)])
@pytest.mark.parametrize("defer", [True, False])
@pytest.mark.parametrize("packets", [True, False])
def test_gbuffer_literal_rays_bit_exact(gpu_ctx, defer, size, packets):
    """Primary rays with an axis-parallel direction component take the box test's special cases (bvh.h:138-148) and visit
    a large part of the tree.  rdh_gbuffer_render lists them for k_gbuffer_literal, where a whole workgroup traces
    each (wg_trace.h) — unless the frame has more than 256 of them (the second size) or RDH_PT_NO_DEFER is given: then one wave
    traces each in place.  An axis-aligned camera with odd dimensions makes the whole centre row and column such rays; planes
    and work counters equal the oracle's either way."""
    from oracle import pyoracle
    from radish_pt_amd import api, hostlib, scenes

    sd = scenes.teapots(segments=12, bands=8, grid=2, emissive_grid=(4, 8))
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    o = _oracle(sd)
    W, H = size
    cams = [hostlib.make_camera(W, H, eye=(0.0, 1.0 + 0.5 * f, 9.0), rotation=(-90.0, 0.0, 0.0), fovy=19.0) for f in range(2)]
    n_lit = _count_literal_primary_rays(cams[0])
    assert (50 <= n_lit <= 256) if W == 151 else n_lit > 256
    gb_ref = pyoracle.GBufferHost(W, H)
    gb = api.GBuffer()
    gb.create(W, H)
    flags = api.RDH_PT_COUNT | (0 if defer else api.RDH_PT_NO_DEFER) | (0 if packets else api.RDH_PT_NO_PACKETS)
    for cam in cams:
        before = o.stats()
        o.gbuffer_render(cam, gb_ref)
        after = o.stats()
        gpu_ctx.set_camera(cam)
        gpu_ctx.counters_reset()
        gpu_ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), flags)
        gpu_ctx.synchronize()
        cur = gb.frameIdx
        assert np.array_equal(gb.primId[cur].cpu().numpy(), gb_ref.primId[cur])
        assert np.array_equal(gb.motion.cpu().numpy(), gb_ref.motion)
        assert_bit_equal(gb.albedo.cpu().numpy(), gb_ref.albedo, "albedo")
        assert_bit_equal(gb.normal[cur].cpu().numpy(), gb_ref.normal[cur], "normal")
        assert_bit_equal(gb.depth[cur].cpu().numpy(), gb_ref.depth[cur], "depth")
        ct = gpu_ctx.counters()
        for k in ("closestRays", "nodeVisits", "triTests", "closestHits"):
            assert ct[k] == after[k] - before[k], k
        gb_ref.update(cam)
        gb.update(cam)
    assert (gb_ref.primId[0] >= 0).mean() > 0.2


# ---------------------------------------------------------------------------------------------------------------------
# Full-size properties (oracle too slow): implementations must agree with each other bit for bit at 1080p
# ---------------------------------------------------------------------------------------------------------------------
def test_1080p_variants_agree(gpu_ctx, cornell_full):
    from radish_pt_amd import api, scenes

    torch = _torch()
    W, H = 1920, 1080
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(cornell_full)
    cam = scenes.cornell_camera(W, H)
    gpu_ctx.set_camera(cam)
    out = {}
    for name, flags in (("mega", 0), ("wave", api.RDH_PT_WAVEFRONT), ("sort", api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL),
                        ("persist", api.RDH_PT_PERSISTENT)):
        d = torch.zeros(W * H, 3, device="cuda")
        i = torch.zeros(W * H, 3, device="cuda")
        gpu_ctx.counters_reset()
        gpu_ctx.path_trace(d, i, 0, 0, 8, flags | api.RDH_PT_COUNT)
        out[name] = (d.cpu().numpy(), i.cpu().numpy(), gpu_ctx.counters())
    for name in ("wave", "sort", "persist"):
        assert_bit_equal(out[name][0], out["mega"][0], f"{name} direct")
        assert_bit_equal(out[name][1], out["mega"][1], f"{name} indirect")
        assert out[name][2] == out["mega"][2]
    # spot-check 2 000 scattered pixels of the 1080p frame against the oracle
    o = _oracle(cornell_full)
    ref_d = np.zeros((W * H, 3), np.float32)
    ref_i = np.zeros((W * H, 3), np.float32)
    stride = (W * H) // 2000 + 1
    o.path_trace(cam, ref_d, ref_i, 0, 0, 8, pix=(0, W * H, stride))
    idx = np.arange(0, W * H, stride)
    assert_bit_equal(out["mega"][0][idx], ref_d[idx], "1080p direct vs oracle sample")
    assert_bit_equal(out["mega"][1][idx], ref_i[idx], "1080p indirect vs oracle sample")
    c = out["mega"][2]
    assert c["closestRays"] >= W * H and np.isfinite(out["mega"][1]).all()


def test_gbuffer_back_to_back_without_sync(gpu_ctx):
    """rdh_gbuffer_render forks to an internal second stream and joins again; calls enqueued back to back (events re-recorded while
    earlier waits are still pending) must leave exactly what synchronised calls leave."""
    from radish_pt_amd import api, hostlib, scenes

    sd = scenes.teapots(segments=12, bands=8, grid=2, emissive_grid=(4, 8))
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    W, H = 151, 91
    cams = [hostlib.make_camera(W, H, eye=(0.0, 1.0 + 0.1 * f, 9.0), rotation=(-90.0, 0.0, 0.0), fovy=19.0) for f in range(12)]
    ref = api.GBuffer()
    ref.create(W, H)
    gpu_ctx.set_camera(cams[-1])
    gpu_ctx.gbuffer_render(ref.c_struct(cam_fallback=cams[-1]), 0)
    gpu_ctx.synchronize()
    gb = api.GBuffer()
    gb.create(W, H)
    for cam in cams:  # same buffers every time: a late write of an earlier call would show
        gpu_ctx.set_camera(cam)
        gpu_ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), 0)
    gpu_ctx.synchronize()
    f = gb.frameIdx
    for a, b, plane in ((gb.albedo, ref.albedo, "albedo"), (gb.normal[f], ref.normal[f], "normal"), (gb.depth[f], ref.depth[f], "depth"),
                        (gb.primId[f], ref.primId[f], "primId")):
        assert_bit_equal(a.cpu().numpy(), b.cpu().numpy(), plane)


def test_gbuffer_1080p_variants_agree(gpu_ctx):
    """Full size (the oracle is too slow for 2 M primary rays of the teapots frame): the three G-buffer structures — lane refill
    with the workgroup-per-ray launch for literal-class rays, lane refill tracing them in place, one lane per pixel — write
    identical planes and count identical work, for the bench camera (a handful of such rays) and for an axis-aligned camera
    (thousands: above the cap, none is set aside)."""
    from radish_pt_amd import api, hostlib, scenes

    W, H = 1920, 1080
    sd = scenes.teapots(emissive_grid=(16, 32))
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    cams = [scenes.teapots_camera(W, H), hostlib.make_camera(W - 1, H - 1, eye=(0.0, 1.5, 9.0), rotation=(-90.0, 0.0, 0.0), fovy=19.0)]
    assert _count_literal_primary_rays(cams[1]) > 256
    for cam in cams:
        w, h = int(cam["resolution"][0]), int(cam["resolution"][1])
        gpu_ctx.set_camera(cam)
        out = {}
        for name, flags in (("defer", 0), ("in_place", api.RDH_PT_NO_DEFER), ("one_lane", api.RDH_PT_ONE_LANE_PER_PIXEL)):
            gb = api.GBuffer()
            gb.create(w, h)
            gpu_ctx.counters_reset()
            gpu_ctx.gbuffer_render(gb.c_struct(cam_fallback=cam), flags | api.RDH_PT_COUNT)
            gpu_ctx.synchronize()
            f = gb.frameIdx
            out[name] = ([gb.albedo.cpu().numpy(), gb.normal[f].cpu().numpy(), gb.depth[f].cpu().numpy(),
                          gb.primId[f].cpu().numpy(), gb.motion.cpu().numpy()], gpu_ctx.counters())
        for name in ("defer", "in_place"):
            for a, b, plane in zip(out[name][0], out["one_lane"][0], ("albedo", "normal", "depth", "primId", "motion")):
                assert_bit_equal(a, b, f"{name} {plane}")
            assert out[name][1] == out["one_lane"][1], name
        assert out["defer"][1]["closestRays"] == w * h and (out["defer"][0][3] >= 0).mean() > 0.3


def test_tile_partition_matches_frame(gpu_ctx, cornell_small):
    """N virtual ranks on one GPU: packed tile buffers → (simulated) all-gather → rdh_untile == single-GPU frame."""
    from radish_pt_amd import api, scenes

    torch = _torch()
    W, H = 200, 120  # not multiples of the tile size
    cam = scenes.cornell_camera(W, H)
    gpu_ctx.upload_scene(cornell_small)
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.set_camera(cam)
    full_d = torch.zeros(W * H, 3, device="cuda")
    full_i = torch.zeros(W * H, 3, device="cuda")
    gpu_ctx.path_trace(full_d, full_i, 0, 9, 4, api.RDH_PT_WAVEFRONT)
    for world, tile in ((2, 64), (3, 32), (8, 16)):
        shards_d, shards_i = [], []
        for rank in range(world):
            gpu_ctx.set_partition(rank, world, tile)
            tpr = gpu_ctx.tiles_per_rank()
            d = torch.zeros(tpr * tile * tile, 3, device="cuda")
            i = torch.zeros(tpr * tile * tile, 3, device="cuda")
            gpu_ctx.path_trace(d, i, 0, 9, 4, (api.RDH_PT_WAVEFRONT, api.RDH_PT_MEGAKERNEL, api.RDH_PT_PERSISTENT)[rank % 3])
            shards_d.append(d)
            shards_i.append(i)
        frame_d = torch.zeros(W * H, 3, device="cuda")
        frame_i = torch.zeros(W * H, 3, device="cuda")
        gpu_ctx.untile(torch.cat(shards_d).contiguous(), frame_d)
        gpu_ctx.untile(torch.cat(shards_i).contiguous(), frame_i)
        assert_bit_equal(frame_d.cpu().numpy(), full_d.cpu().numpy(), f"world={world} direct")
        assert_bit_equal(frame_i.cpu().numpy(), full_i.cpu().numpy(), f"world={world} indirect")
    gpu_ctx.set_partition(0, 1, 64)


def test_restir_tile_partition_matches_frame(gpu_ctx):
    """ReSTIR DI on a tile partition (N virtual ranks, one rdh_ctx each, on one GPU): whole-frame G-buffer on every rank,
    pass 1 over the rank's tiles + 8-px apron, pass 2 local, reservoir exchange (pack -> simulated all-gather -> unpack)
    for next frame's temporal reuse.  Image and reservoirs must equal the single-GPU frame bit for bit, with a moving
    camera (motion vectors cross tile borders)."""
    from radish_pt_amd import hostlib, scenes

    sd = scenes.teapots(segments=12, bands=8, grid=2, emissive_grid=(4, 8))
    W, H = 150, 90  # not multiples of the tile size
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.08 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0)
            for f in range(3)]

    def run(world, tile, reuse):
        return _restir_partition_run(gpu_ctx, sd, W, H, cams, world, tile, reuse)

    for reuse in (3, 1, 2):
        ref_frames, ref_resv = run(1, 64, reuse)
        assert max(f.max() for f in ref_frames) > 0
        for world, tile in ((2, 64), (3, 32), (5, 16)):
            frames, resv = run(world, tile, reuse)
            for f in range(len(cams)):
                assert_bit_equal(frames[f], ref_frames[f], f"ReSTIR world={world} tile={tile} reuse={reuse} frame {f}")
                for r in range(world):
                    assert resv[f][r] == ref_resv[f][0], f"reservoirs world={world} rank {r} frame {f}"
    gpu_ctx.set_partition(0, 1, 64)


def test_restir_1080p_partition_and_determinism(gpu_ctx):
    """BASELINE config 4 at full size, in config 5's structure: teapots + 1 024 emissive triangles, 1920x1080, ReSTIR DI M = 32
    with temporal + spatial reuse, two frames with a moving camera.  The oracle is too slow here, so the properties: the
    two-pass ReSTIR is deterministic (the reference's single kernel is not), and 8 virtual ranks with the
    reservoir exchange give the single-GPU frames and reservoirs bit for bit."""
    from radish_pt_amd import hostlib, scenes

    sd = scenes.teapots(emissive_grid=(16, 32))
    W, H = 1920, 1080
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.02 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0) for f in range(2)]
    a_frames, a_resv = _restir_partition_run(gpu_ctx, sd, W, H, cams, 1, 64, 3)
    b_frames, b_resv = _restir_partition_run(gpu_ctx, sd, W, H, cams, 1, 64, 3)
    for f in range(len(cams)):
        assert_bit_equal(b_frames[f], a_frames[f], f"second run, frame {f}")
        assert b_resv[f] == a_resv[f]
        assert np.isfinite(a_frames[f]).all() and a_frames[f].mean() > 0.05
    frames, resv = _restir_partition_run(gpu_ctx, sd, W, H, cams, 8, 64, 3)
    for f in range(len(cams)):
        assert_bit_equal(frames[f], a_frames[f], f"8 ranks, frame {f}")
        for r in range(8):
            assert resv[f][r] == a_resv[f][0], f"reservoirs of rank {r}, frame {f}"
    gpu_ctx.set_partition(0, 1, 64)


# ---------------------------------------------------------------------------------------------------------------------
# Textured materials (base colour / procedural / metallic / roughness / normal map) and the environment map
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("env", [True, False])
def test_textures_and_envmap_bit_exact(gpu_ctx, env):
    from oracle import pyoracle
    from radish_pt_amd import api, layouts as L, scenes

    torch = _torch()
    sd = scenes.cornell_textured(segments=12, bands=8, env=env)
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    o = _oracle(sd)
    W, H, depth = 56, 40, 5
    n = W * H
    cam = scenes.cornell_camera(W, H)
    gpu_ctx.set_camera(cam)
    ref_d, ref_i = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    for it in range(2):
        o.path_trace(cam, ref_d, ref_i, it, 60 + it, depth)
    st = o.stats()
    for name, flags in (("mega", 0), ("wavefront", api.RDH_PT_WAVEFRONT), ("sort", api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL),
                        ("persistent", api.RDH_PT_PERSISTENT)):
        d, i = torch.zeros(n, 3, device="cuda"), torch.zeros(n, 3, device="cuda")
        gpu_ctx.counters_reset()
        for it in range(2):
            gpu_ctx.path_trace(d, i, it, 60 + it, depth, flags | api.RDH_PT_COUNT)
        assert_bit_equal(d.cpu().numpy(), ref_d, f"{name} direct (env={env})")
        assert_bit_equal(i.cpu().numpy(), ref_i, f"{name} indirect (env={env})")
        assert gpu_ctx.counters() == st, name
    ref = np.zeros((n, 3), np.float32)
    o.path_trace_direct(cam, ref, 0, 7)
    dd = torch.zeros(n, 3, device="cuda")
    gpu_ctx.path_trace_direct(dd, 0, 7)
    assert_bit_equal(dd.cpu().numpy(), ref, "pathTraceDirect")
    if env:
        assert ref[0].max() > 0  # the corner pixel sees the environment map
    # G-buffer + ReSTIR (two frames: temporal + spatial), textured metallic/roughness carried across the pass boundary
    gb_ref = pyoracle.GBufferHost(W, H)
    gb = api.GBuffer()
    gb.create(W, H)
    dev = api.DevScene()
    dev.ctx = gpu_ctx
    res = [np.zeros(n, L.RESERVOIR_DTYPE) for _ in range(3)]
    ref_img = np.zeros((n, 3), np.float32)
    img = torch.zeros(n, 3, device="cuda")
    gpu_ctx.restir_init()
    for f in range(2):
        o.gbuffer_render(cam, gb_ref)
        gb.render(dev, cam)
        cur = gb.frameIdx
        assert_bit_equal(gb.albedo.cpu().numpy(), gb_ref.albedo, "gbuffer albedo")
        assert_bit_equal(gb.normal[cur].cpu().numpy(), gb_ref.normal[cur], "gbuffer normal")
        assert np.array_equal(gb.primId[cur].cpu().numpy(), gb_ref.primId[cur])
        o.restir_direct(cam, ref_img, 0, 80 + f, res[0], res[1], res[2], gb_ref, f == 0, 3, 1)
        res[0], res[1] = res[1], res[0]
        gpu_ctx.restir_direct(img, 0, 80 + f, gb.c_struct(cam), 3)
        assert_bit_equal(img.cpu().numpy(), ref_img, f"ReSTIR frame {f} (env={env})")
        assert gpu_ctx.restir_read(1).tobytes() == res[1].tobytes()
        gb_ref.update(cam)
        gb.update(cam)
    gpu_ctx.restir_free()


# ---------------------------------------------------------------------------------------------------------------------
# Display path: copyImageToPBO's four overloads (src/pathtrace.cu:32-147) — RGBA8 bytes equal the oracle's
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", [(37, 29), (1920, 1080)])
def test_copy_image_to_pbo_bit_exact(gpu_ctx, size):
    from oracle import pyoracle

    torch = _torch()
    W, H = size
    n = W * H
    rng = np.random.default_rng(7)
    img = (rng.random((n, 3)).astype(np.float32) ** 3) * 6.0
    special = np.array([0.0, -0.0, -1.5, np.nan, np.inf, -np.inf, 1e-42, 1.0, 255.0, 1e30], np.float32)
    img[: len(special), 0] = special
    img[: len(special), 1] = special[::-1]
    d_img = _dev(img)
    pbo = torch.full((n, 4), 77, dtype=torch.uint8, device="cuda")
    for tone in (0, 1, 2):
        for scale in (1.0, 0.37):
            gpu_ctx.copy_image_to_pbo(pbo, d_img, W, H, 0, tone, scale)
            ref = pyoracle.copy_image_to_pbo(img, W, H, 0, tone, scale)
            assert np.array_equal(pbo.cpu().numpy(), ref), f"vec3 tone={tone} scale={scale}"
    rg = rng.random((n, 2)).astype(np.float32) * 1.5 - 0.2
    gpu_ctx.copy_image_to_pbo(pbo, _dev(rg), W, H, 1)
    assert np.array_equal(pbo.cpu().numpy(), pyoracle.copy_image_to_pbo(rg, W, H, 1))
    g = rng.random(n).astype(np.float32) * 3.0
    gpu_ctx.copy_image_to_pbo(pbo, _dev(g), W, H, 2)
    assert np.array_equal(pbo.cpu().numpy(), pyoracle.copy_image_to_pbo(g, W, H, 2))
    mv = rng.integers(-1, n, n).astype(np.int32)
    gpu_ctx.copy_image_to_pbo(pbo, _dev(mv), W, H, 3)
    assert np.array_equal(pbo.cpu().numpy(), pyoracle.copy_image_to_pbo(mv, W, H, 3))
    with pytest.raises(Exception):
        gpu_ctx.copy_image_to_pbo(pbo, d_img, W, H, 4)


def test_scene_file_renders_bit_exact(gpu_ctx, tmp_path):
    """A Radish scene text file (OBJ meshes, PNG texture, HDR environment map) through rdh_scene_parse → rdh_scene_upload →
    pathTrace / pathTraceDirect: the pixels equal the oracle's on the same loaded arrays."""
    from PIL import Image

    from radish_pt_amd import api, scenes
    from test_scene_loader import CUBE_QUADS, PLANE, write_hdr

    torch = _torch()
    rng = np.random.default_rng(11)
    (tmp_path / "cube.obj").write_text(CUBE_QUADS)
    (tmp_path / "plane.obj").write_text(PLANE)
    Image.fromarray(rng.integers(30, 256, (8, 8, 3), dtype=np.uint8)).save(tmp_path / "albedo.png")
    write_hdr(tmp_path / "sky.hdr", (rng.random((4, 8, 3)) * 2.0 + 0.1).astype(np.float32))
    (tmp_path / "scene.txt").write_text(
        "Material floor\nType Lambertian\nBaseColor albedo.png\nMetallic 0\nRoughness 1\nIor 1.5\nNormalMap Null\n\n"
        "Material steel\nType MetallicWorkflow\nBaseColor 0.9 0.8 0.7\nMetallic 0.9\nRoughness 0.3\nIor 1.5\nNormalMap Null\n\n"
        "Material glass\nType Dielectric\nBaseColor 1 1 1\nMetallic 0\nRoughness 0\nIor 1.5\nNormalMap Null\n\n"
        "Material lamp\nType Light\nBaseColor 12 11 9\nMetallic 0\nRoughness 1\nIor 1\nNormalMap Null\n\n"
        "Object 0\nplane.obj\nMaterial floor\n\n"
        "Object 1\ncube.obj\nMaterial steel\nTranslate -1.2 0 -0.5\nRotate 0 30 0\nScale 0.8 1.2 0.8\n\n"
        "Object 2\ncube.obj\nMaterial glass\nTranslate 0.3 0 0.2\nScale 0.7 0.7 0.7\n\n"
        "Object 3\ncube.obj\nMaterial lamp\nTranslate -0.5 2.5 -0.5\nScale 1 0.02 1\n\n"
        "Camera\nResolution 48 36\nFovY 19.5\nLensRadius 0\nFocalDist 5\nApertureMask Null\nSample 2\nDepth 5\nFile out\n"
        "Eye 0.2 1.6 6\nRotation -91 -10 0\nUp 0 1 0\n\nEnvMap sky.hdr\n")
    sd, cam, settings = scenes.load_scene_file(str(tmp_path / "scene.txt"))
    W, H = (int(v) for v in cam["resolution"])
    n, depth = W * H, settings["trace_depth"]
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    gpu_ctx.set_camera(cam)
    o = _oracle(sd)
    ref_d, ref_i = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    d, i = torch.zeros(n, 3, device="cuda"), torch.zeros(n, 3, device="cuda")
    for it in range(settings["iterations"]):
        o.path_trace(cam, ref_d, ref_i, it, 5 + it, depth)
        gpu_ctx.path_trace(d, i, it, 5 + it, depth, api.RDH_PT_PERSISTENT)
    assert_bit_equal(d.cpu().numpy(), ref_d, "scene file: direct")
    assert_bit_equal(i.cpu().numpy(), ref_i, "scene file: indirect")
    assert ref_i.max() > 0 and (ref_d > 0).mean() > 0.3
    ref = np.zeros((n, 3), np.float32)
    o.path_trace_direct(cam, ref, 0, 9)
    dd = torch.zeros(n, 3, device="cuda")
    gpu_ctx.path_trace_direct(dd, 0, 9)
    assert_bit_equal(dd.cpu().numpy(), ref, "scene file: pathTraceDirect")


# ---------------------------------------------------------------------------------------------------------------------
# Denoisers (src/denoiser.cu): every kernel and both filter classes, bit-exact against the oracle on a rendered G-buffer
# ---------------------------------------------------------------------------------------------------------------------
def test_denoisers_bit_exact(gpu_ctx):
    from oracle import pyoracle
    from radish_pt_amd import api, hostlib, scenes

    torch = _torch()
    sd = scenes.teapots(segments=12, bands=8, grid=2, emissive_grid=(4, 8))
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    o = _oracle(sd)
    W, H = 70, 45  # not multiples of the 32x8 workgroup footprint
    n = W * H
    cams = [hostlib.make_camera(W, H, eye=(0.3 + 0.07 * f, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0) for f in range(3)]
    gb_ref = pyoracle.GBufferHost(W, H)
    gb = api.GBuffer()
    gb.create(W, H)
    dev = api.DevScene()
    dev.ctx = gpu_ctx
    scene = api.Scene.__new__(api.Scene)  # the mirror's globals: State.scene.devScene.ctx / .camera
    scene.devScene, scene.camera, scene.data = dev, cams[0], sd
    api.State.scene = scene
    rng = np.random.default_rng(2)
    eaw = api.LeveledEAWFilter()
    eaw.create(W, H, 5)
    stf = api.SpatioTemporalFilter()
    stf.create(W, H, 5)
    # the oracle's copy of SpatioTemporalFilter's state
    accC = [np.zeros((n, 3), np.float32) for _ in range(2)]
    accM = [np.zeros((n, 3), np.float32) for _ in range(2)]
    first, fidx = True, 0
    for f, cam in enumerate(cams):
        scene.camera = cam
        o.gbuffer_render(cam, gb_ref)
        gb.render(dev, cam)
        noisy = (rng.random((n, 3)).astype(np.float32) ** 2) * 2.0
        noisy[rng.integers(0, n, 5)] = 40.0  # fireflies
        d_noisy = _dev(noisy)
        # ---- LeveledEAWFilter::filter: five a-trous levels (denoiser.cu:419-434) ----
        ref = pyoracle.denoise_eaw(noisy, gb_ref, cam, 64.0, 0.2, 1.0, 0)
        for lv in (1, 2, 3, 4):
            ref = pyoracle.denoise_eaw(ref, gb_ref, cam, 64.0, 0.2, 1.0, lv)
        out = eaw.filter(torch.zeros(n, 3, device="cuda"), d_noisy, gb, cam)
        assert_bit_equal(out.cpu().numpy(), ref, f"LeveledEAWFilter frame {f}")
        assert np.abs(ref - noisy).max() > 0.1  # it really filtered
        # ---- SpatioTemporalFilter::filter (denoiser.cu:525-558) ----
        accC[fidx], accM[fidx] = pyoracle.denoise_temporal_accumulate(accC[fidx ^ 1], accM[fidx ^ 1], noisy, gb_ref, first)
        first = False
        var = pyoracle.denoise_estimate_variance(accM[fidx], W, H)
        fvar = pyoracle.denoise_filter_variance(var, W, H)
        colorOut, tmpVar = pyoracle.denoise_svgf(accC[fidx], var, fvar, gb_ref, cam, 4.0, 128.0, 1.0, 0)
        colorOut, accC[fidx] = accC[fidx], colorOut  # std::swap(colorOut, accumColor[frameIdx])
        var = tmpVar
        fvar = pyoracle.denoise_filter_variance(var, W, H)
        colorOut, var = pyoracle.denoise_svgf(accC[fidx], var, fvar, gb_ref, cam, 4.0, 128.0, 1.0, 1)
        for lv in (2, 3, 4):
            fvar = pyoracle.denoise_filter_variance(var, W, H)
            colorOut, var = pyoracle.denoise_svgf(colorOut, var, fvar, gb_ref, cam, 4.0, 128.0, 1.0, lv)
        got = stf.filter(torch.zeros(n, 3, device="cuda"), d_noisy, gb, cam)
        assert_bit_equal(got.cpu().numpy(), colorOut, f"SpatioTemporalFilter frame {f}")
        assert_bit_equal(stf.variance.cpu().numpy(), var, f"SVGF variance frame {f}")
        assert_bit_equal(stf.accumColor[stf.frameIdx].cpu().numpy(), accC[fidx], f"accumColor frame {f}")
        assert np.isfinite(colorOut).all()
        stf.nextFrame()
        fidx ^= 1
        # ---- modulateAlbedo + addImage ----
        img = torch.from_numpy(colorOut.copy()).cuda()
        api.modulateAlbedo(img, gb)
        mod = pyoracle.denoise_modulate(colorOut, gb_ref)
        assert_bit_equal(img.cpu().numpy(), mod, "modulate")
        api.addImage(img, d_noisy, W, H)
        assert_bit_equal(img.cpu().numpy(), pyoracle.denoise_add(mod, noisy, W, H), "addImage (two-image overload)")
        tri = torch.zeros(n, 3, device="cuda")
        api.addImage(tri, img, d_noisy, W, H)
        assert_bit_equal(tri.cpu().numpy(), pyoracle.denoise_add(img.cpu().numpy(), noisy, W, H), "addImage (three-image overload)")
        gb_ref.update(cam)
        gb.update(cam)
    api.State.scene = None


def test_mirror_frame_loop_like_runCuda():
    """The reference's frame loop (`runCuda`, /root/reference/src/main.cpp:163-202) written against the Python mirror of its
    API — gBuffer.render, ReSTIRDirect / pathTraceDirect / pathTrace with the global State.looper, copyImageToPBO,
    gBuffer.update — with an animated camera; every frame's image and RGBA8 preview equal the oracle driven the same way."""
    from oracle import pyoracle
    from radish_pt_amd import api, hostlib, layouts as L, scenes

    torch = _torch()
    sd = scenes.teapots(segments=12, bands=8, grid=2, emissive_grid=(4, 8))
    W, H = 56, 40
    n = W * H

    def camera_at(f):  # Settings::animateCamera: the eye moves, then Camera::update
        return hostlib.make_camera(W, H, eye=(0.3 + 0.06 * np.cos(0.4 * f), 1.9, 7.4 + 0.06 * np.sin(0.4 * f)),
                                   rotation=(-91.5, -11.0, 0.0), fovy=19.0)

    api.State.looper = 5
    api.State.scene = api.Scene(sd, camera_at(0))
    api.Settings.traceDepth = 3
    api.Settings.reservoirReuse = api.ReservoirReuse.TemporalSpatial
    api.Settings.ptFlags = api.RDH_PT_PERSISTENT
    o = _oracle(sd)
    looper = 5
    try:
        api.pathTraceInit()
        api.ReSTIRInit()
        gb, gb_ref = api.GBuffer(), pyoracle.GBufferHost(W, H)
        gb.create(W, H)
        direct, indirect = torch.zeros(n, 3, device="cuda"), torch.zeros(n, 3, device="cuda")
        pbo = torch.zeros(n, 4, dtype=torch.uint8, device="cuda")
        res = [np.zeros(n, L.RESERVOIR_DTYPE) for _ in range(3)]
        ref_d, ref_i = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        for f in range(4):
            cam = camera_at(f)
            api.State.scene.camera = cam
            iteration = 0  # camChanged: runCuda resets it every frame
            gb.render(api.State.scene.devScene, cam)
            o.gbuffer_render(cam, gb_ref)
            mode = ("restir", "direct", "pt", "restir")[f]
            if mode == "restir":
                api.ReSTIRDirect(direct, iteration, gb)
                o.restir_direct(cam, ref_d, iteration, looper, res[0], res[1], res[2], gb_ref, f == 0, 3, 1)
                res[0], res[1] = res[1], res[0]
            elif mode == "direct":
                api.pathTraceDirect(direct, iteration)
                o.path_trace_direct(cam, ref_d, iteration, looper)
            else:
                api.pathTrace(direct, indirect, iteration)
                o.path_trace(cam, ref_d, ref_i, iteration, looper, 3)
                assert_bit_equal(indirect.cpu().numpy(), ref_i, f"frame {f} indirect")
            looper = (looper + 1) % 10000
            assert api.State.looper == looper
            assert_bit_equal(direct.cpu().numpy(), ref_d, f"frame {f} ({mode}) direct")
            api.copyImageToPBO(pbo, direct, W, H, api.ToneMapping.ACES)
            assert np.array_equal(pbo.cpu().numpy(), pyoracle.copy_image_to_pbo(ref_d, W, H, 0, 2, 1.0)), f"frame {f} preview"
            gb.update(cam)
            gb_ref.update(cam)
        api.ReSTIRFree()
        api.pathTraceFree()
    finally:
        api.State.scene.clear()
        api.State.scene = None
        api.State.looper = 0


@pytest.mark.parametrize("n_stack,expect_pairs", [(1500, True), (2300, False)])
def test_chain_shaped_tree_ties_deep_stacks_and_the_threaded_fallback(gpu_ctx, n_stack, expect_pairs):
    """n coincident triangles: the SAH builder has nothing to split on and returns a CHAIN (one leaf and one inner child per level,
    n - 1 levels), and every ray that hits one of them hits all of them at the SAME distance — the strict `<` of DevScene::intersect
    (scene.h:285) decides by visiting order, which differs between the six orderings.  The materials alternate, so a wrong winner
    shows in the image.  1 500 levels: the sibling-pair walks keep up to that many pending far children per lane when they count
    (failed far children are stacked too) — an LDS ring of 8 entries, the rest in the global strip (traverse.h, pairPush).  2 300
    levels is past the depth the strips are sized for (radish_hip.hip, kMaxPairDepth): the scene silently keeps the threaded walk,
    and asking for pairs by flag is refused.  Either way: hits, images and work counters equal the oracle's."""
    from radish_pt_amd import api, hostlib, layouts as L, scenes

    torch = _torch()
    mats = [L.make_material(L.LAMBERTIAN, (0.8, 0.3, 0.2)), L.make_material(L.LAMBERTIAN, (0.2, 0.7, 0.9)),
            L.make_material(L.METALLIC_WORKFLOW, (0.9, 0.9, 0.9), metallic=0.6, roughness=0.35), L.make_material(L.LIGHT, (7.0, 6.5, 6.0))]
    tri = np.array([[-1.2, -0.4, 0.1], [1.3, -0.5, -0.2], [0.1, 1.4, 0.3]], np.float32)
    nrm = np.cross(tri[1] - tri[0], tri[2] - tri[0])
    nrm = np.tile((nrm / np.linalg.norm(nrm)).astype(np.float32), (3, 1))
    uv = np.array([[0, 0], [1, 0], [0, 1]], np.float32)
    floor = np.array([[-3, -0.8, -3], [3, -0.8, -3], [3, -0.8, 3], [-3, -0.8, -3], [3, -0.8, 3], [-3, -0.8, 3]], np.float32)
    lamp = np.array([[-0.6, 2.4, -0.6], [0.6, 2.4, -0.6], [0.6, 2.4, 0.6], [-0.6, 2.4, -0.6], [0.6, 2.4, 0.6], [-0.6, 2.4, 0.6]], np.float32)  # facing down
    v = np.concatenate([np.tile(tri, (n_stack, 1)), floor, lamp])
    n = np.concatenate([np.tile(nrm, (n_stack, 1)), np.tile(np.array([0, 1, 0], np.float32), (6, 1)), np.tile(np.array([0, -1, 0], np.float32), (6, 1))])
    t = np.tile(uv, (n_stack + 4, 1))
    ids = np.array([i % 3 for i in range(n_stack)] + [0, 0, 3, 3], np.int32)
    sd = scenes.SceneData(f"chain{n_stack}", v, n, t, ids, np.array(mats, dtype=L.MATERIAL_DTYPE))
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    o = _oracle(sd)
    # ---- ray batches: closest hit and any hit, every walker, equal counters ----
    rays = random_rays(1536, seed=41)
    rays[:, :3] *= 0.6
    seg = random_segments(1536, seed=43)
    ref_h = o.trace_closest(rays)
    st_h = o.stats()
    o.reset_stats()
    ref_o = o.trace_occluded(seg)
    st_o = o.stats()
    assert 0.1 < (ref_h["primId"] >= 0).mean() and len(set(ref_h["primId"][ref_h["primId"] >= 0] % 3)) >= 2  # ties resolved both ways
    walkers = [0, api.RDH_PT_PERSISTENT | api.RDH_PT_NO_PAIRS, api.RDH_PT_PERSISTENT] + ([api.RDH_PT_PERSISTENT | api.RDH_PT_PAIRS] if expect_pairs else [])
    for flags in walkers:
        d_hits = torch.full((len(rays), 4), 0x7fffffff, dtype=torch.int32, device="cuda")
        gpu_ctx.counters_reset()
        gpu_ctx.trace_closest(_dev(rays), d_hits, api.RDH_PT_COUNT | flags)
        got = d_hits.cpu().numpy().view(L.HIT_DTYPE).reshape(-1)
        assert np.array_equal(got["primId"], ref_h["primId"]), flags
        for f in ("u", "v", "t"):
            assert_bit_equal(got[f], ref_h[f], f"hit.{f} flags={flags}")
        ct = gpu_ctx.counters()
        assert ct["nodeVisits"] == st_h["nodeVisits"] and ct["triTests"] == st_h["triTests"], flags
        gpu_ctx.trace_closest(_dev(rays), d_hits, flags)  # and without the counters (failed far children are not stacked then)
        assert np.array_equal(d_hits.cpu().numpy().view(L.HIT_DTYPE).reshape(-1)["primId"], ref_h["primId"]), flags
        d_occ = torch.full((len(seg),), -7, dtype=torch.int32, device="cuda")
        gpu_ctx.counters_reset()
        gpu_ctx.trace_occluded(_dev(seg), d_occ, api.RDH_PT_COUNT | flags)
        assert np.array_equal(d_occ.cpu().numpy(), ref_o), flags
        ct = gpu_ctx.counters()
        assert ct["nodeVisits"] == st_o["nodeVisits"] and ct["triTests"] == st_o["triTests"], flags
    if not expect_pairs:
        with pytest.raises(api.RadishError):
            gpu_ctx.trace_closest(_dev(rays), torch.zeros(len(rays), 4, dtype=torch.int32, device="cuda"), api.RDH_PT_PERSISTENT | api.RDH_PT_PAIRS)
    # ---- whole frames through every structure ----
    W, H = 48, 32
    npx = W * H
    cam = hostlib.make_camera(W, H, eye=(0.2, 0.7, 4.5), rotation=(-90.0, 0.0, 0.0), fovy=19.5)
    gpu_ctx.set_camera(cam)
    o.reset_stats()
    ref_d, ref_i = np.zeros((npx, 3), np.float32), np.zeros((npx, 3), np.float32)
    o.path_trace(cam, ref_d, ref_i, 0, 5, 4)
    st = o.stats()
    for flags in (api.RDH_PT_PERSISTENT, api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL, api.RDH_PT_WAVEFRONT | api.RDH_PT_WF_SUBFRAMES, api.RDH_PT_AUTO):
        d, i = torch.zeros(npx, 3, device="cuda"), torch.zeros(npx, 3, device="cuda")
        gpu_ctx.counters_reset()
        gpu_ctx.path_trace(d, i, 0, 5, 4, flags | api.RDH_PT_COUNT)
        assert_bit_equal(d.cpu().numpy(), ref_d, f"chain direct flags={flags}")
        assert_bit_equal(i.cpu().numpy(), ref_i, f"chain indirect flags={flags}")
        assert gpu_ctx.counters() == st, flags
        d.zero_(); i.zero_()
        gpu_ctx.path_trace(d, i, 0, 5, 4, flags)
        assert_bit_equal(d.cpu().numpy(), ref_d, f"chain direct flags={flags} (no counters)")
        assert_bit_equal(i.cpu().numpy(), ref_i, f"chain indirect flags={flags} (no counters)")
    assert (ref_d > 0).any() and (ref_i > 0).any()


@pytest.mark.parametrize("case", ["one_triangle", "no_lights", "only_lights"])
def test_degenerate_scenes_bit_exact(gpu_ctx, case):
    """Smallest inputs: a single triangle (BVH of one node), a scene without emitters (NEE and RIS have nothing to pick: the
    alias table is empty), a scene of emitters only.  Every kernel family still equals the oracle."""
    from oracle import pyoracle
    from radish_pt_amd import api, hostlib, layouts as L, scenes

    torch = _torch()
    white = L.make_material(L.LAMBERTIAN, (0.8, 0.7, 0.6))
    lamp = L.make_material(L.LIGHT, (6.0, 5.0, 4.0))
    tri = np.array([[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]], np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (3, 1))
    uv = np.array([[0, 0], [1, 0], [0, 1]], np.float32)
    if case == "one_triangle":
        v, n, t, ids, mats = tri, nrm, uv, [0], [white]
    else:
        quad = np.array([[-2, -0.5, -2], [2, -0.5, -2], [2, -0.5, 2], [-2, -0.5, -2], [2, -0.5, 2], [-2, -0.5, 2]], np.float32)
        v = np.concatenate([tri, tri + np.array([0.3, 0.2, -0.7], np.float32), quad])
        n = np.concatenate([nrm, nrm, np.tile(np.array([0, 1, 0], np.float32), (6, 1))])
        t = np.concatenate([uv, uv, uv, uv])
        ids = [0, 0, 0, 0] if case == "no_lights" else [1, 1, 1, 1]
        mats = [white, lamp]
    sd = scenes.SceneData(case, v, n, t, np.array(ids, np.int32), np.array(mats, dtype=L.MATERIAL_DTYPE))
    assert sd.num_lights == (0 if case != "only_lights" else 4)
    W, H = 40, 24
    npx = W * H
    cam = hostlib.make_camera(W, H, eye=(0.1, 0.6, 4.0), rotation=(-90.0, 0.0, 0.0), fovy=19.5)
    gpu_ctx.set_partition(0, 1, 64)
    gpu_ctx.upload_scene(sd)
    gpu_ctx.set_camera(cam)
    o = _oracle(sd)
    ref_d, ref_i = np.zeros((npx, 3), np.float32), np.zeros((npx, 3), np.float32)
    for it in range(2):
        o.path_trace(cam, ref_d, ref_i, it, 3 + it, 4)
    st = o.stats()
    for flags in (api.RDH_PT_PERSISTENT, api.RDH_PT_MEGAKERNEL, api.RDH_PT_WAVEFRONT | api.RDH_PT_SORT_MATERIAL):
        d, i = torch.zeros(npx, 3, device="cuda"), torch.zeros(npx, 3, device="cuda")
        gpu_ctx.counters_reset()
        for it in range(2):
            gpu_ctx.path_trace(d, i, it, 3 + it, 4, flags | api.RDH_PT_COUNT)
        assert_bit_equal(d.cpu().numpy(), ref_d, f"{case} direct flags={flags}")
        assert_bit_equal(i.cpu().numpy(), ref_i, f"{case} indirect flags={flags}")
        assert gpu_ctx.counters() == st
    ref = np.zeros((npx, 3), np.float32)
    o.path_trace_direct(cam, ref, 0, 11)
    dd = torch.zeros(npx, 3, device="cuda")
    gpu_ctx.path_trace_direct(dd, 0, 11)
    assert_bit_equal(dd.cpu().numpy(), ref, f"{case} pathTraceDirect")
    gb_ref = pyoracle.GBufferHost(W, H)
    gb = api.GBuffer()
    gb.create(W, H)
    dev = api.DevScene()
    dev.ctx = gpu_ctx
    res = [np.zeros(npx, L.RESERVOIR_DTYPE) for _ in range(3)]
    ref_r = np.zeros((npx, 3), np.float32)
    img = torch.zeros(npx, 3, device="cuda")
    gpu_ctx.restir_init()
    for f in range(2):
        o.gbuffer_render(cam, gb_ref)
        gb.render(dev, cam)
        assert np.array_equal(gb.primId[gb.frameIdx].cpu().numpy(), gb_ref.primId[gb_ref.frameIdx])
        o.restir_direct(cam, ref_r, 0, 20 + f, res[0], res[1], res[2], gb_ref, f == 0, 3, 1)
        res[0], res[1] = res[1], res[0]
        gpu_ctx.restir_direct(img, 0, 20 + f, gb.c_struct(cam), 3)
        assert_bit_equal(img.cpu().numpy(), ref_r, f"{case} ReSTIR frame {f}")
        assert gpu_ctx.restir_read(1).tobytes() == res[1].tobytes()
        gb_ref.update(cam)
        gb.update(cam)
    gpu_ctx.restir_free()
    assert (ref_d > 0).any()


def test_error_behaviour(gpu_ctx, cornell_small):
    from radish_pt_amd import api, scenes

    torch = _torch()
    ctx = api.Context(0)
    img = torch.zeros(16, 3, device="cuda")
    with pytest.raises(api.RadishError):  # no scene yet
        ctx.path_trace(img, img, 0, 0, 4)
    ctx.upload_scene(cornell_small)
    ctx.set_camera(scenes.cornell_camera(4, 4))
    with pytest.raises(api.RadishError):  # 4 + 7*29 Sobol dims > 200
        ctx.path_trace(img, img, 0, 0, 29)
    with pytest.raises(api.RadishError):  # ReSTIR before init
        gb = api.GBuffer()
        gb.create(4, 4)
        ctx.restir_direct(img, 0, 0, gb.c_struct(scenes.cornell_camera(4, 4)), 3)
    ctx.close()
