"""Host-side logic on CPU: BVH builder, alias tables, light list, camera, tile partition, and that libradish_hip.so
loads and exports every symbol include/radish_hip.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hip_library_exports_every_declared_symbol():
    from radish_pt_amd import api

    header = open(os.path.join(ROOT, "include", "radish_hip.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|long long|void|const char \*|int32_t)\s*(rdh_[a-z_]+)\s*\(", header, re.M)))
    assert declared == sorted(api.EXPORTS), "api.EXPORTS is out of sync with include/radish_hip.h"
    lib = ctypes.CDLL(api.HIP_LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libradish_hip.so does not export {name}"


def test_host_library_exports():
    from radish_pt_amd import hostlib

    header = open(os.path.join(ROOT, "include", "radish_host.h")).read()
    names = sorted(set(re.findall(r"^(?:int|void|int32_t)\s*(rdh_[a-z_]+)\s*\(", header, re.M)))
    assert len(names) == 7
    for name in names:
        assert hasattr(hostlib.lib(), name)


def test_no_gpu_means_loud_failure():
    """The product path has no CPU fallback: without a HIP device creating a context raises."""
    import torch

    from radish_pt_amd import api

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(api.RadishError):
        api.Context(0)
    h = ctypes.c_void_p()
    assert api.lib().rdh_create(ctypes.byref(h), 0) == -4  # RDH_ERR_NO_DEVICE
    assert not h.value


def test_product_does_not_touch_the_oracle():
    """Nothing under radish_pt_amd/ or include/ may reference oracle/ (the oracle is test infrastructure)."""
    pat = re.compile(r"(?<![A-Za-z_.])(from\s+oracle|import\s+oracle|liboracle|orc_[a-z_]+\s*\(|[\"'/]oracle/)")
    for base in ("radish_pt_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    code = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith(("//", "#", "*", "/*")))
                    assert not pat.search(code), f"{base}/{f} references the oracle"


# ---------------------------------------------------------------------------------------------------------------------
# BVH builder (restates src/bvh.cpp)
# ---------------------------------------------------------------------------------------------------------------------
def _check_bvh(sd):
    n, size = sd.num_prims, sd.bvh_size
    assert size == 2 * n - 1
    v = sd.vertices.reshape(n, 3, 3)
    tri_lo, tri_hi = v.min(axis=1), v.max(axis=1)
    for k in range(6):
        nodes = sd.nodes[k]
        prim, box, nxt = nodes["primitiveId"], nodes["boundingBoxId"], nodes["nextNodeIfMiss"]
        # each array is a permutation of the depth-first boxes; every primitive is exactly one leaf
        assert sorted(box.tolist()) == list(range(size))
        leaves = prim[prim >= 0]
        assert sorted(leaves.tolist()) == list(range(n)) and (prim >= -1).all()
        # miss links: strictly forward, a leaf links to the next slot, the root links past the end
        idx = np.arange(size)
        assert (nxt > idx).all() and (nxt <= size).all() and nxt[0] == size
        assert (nxt[prim >= 0] == idx[prim >= 0] + 1).all()
        # subtree [i, next(i)) is contained in box(i); leaf boxes are the triangle bounds
        lo, hi = sd.boxes[box, :3], sd.boxes[box, 3:]
        leaf = prim >= 0
        assert np.array_equal(lo[leaf], tri_lo[prim[leaf]]) and np.array_equal(hi[leaf], tri_hi[prim[leaf]])
        for i in np.random.default_rng(k).choice(size, min(size, 300), replace=False):
            sub = slice(i, nxt[i])
            assert (lo[sub] >= lo[i]).all() and (hi[sub] <= hi[i]).all()
            assert nxt[sub].max() <= nxt[i]
        # inner node i has children i+1 and next(i+1); the one visited first is the nearer along axis k//2
        inner = np.where(~leaf)[0][:200]
        for i in inner:
            a, b = i + 1, nxt[i + 1]
            ca = (lo[a] + hi[a]) * np.float32(0.5)
            cb = (lo[b] + hi[b]) * np.float32(0.5)
            d = k // 2
            if ca[d] != cb[d]:
                first_is_greater = ca[d] > cb[d]
                assert first_is_greater == (k % 2 == 0)  # array 0/2/4 serve rays travelling toward -axis


def test_bvh_invariants(cornell_small, tiny_scene):
    _check_bvh(cornell_small)
    _check_bvh(tiny_scene)


def test_bvh_two_triangles_by_hand():
    """Two triangles side by side in x: root + two leaves; ordering arrays 0/1 differ in which leaf comes first."""
    from radish_pt_amd import hostlib

    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 0, 0], [3, 0, 0], [2, 1, 0]], np.float32)
    boxes, nodes = hostlib.build_bvh(v)
    assert boxes.shape == (3, 6)
    np.testing.assert_array_equal(boxes[0], [0, 0, 0, 3, 1, 0])
    np.testing.assert_array_equal(boxes[1], [0, 0, 0, 1, 1, 0])  # left = smaller centroid (bucket 0)
    np.testing.assert_array_equal(boxes[2], [2, 0, 0, 3, 1, 0])
    # array 0 (rays with direction.x < 0): visit the larger-x child (box 2, prim 1) first
    assert nodes[0].tolist() == [(-1, 0, 3), (1, 2, 2), (0, 1, 3)]
    # array 1 (rays with direction.x > 0): smaller-x child first
    assert nodes[1].tolist() == [(-1, 0, 3), (0, 1, 2), (1, 2, 3)]
    # y/z arrays: centres tie on those axes → no swap for even arrays (a < b false), swap for odd ones
    assert nodes[2].tolist() == nodes[4].tolist() == [(-1, 0, 3), (0, 1, 2), (1, 2, 3)]
    assert nodes[3].tolist() == nodes[5].tolist() == [(-1, 0, 3), (1, 2, 2), (0, 1, 3)]


def test_bvh_identical_centroids():
    """A quad's two triangles share one AABB centre → 0/0 bucket; the builder must still split them."""
    from radish_pt_amd import hostlib

    q = np.array([[0, 0, 0], [1, 0, 0], [1, 0, 1], [0, 0, 0], [1, 0, 1], [0, 0, 1]], np.float32)
    boxes, nodes = hostlib.build_bvh(q)
    assert sorted(nodes[0]["primitiveId"].tolist()) == [-1, 0, 1]
    four = np.concatenate([q, q])  # four triangles, all the same centre
    boxes, nodes = hostlib.build_bvh(four)
    assert sorted(nodes[3]["primitiveId"][nodes[3]["primitiveId"] >= 0].tolist()) == [0, 1, 2, 3]


# ---------------------------------------------------------------------------------------------------------------------
# alias table (restates src/sampler.h:81-125) and light list (src/scene.cpp:192-223)
# ---------------------------------------------------------------------------------------------------------------------
def test_alias_table_reproduces_distribution():
    from radish_pt_amd import hostlib

    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 64, 1000):
        w = rng.uniform(0.01, 5.0, n).astype(np.float32)
        w[rng.integers(0, n)] *= 20
        table, s = hostlib.build_alias_table(w)
        assert abs(float(s) - float(w.sum(dtype=np.float64))) < 1e-3 * float(s)
        assert (table["failId"] >= 0).all() and (table["failId"] < n).all()
        # probability of outcome i = (prob_i + sum over j with failId_j == i of (1 - prob_j)) / n
        p = table["prob"].astype(np.float64).clip(0, 1)
        got = p.copy()
        np.add.at(got, table["failId"], 1 - p)
        got /= n
        np.testing.assert_allclose(got, w / w.sum(dtype=np.float64), atol=2e-6, rtol=2e-4)


def test_alias_table_by_hand():
    from radish_pt_amd import hostlib

    table, s = hostlib.build_alias_table(np.array([1.0, 3.0], np.float32))
    # normalised to mean 1: [0.5, 1.5] → entry 0 keeps 0.5 and falls to 1; entry 1 keeps 1.0
    assert s == 4.0 and table.tolist() == [(0.5, 1), (1.0, 1)]


def test_light_list(cornell_small):
    sd = cornell_small
    assert sd.num_lights == 2
    assert (sd.materials["type"][sd.material_ids[sd.light_prim_ids]] == 4).all()
    # each light triangle is half of the 0.5 x 0.5 quad: area 0.125; power = luminance * 2*pi * area
    lum = 0.2126 * 17 + 0.7152 * 12 + 0.0722 * 4
    np.testing.assert_allclose(sd.light_power, [lum * 2 * np.pi * 0.125] * 2, rtol=1e-6)
    np.testing.assert_allclose(sd.sum_light_power_inv, 1 / (2 * lum * 2 * np.pi * 0.125), rtol=1e-6)
    np.testing.assert_array_equal(sd.light_unit_radiance, [[17, 12, 4]] * 2)


def test_camera_update():
    from radish_pt_amd import hostlib

    cam = hostlib.make_camera(640, 480, eye=(1, 2, 3), rotation=(-90, 0, 0), fovy=30.0)
    np.testing.assert_allclose(cam["view"], [0, 0, -1], atol=1e-6)
    np.testing.assert_allclose(cam["right"], [1, 0, 0], atol=1e-6)
    np.testing.assert_allclose(cam["up"], [0, 1, 0], atol=1e-6)
    # rotationMatInv * [right up view] = I
    M = np.stack([cam["right"], cam["up"], cam["view"]], axis=1).astype(np.float64)
    Minv = cam["rotationMatInv"].reshape(3, 3).T.astype(np.float64)  # column-major storage
    np.testing.assert_allclose(Minv @ M, np.eye(3), atol=1e-6)
    np.testing.assert_allclose(cam["tanFovY"], np.tan(np.radians(15.0)), rtol=1e-6)
    np.testing.assert_allclose(cam["fov"][0], np.degrees(np.arctan(np.tan(np.radians(30.0)) * 640 / 480)), rtol=1e-6)
    pitched = hostlib.make_camera(64, 64, eye=(0, 0, 0), rotation=(0, 30, 0), fovy=20.0)
    np.testing.assert_allclose(pitched["view"], [np.cos(np.radians(30)), np.sin(np.radians(30)), 0], atol=1e-6)


def test_sobol_table_matches_joe_kuo_first_points():
    from radish_pt_amd import scenes

    t = scenes.sobol_table()
    assert t.shape == (10000, 200) and t.dtype == np.uint32
    assert (t[0] == 0).all() and (t[1] == 0x80000000).all()
    assert t[2, 0] == 0xC0000000 and t[2, 1] == 0x40000000 and t[3, 0] == 0x40000000 and t[3, 1] == 0xC0000000
    # dimension 0 is the van der Corput sequence in Gray-code order: first 2^k points are a permutation of j / 2^k
    for k in (4, 8, 12):
        assert sorted((t[: 1 << k, 0] >> (32 - k)).tolist()) == list(range(1 << k))
    # every dimension is a (0,1)-sequence in base 2 as well
    assert sorted((t[:256, 57] >> 24).tolist()) == list(range(256))


# ---------------------------------------------------------------------------------------------------------------------
# tile partition arithmetic
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H,world,tile", [(200, 120, 2, 64), (1920, 1080, 8, 64), (37, 29, 3, 8), (64, 64, 1, 64)])
def test_partition_is_a_bijection(W, H, world, tile):
    from radish_pt_amd import partition

    shard = partition.shard_elems(W, H, world, tile)
    seen = np.zeros(W * H, np.int32)
    gathered = np.full(world * shard, -1, np.int64)
    for r in range(world):
        frame_idx, packed_idx = partition.rank_pixels(W, H, r, world, tile)
        seen[frame_idx] += 1
        assert packed_idx.max(initial=0) < shard
        gathered[r * shard + packed_idx] = frame_idx
    assert (seen == 1).all()
    assert np.array_equal(gathered[partition.untile_indices(W, H, world, tile)], np.arange(W * H))
    # balance of the interleaved assignment at the benchmark size: within 3 % of the mean pixel count
    if (W, H, world) == (1920, 1080, 8):
        counts = [len(partition.rank_pixels(W, H, r, world, tile)[0]) for r in range(world)]
        assert max(counts) <= 1.03 * (W * H / world)
