#!/usr/bin/env python3
"""Writes tests/golden/cornell_small.npz: inputs AND expected outputs of the hot path on a small scene.

The reference cannot be built or run here and ships no fixtures, so these vectors are produced by
this repository's CPU oracle (oracle/liboracle.so) — they pin the oracle and the HIP path against regressions and
against each other; they are not outputs of the reference ("parity unpinned").  The scene arrays themselves are stored
(not regenerated) so the fixture does not depend on numpy's sin/cos bits on the machine that runs the tests.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from helpers import random_rays, random_segments
    from oracle import pyoracle
    from radish_pt_amd import layouts as L, scenes

    sd = scenes.cornell(segments=10, bands=8)  # 10 + 2 + 3*160 = 492 triangles
    W, H = 48, 36
    cam = scenes.cornell_camera(W, H)
    o = pyoracle.OracleScene(sd)
    n = W * H
    out = dict(
        vertices=sd.vertices, normals=sd.normals, texcoords=sd.texcoords, material_ids=sd.material_ids,
        materials=np.frombuffer(sd.materials.tobytes(), np.uint8), camera=np.frombuffer(cam.tobytes(), np.uint8),
        bvh_sha256=np.frombuffer(
            hashlib.sha256(sd.boxes.tobytes() + b"".join(a.tobytes() for a in sd.nodes)).digest(), np.uint8),
        light_sampler=np.frombuffer(sd.light_sampler.tobytes(), np.uint8),
    )
    rays = random_rays(2048, 21)
    seg = random_segments(2048, 22)
    out["rays"], out["segments"] = rays, seg
    out["hits"] = np.frombuffer(o.trace_closest(rays).tobytes(), np.uint8)
    out["occluded"] = o.trace_occluded(seg)
    d = np.zeros((n, 3), np.float32)
    i = np.zeros((n, 3), np.float32)
    for it in range(4):  # 4 spp, depth 4, loopers 0..3
        o.path_trace(cam, d, i, it, it, 4)
    out["pt_direct"], out["pt_indirect"] = d, i
    dd = np.zeros((n, 3), np.float32)
    o.path_trace_direct(cam, dd, 0, 9)
    out["ptd_direct"] = dd
    gb = pyoracle.GBufferHost(W, H)
    res = [np.zeros(n, L.RESERVOIR_DTYPE) for _ in range(3)]
    img = np.zeros((n, 3), np.float32)
    for f in range(3):  # ReSTIR DI, temporal + spatial, faithful RIS, loopers 30..32
        o.gbuffer_render(cam, gb)
        o.restir_direct(cam, img, 0, 30 + f, res[0], res[1], res[2], gb, f == 0, 3, 1)
        res[0], res[1] = res[1], res[0]
        gb.update(cam)
    out["restir_direct"] = img
    out["restir_reservoirs"] = np.frombuffer(res[1].tobytes(), np.uint8)
    cur = gb.frameIdx ^ 1
    out["gb_albedo"], out["gb_normal"], out["gb_depth"] = gb.albedo, gb.normal[cur], gb.depth[cur]
    out["gb_primId"], out["gb_motion"] = gb.primId[cur], gb.motion
    path = os.path.join(ROOT, "tests", "golden", "cornell_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
