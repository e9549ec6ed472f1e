#!/usr/bin/env python3
"""Writes tests/golden/post_small.npz: inputs AND expected outputs of the steps after the hot path — display
(sendImageToPBO, src/pathtrace.cu:32-118) and the denoisers (src/denoiser.cu) — on a 40x30 G-buffer of the small Cornell
scene.  Produced by this repository's CPU oracle (the reference cannot be built or run here and ships no fixtures:
parity unpinned); pins the oracle and the HIP kernels against regressions and against each other.

Run from the repo root:  python tests/golden/make_golden_post.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from oracle import pyoracle
    from radish_pt_amd import hostlib, scenes

    sd = scenes.cornell(segments=10, bands=8)
    W, H = 40, 30
    n = W * H
    o = pyoracle.OracleScene(sd)
    cams = [hostlib.make_camera(W, H, eye=(0.05 * f, 1.0, 4.2), rotation=(-90.0, 0.0, 0.0), fovy=19.5) for f in range(2)]
    gb = pyoracle.GBufferHost(W, H)
    rng = np.random.default_rng(17)
    out = {}
    accC = [np.zeros((n, 3), np.float32) for _ in range(2)]
    accM = [np.zeros((n, 3), np.float32) for _ in range(2)]
    for f, cam in enumerate(cams):
        o.gbuffer_render(cam, gb)
        cur = gb.frameIdx
        noisy = (rng.random((n, 3)).astype(np.float32) ** 2) * 1.5
        noisy[rng.integers(0, n, 3)] = 25.0
        out[f"camera{f}"] = np.frombuffer(cam.tobytes(), np.uint8)
        out[f"noisy{f}"] = noisy
        for name, arr in (("albedo", gb.albedo), ("normal", gb.normal[cur]), ("depth", gb.depth[cur]), ("primId", gb.primId[cur]),
                          ("motion", gb.motion)):
            out[f"gb{f}_{name}"] = arr.copy()
        ref = pyoracle.denoise_eaw(noisy, gb, cam, 64.0, 0.2, 1.0, 0)
        for lv in (1, 2, 3, 4):
            ref = pyoracle.denoise_eaw(ref, gb, cam, 64.0, 0.2, 1.0, lv)
        out[f"eaw{f}"] = ref
        accC[f], accM[f] = pyoracle.denoise_temporal_accumulate(accC[f ^ 1], accM[f ^ 1], noisy, gb, f == 0)
        var = pyoracle.denoise_estimate_variance(accM[f], W, H)
        fvar = pyoracle.denoise_filter_variance(var, W, H)
        col, var2 = pyoracle.denoise_svgf(accC[f], var, fvar, gb, cam, 4.0, 128.0, 1.0, 0)
        out[f"accum_color{f}"], out[f"accum_moment{f}"] = accC[f], accM[f]
        out[f"variance{f}"], out[f"filtered_variance{f}"] = var, fvar
        out[f"svgf_color{f}"], out[f"svgf_variance{f}"] = col, var2
        out[f"modulated{f}"] = pyoracle.denoise_modulate(ref, gb)
        for tone in (0, 1, 2):
            out[f"pbo{f}_tone{tone}"] = pyoracle.copy_image_to_pbo(out[f"modulated{f}"], W, H, 0, tone, 0.8)
        out[f"pbo{f}_motion"] = pyoracle.copy_image_to_pbo(gb.motion, W, H, 3)
        out[f"pbo{f}_depth"] = pyoracle.copy_image_to_pbo(gb.depth[cur] * np.float32(0.2), W, H, 2)
        gb.update(cam)
    path = os.path.join(ROOT, "tests", "golden", "post_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
