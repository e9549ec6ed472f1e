// radish_pt_amd/csrc/device/kernels_denoise.h — the EAW à-trous and SVGF denoisers.
//
// Restates the kernels of /root/reference/src/denoiser.cu: waveletFilter (EAW, :17-84), waveletFilter (SVGF with
// variance, :92-173), modulate (:175-185), add (:187-206), temporalAccumulate (:208-262), estimateVariance (:264-299),
// filterVariance (:301-328), plus Camera::getPosition (/root/reference/src/sceneStructs.h:50-70) with the compile-time
// switches in force (DENOISER_ENCODE_POSITION true: depth plane; DENOISER_ENCODE_NORMAL false: vec3 normals,
// src/common.h:12-14).  The shipped frame loop never calls them; they consume exactly what the hot path
// produces: the direct / indirect split, demodulated by the G-buffer albedo.
//
// Stencil kernels over 12-B pixels: HBM/L2-bound (25 taps x 32 B of G-buffer + colour per pixel, all neighbours shared
// with adjacent lanes).  One lane per pixel, 8x8 pixels per wave so that the 5x5 footprint of a wave overlaps in L1.
// exp / pow are the reference's glm::exp / glm::pow = CUDA libdevice; as for sin/cos, both this file and the oracle
// evaluate fixed binary32 recipes (exp_det, pow_det below) so that they agree bit for bit.
#pragma once
#include "kernels_display.h"
#include "kernels_pt.h"

namespace rd {

// e^x: n = rint(x log2 e), r = x - n ln2 (two-term Cody-Waite), degree-6 polynomial on [-ln2/2, ln2/2], scale by 2^n.
RD_DEV float exp_det(float x) {
    if (!(x == x)) return x;
    if (x > 88.7f) return __builtin_inff();
    if (x < -87.3f) return 0.f;  // below the smallest normal: the reference's weights flush to ~0 there as well
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = (x - n * 0.693359375f) - n * -2.12194440e-4f;
    float p = 1.f + r * (1.f + r * (0.5f + r * (0.166666671633720f + r * (0.0416666679084301f +
                  r * (0.00833333376795053f + r * 0.00138888892252f)))));
    return __uint_as_float(__float_as_uint(p) + ((uint32_t)(int)n << 23));
}
// x^y for x >= 0 (Math::satDot(...) ^ sigNormal): 0^y = 0 for y > 0, else 2^(y log2 x) with pow_gamma_det's log2 / exp2.
RD_DEV float pow_det(float x, float y) {
    if (!(x > 0.f)) return (x == 0.f && y > 0.f) ? 0.f : ((x == 0.f && y == 0.f) ? 1.f : __builtin_nanf(""));
    if (x == __builtin_inff()) return x;
    int eAdj = 0;
    if (x < 1.17549435e-38f) {
        x = x * 16777216.f;
        eAdj = -24;
    }
    uint32_t bits = __float_as_uint(x);
    int e = int((bits >> 23) & 0xffu) - 126 + eAdj;
    float m = __uint_as_float((bits & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752f) {
        m = m + m;
        e = e - 1;
    }
    float sN = (m - 1.f) / (m + 1.f);
    float z = sN * sN;
    float p = sN + sN * z * (0.333333333333f + z * (0.2f + z * (0.142857142857f + z * 0.111111111111f)));
    float l2 = p * 2.8853900817779268f;
    float t = (float(e) + l2) * y;
    if (t > 127.9f) return __builtin_inff();
    if (t < -125.9f) return 0.f;
    float n = __builtin_rintf(t);
    float f = t - n;
    float q = 1.f + f * (0.69314718056f + f * (0.240226506959f + f * (0.0555041086648f + f * (0.00961812910763f +
                  f * (0.00133335581464f + f * 0.000154035303934f)))));
    return __uint_as_float(__float_as_uint(q) + ((uint32_t)(int)n << 23));
}

struct DenoiseGB {  // the planes of GBuffer the denoisers read (gBuffer.h:29-57), current / last already selected
    const float *albedo, *normal, *lastNormal, *depth;
    const float *position;  // k_position_plane's output (filters only): Camera::getPosition of every pixel, computed once
    const int *motion, *primId, *lastPrimId;
    int width, height;
};

RD_DEV v3 cameraGetPosition(const DCamera &cam, int x, int y, float dist) {  // sceneStructs.h:50-70
    float aspect = float(cam.resx) / float(cam.resy);
    v2 pixelSize = {1.f / float(cam.resx), 1.f / float(cam.resy)};
    v2 scr = mk2(float(x), float(y)) * pixelSize;
    v2 ruv = scr + pixelSize * mk2(0.5f, 0.5f);
    ruv = {1.f - ruv.x * 2.f, 1.f - ruv.y * 2.f};
    v3 pLens = mk3(0.f);
    v2 f = (ruv * mk2(aspect, 1.f)) * cam.tanFovY;
    v3 pFocus = mk3(f.x, f.y, 1.f) * cam.focalDist;
    v3 dir = pFocus - pLens;
    dir = normalize(mul(m3{cam.right, cam.up, cam.view}, dir));
    v3 ori = cam.position + cam.right * pLens.x + cam.up * pLens.y;
    return ori + dir * dist;
}

__device__ const float kGaussian3x3[3][3] = {{.075f, .124f, .075f}, {.124f, .204f, .124f}, {.075f, .124f, .075f}};
__device__ const float kGaussian5x5[5][5] = {{.0030f, .0133f, .0219f, .0133f, .0030f},
                                             {.0133f, .0596f, .0983f, .0596f, .0133f},
                                             {.0219f, .0983f, .1621f, .0983f, .0219f},
                                             {.0133f, .0596f, .0983f, .0596f, .0133f},
                                             {.0030f, .0133f, .0219f, .0133f, .0030f}};

// pixel of this lane: 8x8 pixels per wave, four waves per 256-thread workgroup side by side
RD_DEV bool denoisePixel(int width, int height, int &x, int &y) {
    const int bx = int(blockIdx.x) * 32 + int(threadIdx.x >> 6) * 8 + int(threadIdx.x & 7u);
    const int by = int(blockIdx.y) * 8 + int((threadIdx.x >> 3) & 7u);
    x = bx;
    y = by;
    return x < width && y < height;
}

// The filters need Camera::getPosition(x, y, depth) of the pixel and of each of its 25 taps (denoiser.cu:47-48, :127-128):
// ~110 instructions with an IEEE square root and three divisions, 40 % of a tap.  It depends on the tap's pixel only, so it
// is computed once per pixel into a plane (same function, same bits) and the taps read 12 B instead.
__global__ __launch_bounds__(256) void k_position_plane(float *__restrict__ position, const float *__restrict__ depth, DCamera cam) {
    int x, y;
    if (!denoisePixel(cam.resx, cam.resy, x, y)) return;
    const int idx = x + y * cam.resx;
    store3(position, idx, cameraGetPosition(cam, x, y, depth[idx]));
}

__global__ __launch_bounds__(256) void k_eaw_filter(float *__restrict__ colorOut, const float *__restrict__ colorIn, DenoiseGB gb,
                                                    float sigDepth, float sigNormal, float sigLuminance, DCamera cam, int level) {
    int x, y;
    if (!denoisePixel(cam.resx, cam.resy, x, y)) return;
    const int step = 1 << level;
    const int idxP = x + y * cam.resx;
    const int primIdP = gb.primId[idxP];
    if (primIdP <= -1) {
        store3(colorOut, idxP, load3(colorIn, idxP));
        return;
    }
    const v3 colorP = load3(colorIn, idxP);
    const v3 normalP = load3(gb.normal, idxP);
    const v3 posP = load3(gb.position, idxP);
    v3 sum = mk3(0.f);
    float weightSum = 0.f;
    for (int i = -2; i <= 2; i++)
        for (int j = -2; j <= 2; j++) {
            const int qx = x + j * step, qy = y + i * step;
            if (qx >= cam.resx || qy >= cam.resy || qx < 0 || qy < 0) continue;
            const int idxQ = qx + qy * cam.resx;
            if (gb.primId[idxQ] != primIdP) continue;
            const v3 normalQ = load3(gb.normal, idxQ);
            const v3 posQ = load3(gb.position, idxQ);
            const v3 colorQ = load3(colorIn, idxQ);
            const v3 dc = colorP - colorQ, dn = normalP - normalQ, dp = posP - posQ;
            const float wColor = gmin(1.f, exp_det(-dot(dc, dc) / sigLuminance));
            const float wNormal = gmin(1.f, exp_det(-dot(dn, dn) / sigNormal));
            const float wPos = gmin(1.f, exp_det(-dot(dp, dp) / sigDepth));
            const float weight = wColor * wNormal * wPos * kGaussian5x5[i + 2][j + 2];
            sum = sum + colorQ * weight;
            weightSum += weight;
        }
    store3(colorOut, idxP, (weightSum == 0.f) ? load3(colorIn, idxP) : sum / weightSum);
}

__global__ __launch_bounds__(256) void k_svgf_filter(float *__restrict__ colorOut, const float *__restrict__ colorIn,
                                                     float *__restrict__ varianceOut, const float *__restrict__ varianceIn,
                                                     const float *__restrict__ varFiltered, DenoiseGB gb, float sigDepth,
                                                     float sigNormal, float sigLuminance, DCamera cam, int level) {
    int x, y;
    if (!denoisePixel(cam.resx, cam.resy, x, y)) return;
    const int step = 1 << level;
    const int idxP = x + y * cam.resx;
    const int primIdP = gb.primId[idxP];
    if (primIdP <= -1) {
        store3(colorOut, idxP, load3(colorIn, idxP));
        varianceOut[idxP] = varianceIn[idxP];
        return;
    }
    const v3 colorP = load3(colorIn, idxP);
    const v3 normalP = load3(gb.normal, idxP);
    const v3 posP = load3(gb.position, idxP);
    v3 colorSum = mk3(0.f);
    float varianceSum = 0.f, weightSum = 0.f, weight2Sum = 0.f;
    for (int i = -2; i <= 2; i++)
        for (int j = -2; j <= 2; j++) {
            const int qx = x + j * step, qy = y + i * step;
            if (qx >= cam.resx || qy >= cam.resy || qx < 0 || qy < 0) continue;
            const int idxQ = qx + qy * cam.resx;
            const v3 normalQ = load3(gb.normal, idxQ);
            const v3 posQ = load3(gb.position, idxQ);
            const float varQ = varianceIn[idxQ];
            const v3 colorQ = load3(colorIn, idxQ);
            const v3 dp = posP - posQ;
            const float wPos = exp_det(-dot(dp, dp) / (sigDepth + 1e-4f));
            const float wNormal = pow_det(satDot(normalP, normalQ), sigNormal) + 1e-4f;
            const float denom = sigLuminance * __builtin_sqrtf(gmax(varFiltered[idxP], 0.f)) + 1e-4f;
            const float wColor = exp_det(-fabs_(luminance(colorP) - luminance(colorQ)) / denom) + 1e-4f;
            const float weight = wColor * wNormal * wPos * kGaussian5x5[i + 2][j + 2];
            const float weight2 = weight * weight;
            colorSum = colorSum + colorQ * weight;
            varianceSum += varQ * weight2;
            weightSum += weight;
            weight2Sum += weight2;
        }
    store3(colorOut, idxP, (weightSum < 1.1920928955078125e-7f) ? load3(colorIn, idxP) : colorSum / weightSum);
    varianceOut[idxP] = (weight2Sum < 1.1920928955078125e-7f) ? varianceIn[idxP] : varianceSum / weight2Sum;
}

// modulate (:175-185): LDRToHDR is the identity (`return c /= 1.f;`, mathUtil.h:53-56)
__global__ __launch_bounds__(256) void k_modulate(float *__restrict__ image, const float *__restrict__ albedo, long long n) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    v3 color = load3(image, (int)idx) / 1.f;
    v3 a = load3(albedo, (int)idx);
    store3(image, (int)idx, color * mk3(gmax(a.x, 0.f), gmax(a.y, 0.f), gmax(a.z, 0.f)));
}
// add (:187-206); out may alias in1 (the two-argument overload)
__global__ __launch_bounds__(256) void k_add_images(float *out, const float *in1, const float *in2, long long n3) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) out[i] = in1[i] + in2[i];
}

__global__ __launch_bounds__(256) void k_temporal_accumulate(float *__restrict__ colorAccumOut, const float *__restrict__ colorAccumIn,
                                                             float *__restrict__ momentAccumOut, const float *__restrict__ momentAccumIn,
                                                             const float *__restrict__ colorIn, DenoiseGB gb, int first) {
    const float alpha = 0.2f;
    int x, y;
    if (!denoisePixel(gb.width, gb.height, x, y)) return;
    const int idx = x + y * gb.width;
    const int primId = gb.primId[idx];
    const int lastIdx = gb.motion[idx];
    bool diff = first != 0;
    if (lastIdx < 0) diff = true;
    else if (primId <= -1) diff = true;
    else if (gb.lastPrimId[lastIdx] != primId) diff = true;
    else if (fabs_(dot(load3(gb.normal, idx), load3(gb.lastNormal, lastIdx))) < .1f) diff = true;
    const v3 color = load3(colorIn, idx);
    const float lum = luminance(color);
    v3 colorAccum, momentAccum;
    if (diff) {  // the reference reads colorAccumIn[lastIdx] before this test, out of bounds for lastIdx = -1; unused then
        colorAccum = color;
        momentAccum = mk3(lum, lum * lum, 0.f);
    } else {
        const v3 lastColor = load3(colorAccumIn, lastIdx), lastMoment = load3(momentAccumIn, lastIdx);
        colorAccum = lastColor * (1.f - alpha) + color * alpha;  // glm::mix
        momentAccum = mk3(lastMoment.x * (1.f - alpha) + lum * alpha, lastMoment.y * (1.f - alpha) + (lum * lum) * alpha,
                          lastMoment.z + 1.f);
    }
    store3(colorAccumOut, idx, colorAccum);
    store3(momentAccumOut, idx, momentAccum);
}

__global__ __launch_bounds__(256) void k_estimate_variance(float *__restrict__ variance, const float *__restrict__ moment, int width,
                                                           int height) {
    int x, y;
    if (!denoisePixel(width, height, x, y)) return;
    const int idx = x + y * width;
    const v3 m = load3(moment, idx);
    if (m.z > 3.5f) {
        variance[idx] = m.y - m.x * m.x;
    } else {
        float sx = 0.f, sy = 0.f;
        int pixelCount = 0;
        for (int i = -1; i <= 1; i++)
            for (int j = -1; j <= 1; j++) {
                const int qx = x + j, qy = y + i;
                if (qx < 0 || qx >= width || qy < 0 || qy >= height) continue;
                const v3 q = load3(moment, qx + qy * width);
                sx += q.x;
                sy += q.y;
                pixelCount++;
            }
        sx = sx / float(pixelCount);
        sy = sy / float(pixelCount);
        variance[idx] = sy - sx * sx;
    }
}

__global__ __launch_bounds__(256) void k_filter_variance(float *__restrict__ varianceOut, const float *__restrict__ varianceIn, int width,
                                                         int height) {
    int x, y;
    if (!denoisePixel(width, height, x, y)) return;
    float sum = 0.f, weightSum = 0.f;
    for (int i = -1; i <= 1; i++)
        for (int j = -1; j <= 1; j++) {
            const int qx = x + i, qy = y + j;
            if (qx < 0 || qx >= width || qy < 0 || qy >= height) continue;
            const float weight = kGaussian3x3[i + 1][j + 1];
            sum += varianceIn[qx + qy * width] * weight;
            weightSum += weight;
        }
    varianceOut[x + y * width] = sum / weightSum;
}

}  // namespace rd
