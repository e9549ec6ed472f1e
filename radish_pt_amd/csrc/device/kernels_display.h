// radish_pt_amd/csrc/device/kernels_display.h — the step AFTER the path: image → RGBA8 for the preview PBO.
//
// Restates sendImageToPBO's four overloads (/root/reference/src/pathtrace.cu:32-118) and the tone-mapping helpers
// Math::filmic / ACES / gammaCorrection (/root/reference/src/mathUtil.h:110-126).  Elementwise, HBM-bound: 12 B read +
// 4 B written per pixel.  The reference's gamma is glm::pow(c, 1/2.2) = CUDA libdevice powf, whose bits no other
// platform reproduces; both this file and the oracle evaluate pow_gamma_det below (binary32 operations in this order
// only, ~2e-7 relative), so the RGBA8 output is bit-identical between them.  Float → int follows CUDA's conversion
// (NaN → 0, saturating), which is what the reference's glm::ivec3(color * 255.f) compiles to on its GPU.
#pragma once
#include "rmath.h"

namespace rd {

// x^(1/2.2f) for finite x > 0: log2 by the atanh series on the mantissa in [sqrt(1/2), sqrt(2)), exp2 by a degree-6
// polynomial on [-1/2, 1/2].
RD_DEV float pow_gamma_det(float x) {
    if (!(x > 0.f)) return x == 0.f ? 0.f : __builtin_nanf("");  // pow(-c, 1/2.2) and pow(NaN, .) are NaN; pow(0, .) = 0
    if (x == __builtin_inff()) return x;
    int eAdj = 0;
    if (x < 1.17549435e-38f) {  // subnormal: scale into the normal range first
        x = x * 16777216.f;
        eAdj = -24;
    }
    uint32_t bits = __float_as_uint(x);
    int e = int((bits >> 23) & 0xffu) - 126 + eAdj;
    float m = __uint_as_float((bits & 0x007fffffu) | 0x3f000000u);  // [0.5, 1)
    if (m < 0.70710678118654752f) {
        m = m + m;
        e = e - 1;
    }
    float sN = (m - 1.f) / (m + 1.f);
    float z = sN * sN;
    float p = sN + sN * z * (0.333333333333f + z * (0.2f + z * (0.142857142857f + z * 0.111111111111f)));
    float l2 = p * 2.8853900817779268f;  // 2 / ln 2
    float t = (float(e) + l2) * (1.f / 2.2f);
    float n = __builtin_rintf(t);
    float f = t - n;
    float q = 1.f + f * (0.69314718056f + f * (0.240226506959f + f * (0.0555041086648f + f * (0.00961812910763f +
                  f * (0.00133335581464f + f * 0.000154035303934f)))));
    int ni = (int)n;
    if (ni < -125) return 0.f;  // cannot happen for a binary32 input (t >= -149/2.2); kept so the bit trick below is safe
    return __uint_as_float(__float_as_uint(q) + ((uint32_t)ni << 23));
}
RD_DEV v3 gammaCorrection(v3 c) { return mk3(pow_gamma_det(c.x), pow_gamma_det(c.y), pow_gamma_det(c.z)); }  // mathUtil.h:124-126
RD_DEV float calcFilmic1(float c) {  // mathUtil.h:110-113
    return (c * (c * 0.22f + 0.03f) + 0.002f) / (c * (c * 0.22f + 0.3f) + 0.06f) - 1.f / 30.f;
}
RD_DEV v3 filmic(v3 c) {  // mathUtil.h:114-116
    c = c * 1.6f;
    float d = calcFilmic1(11.2f);
    return mk3(calcFilmic1(c.x) / d, calcFilmic1(c.y) / d, calcFilmic1(c.z) / d);
}
RD_DEV float aces1(float c) { return (c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f); }  // mathUtil.h:118-121
RD_DEV v3 ACES(v3 c) { return mk3(aces1(c.x), aces1(c.y), aces1(c.z)); }

// clamp(int(c * 255.f), 0, 255) with CUDA's float→int conversion (NaN → 0, saturating)
RD_DEV uint32_t toByte(float c) {
    float v = c * 255.f;
    if (!(v > 0.f)) return 0u;
    if (v >= 255.f) return 255u;
    return (uint32_t)(int)v;
}
RD_DEV uint32_t packRGBA(v3 color) {  // make_uchar4(r, g, b, 0), little-endian
    color = gammaCorrection(color);
    return toByte(color.x) | (toByte(color.y) << 8) | (toByte(color.z) << 16);
}

// kind 0: vec3 image (tone mapping + scale, pathtrace.cu:32-59); 1: vec2 (:61-78); 2: float (:80-97);
// 3: int pixel index shown as normalised coordinates (:99-118)
__global__ __launch_bounds__(256) void k_send_image_to_pbo(uint32_t *__restrict__ pbo, const void *__restrict__ image, int width,
                                                           int height, int kind, int toneMapping, float scale) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)width * height) return;
    v3 color;
    if (kind == 0) {
        const float *img = static_cast<const float *>(image);
        color = mk3(img[3 * idx], img[3 * idx + 1], img[3 * idx + 2]) * scale;
        if (toneMapping == 1) color = filmic(color);
        else if (toneMapping == 2) color = ACES(color);
    } else if (kind == 1) {
        const float *img = static_cast<const float *>(image);
        color = mk3(img[2 * idx], img[2 * idx + 1], 0.f);
    } else if (kind == 2) {
        float g = static_cast<const float *>(image)[idx];
        color = mk3(g, g, g);
    } else {
        int v = static_cast<const int *>(image)[idx];
        int px = v % width, py = v / width;
        color = mk3(float(px) / float(width), float(py) / float(height), 0.f);
    }
    pbo[idx] = packRGBA(color);
}

}  // namespace rd
