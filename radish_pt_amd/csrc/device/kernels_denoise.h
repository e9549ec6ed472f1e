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

// ---- the two à-trous filters, MI355X-first ---------------------------------------------------------------------------------------
// The reference's kernels (denoiser.cu:17-84, :92-173) give every pixel a thread that gathers its 25 taps — 100 scattered 12-B /
// 4-B loads — and evaluates three weights per tap with exp / pow and IEEE divisions: ~150 instructions per tap, so the kernel is
// bound by VALU issue (0.2 ms of issue per SIMD at 1080p), not by memory.  The bit-exactness contract fixes every rounding, not
// the instruction selection; this version keeps all roundings and changes everything else:
//   * À-trous taps of step 2^level only ever connect pixels of the same residue class (x mod step, y mod step): a workgroup
//     filters a 16x16 tile of ONE residue class, which is an ordinary dense 5x5 stencil on that sub-lattice whatever the level.
//     Its 20x20 footprint (colour, normal, position, id — and for SVGF variance and luminance) is staged ONCE in LDS as SoA
//     (16 KB / 19 KB); the taps then read LDS, conflict-free, instead of L1.
//   * Two taps are evaluated per lane at a time on register pairs: v_pk_add_f32 / v_pk_mul_f32 are IEEE-exact per half, so
//     differences, dot products and exp_det's range reduction and polynomial cost half the instructions with the same bits.
//   * A division by a sigma that is a power of two (sigLumin = 64, sigDepth = 1 in LeveledEAWFilter) is a multiplication by its
//     exactly representable reciprocal — the same correctly rounded quotient for every operand — chosen per launch.
//   * Per-pixel values every tap used to recompute (SVGF's luminance(colorQ), the centre's denominator) are computed once.
// Accumulation order is the reference's (row-major taps), so every sum rounds as before.
RD_DEV f2v splat2(float v) { return f2v{v, v}; }
RD_DEV f2v exp_det2(f2v x) {  // exp_det on both halves
    f2v n = x * splat2(1.44269504088896341f);
    n = f2v{__builtin_rintf(n.x), __builtin_rintf(n.y)};
    const f2v r = (x - n * splat2(0.693359375f)) - n * splat2(-2.12194440e-4f);
    const f2v p = splat2(1.f) + r * (splat2(1.f) + r * (splat2(0.5f) + r * (splat2(0.166666671633720f) + r * (splat2(0.0416666679084301f) +
                  r * (splat2(0.00833333376795053f) + r * splat2(0.00138888892252f))))));
    float a = __uint_as_float(__float_as_uint(p.x) + ((uint32_t)(int)n.x << 23));
    float b = __uint_as_float(__float_as_uint(p.y) + ((uint32_t)(int)n.y << 23));
    a = (x.x < -87.3f) ? 0.f : a;
    b = (x.y < -87.3f) ? 0.f : b;
    a = (x.x > 88.7f) ? __builtin_inff() : a;
    b = (x.y > 88.7f) ? __builtin_inff() : b;
    a = (x.x == x.x) ? a : x.x;
    b = (x.y == x.y) ? b : x.y;
    return f2v{a, b};
}
// x / sigma: when sigma is a power of two (with a normal reciprocal) it is the multiplication by that reciprocal — exact either
// way — and which one is compiled in is a template parameter the host picks per launch (a run-time choice made the compiler
// evaluate both and select).
inline bool sigmaIsPow2(float sigma) {
    uint32_t b;
    memcpy(&b, &sigma, 4);
    const uint32_t e = (b >> 23) & 0xffu;
    return (b & 0x807fffffu) == 0u && e >= 2u && e <= 252u;
}
template <bool POW2>
RD_DEV float sigmaDiv1(float x, float sigma) { return POW2 ? x * (1.f / sigma) : x / sigma; }
template <bool POW2>
RD_DEV f2v sigmaDiv2(f2v x, float sigma) {
    if (POW2) return x * splat2(1.f / sigma);  // 1 / 2^k is exact; hoisted out of the tap loop
    return f2v{x.x / sigma, x.y / sigma};
}
RD_DEV f2v min1_2(f2v v) { return f2v{gmin(1.f, v.x), gmin(1.f, v.y)}; }
RD_DEV f2v dot2(f2v ax, f2v ay, f2v az, f2v bx, f2v by, f2v bz) { return (ax * bx + ay * by) + az * bz; }  // rmath.h dot(v3, v3), twice

constexpr int kDT = 16, kDA = 2, kDL = kDT + 2 * kDA, kDN = kDL * kDL;  // output tile edge, apron, staged edge, staged pixels
// The 25 taps in the reference's order (i = -2..2 outer = rows, j = -2..2 inner), two per entry: offset in the staged tile and
// Gaussian weight of each; the 26th "tap" repeats the 25th with the flag cleared.
struct TapPair {
    int e0, e1;
    float g0, g1;
    int second;  // 0: only the first tap of the pair exists
};
#define RD_TAP(t) (((t) / 5 - 2) * kDL + ((t) % 5 - 2))
#define RD_GAUSS5(a) ((a) == 0 ? .0030f : (a) == 1 ? .0133f : (a) == 2 ? .0219f : (a) == 3 ? .0133f : .0030f)
#define RD_GAUSS_ROW(i, j) ((i) == 2 ? ((j) == 2 ? .1621f : ((j) == 1 || (j) == 3) ? .0983f : .0219f) \
                          : ((i) == 1 || (i) == 3) ? ((j) == 2 ? .0983f : ((j) == 1 || (j) == 3) ? .0596f : .0133f) : RD_GAUSS5(j))
#define RD_G(t) RD_GAUSS_ROW((t) / 5, (t) % 5)
#define RD_PAIR(t) {RD_TAP(t), RD_TAP((t) + 1 < 25 ? (t) + 1 : (t)), RD_G(t), RD_G((t) + 1 < 25 ? (t) + 1 : (t)), (t) + 1 < 25 ? 1 : 0}
constexpr int kNoPixel = (int)0x80000000;                               // id of a staged slot outside the image

struct DenoiseTile {  // which residue class and tile this workgroup filters
    int step, rx, ry, u0, v0;  // image pixel of sub-lattice coordinate (u, v): (rx + step*u, ry + step*v)
};
RD_DEV DenoiseTile denoiseTile(int width, int height, int level) {
    DenoiseTile t;
    t.step = 1 << level;
    const int subW = (width + t.step - 1) / t.step, subH = (height + t.step - 1) / t.step;
    const int tilesX = (subW + kDT - 1) / kDT, tilesY = (subH + kDT - 1) / kDT;
    t.rx = int(blockIdx.x) / tilesX;
    t.ry = int(blockIdx.y) / tilesY;
    t.u0 = (int(blockIdx.x) % tilesX) * kDT;
    t.v0 = (int(blockIdx.y) % tilesY) * kDT;
    return t;
}
// grid of the tiled filters: tiles of every residue class (host side)
inline dim3 denoiseTileGrid(int width, int height, int level) {
    const int step = 1 << level;
    const int subW = (width + step - 1) / step, subH = (height + step - 1) / step;
    return dim3((unsigned)(((subW + kDT - 1) / kDT) * step), (unsigned)(((subH + kDT - 1) / kDT) * step));
}

// EAW: every weight is SYMMETRIC — w(p, q) = min(1, e^(-|cP - cQ|^2 / sL)) min(1, e^(-|nP - nQ|^2 / sN)) min(1, e^(-|xP - xQ|^2 / sD)) g(q - p),
// counted only when both pixels are in the image and carry the same id: the same bits whichever end evaluates it (the differences
// only change sign before they are squared).  So each unordered pair is evaluated ONCE: phase 1 computes, for every pixel of the
// tile and of a 2-row / 2-column rim before it, the 12 "forward" weights (taps after the centre in row-major order) — 360 pixels
// x 6 packed pairs dealt evenly over the 256 threads — into LDS; phase 2 adds up the 25 taps of each output pixel in the
// reference's order, taking a backward tap's weight from the neighbour that computed it.  4 320 weights per tile instead of 6 144.
constexpr int kEO = 16;                        // output tile edge
constexpr int kECR = kEO + 2, kECC = kEO + 4;  // compute region: rows -2..15, columns -2..17 (360 pixels)
constexpr int kEDR = kEO + 4, kEDC = kEO + 8;  // data region:    rows -2..17, columns -4..19 (480 pixels)
constexpr int kEDN = kEDR * kEDC, kECN = kECR * kECC;
RD_DEV constexpr int eawData(int r, int c) { return (r + 2) * kEDC + (c + 4); }
RD_DEV constexpr int eawComp(int r, int c) { return (r + 2) * kECC + (c + 2); }
// forward tap f = 0..11: (di, dj) = (0,1) (0,2) (1,-2) .. (1,2) (2,-2) .. (2,2)
RD_DEV constexpr int eawFwdI(int f) { return f < 2 ? 0 : (f < 7 ? 1 : 2); }
RD_DEV constexpr int eawFwdJ(int f) { return f < 2 ? f + 1 : (f < 7 ? f - 4 : f - 9); }
RD_DEV constexpr int eawFwdIndex(int di, int dj) { return di == 0 ? dj - 1 : (di == 1 ? dj + 4 : dj + 9); }
__device__ const int kEawFwdOfs[12] = {1, 2, kEDC - 2, kEDC - 1, kEDC, kEDC + 1, kEDC + 2, 2 * kEDC - 2, 2 * kEDC - 1, 2 * kEDC, 2 * kEDC + 1, 2 * kEDC + 2};
__device__ const float kEawFwdG[12] = {.0983f, .0219f, .0133f, .0596f, .0983f, .0596f, .0133f, .0030f, .0133f, .0219f, .0133f, .0030f};
constexpr unsigned kEawSkip = 0xBF800000u;  // -1.0f: "this pair does not count" (a weight is never negative)

template <bool P2L, bool P2N, bool P2D>  // sigLuminance / sigNormal / sigDepth is a power of two
__global__ __launch_bounds__(256) void k_eaw_filter(float *__restrict__ colorOut, const float *__restrict__ colorIn, DenoiseGB gb,
                                                    float sigDepth, float sigNormal, float sigLuminance, DCamera cam, int level) {
    __shared__ float sC[3][kEDN], sN[3][kEDN], sP[3][kEDN];
    __shared__ int sId[kEDN];
    __shared__ float sW[12][kECN];
    const DenoiseTile T = denoiseTile(cam.resx, cam.resy, level);
    for (int e = int(threadIdx.x); e < kEDN; e += 256) {  // stage the 20x24 footprint of this tile's residue class
        const int u = T.u0 + e % kEDC - 4, v = T.v0 + e / kEDC - 2;
        const int x = T.rx + T.step * u, y = T.ry + T.step * v;
        int id = kNoPixel;
        v3 c = mk3(0.f), n = mk3(0.f), q = mk3(0.f);
        if (u >= 0 && v >= 0 && x < cam.resx && y < cam.resy) {
            const int idx = x + y * cam.resx;
            id = gb.primId[idx];
            c = load3(colorIn, idx);
            n = load3(gb.normal, idx);
            q = load3(gb.position, idx);
        }
        sId[e] = id;
        sC[0][e] = c.x; sC[1][e] = c.y; sC[2][e] = c.z;
        sN[0][e] = n.x; sN[1][e] = n.y; sN[2][e] = n.z;
        sP[0][e] = q.x; sP[1][e] = q.y; sP[2][e] = q.z;
    }
    __syncthreads();
    // ---- phase 1: forward weights of the compute region, two per item ----
#pragma unroll 1
    for (int k = int(threadIdx.x); k < kECN * 6; k += 256) {
        const int cp = k / 6, pr = k - cp * 6;       // pixel of the compute region, pair of forward taps (2 pr, 2 pr + 1)
        const int r = cp / kECC, cc = cp - r * kECC;  // 0-based row / column inside the compute region
        const int de = r * kEDC + cc + 2;             // its slot in the data region (compute column 0 = data column 2)
        const int e0 = de + kEawFwdOfs[2 * pr], e1 = de + kEawFwdOfs[2 * pr + 1];
        const int idP = sId[de];
        const bool ok0 = idP > -1 && sId[e0] == idP, ok1 = idP > -1 && sId[e1] == idP;  // both in the image, same id (:40-44)
        const v3 colorP = mk3(sC[0][de], sC[1][de], sC[2][de]), normalP = mk3(sN[0][de], sN[1][de], sN[2][de]),
                 posP = mk3(sP[0][de], sP[1][de], sP[2][de]);
        const f2v dcx = splat2(colorP.x) - f2v{sC[0][e0], sC[0][e1]}, dcy = splat2(colorP.y) - f2v{sC[1][e0], sC[1][e1]},
                  dcz = splat2(colorP.z) - f2v{sC[2][e0], sC[2][e1]};
        const f2v dnx = splat2(normalP.x) - f2v{sN[0][e0], sN[0][e1]}, dny = splat2(normalP.y) - f2v{sN[1][e0], sN[1][e1]},
                  dnz = splat2(normalP.z) - f2v{sN[2][e0], sN[2][e1]};
        const f2v dpx = splat2(posP.x) - f2v{sP[0][e0], sP[0][e1]}, dpy = splat2(posP.y) - f2v{sP[1][e0], sP[1][e1]},
                  dpz = splat2(posP.z) - f2v{sP[2][e0], sP[2][e1]};
        const f2v wColor = min1_2(exp_det2(sigmaDiv2<P2L>(-dot2(dcx, dcy, dcz, dcx, dcy, dcz), sigLuminance)));
        const f2v wNormal = min1_2(exp_det2(sigmaDiv2<P2N>(-dot2(dnx, dny, dnz, dnx, dny, dnz), sigNormal)));
        const f2v wPos = min1_2(exp_det2(sigmaDiv2<P2D>(-dot2(dpx, dpy, dpz, dpx, dpy, dpz), sigDepth)));
        const f2v weight = wColor * wNormal * wPos * f2v{kEawFwdG[2 * pr], kEawFwdG[2 * pr + 1]};
        sW[2 * pr][cp] = ok0 ? weight.x : __uint_as_float(kEawSkip);
        sW[2 * pr + 1][cp] = ok1 ? weight.y : __uint_as_float(kEawSkip);
    }
    __syncthreads();
    // ---- phase 2: the 25 taps of every output pixel, in the reference's order ----
    const int lx = int(threadIdx.x & 15u), ly = int(threadIdx.x >> 4);
    const int x = T.rx + T.step * (T.u0 + lx), y = T.ry + T.step * (T.v0 + ly);
    if (x >= cam.resx || y >= cam.resy) return;
    const int idxP = x + y * cam.resx;
    const int de = eawData(ly, lx), cp = eawComp(ly, lx);
    const v3 colorP = mk3(sC[0][de], sC[1][de], sC[2][de]);
    if (sId[de] <= -1) {
        store3(colorOut, idxP, colorP);
        return;
    }
    v3 sum = mk3(0.f);
    float weightSum = 0.f;
#pragma unroll
    for (int t = 0; t < 25; t++) {
        const int di = t / 5 - 2, dj = t % 5 - 2;
        float w;
        if (di == 0 && dj == 0) {  // the centre: the pair (p, p), evaluated here as the reference evaluates it
            const v3 normalP = mk3(sN[0][de], sN[1][de], sN[2][de]), posP = mk3(sP[0][de], sP[1][de], sP[2][de]);
            const v3 dc = colorP - colorP, dn = normalP - normalP, dp = posP - posP;
            w = gmin(1.f, exp_det(sigmaDiv1<P2L>(-dot(dc, dc), sigLuminance))) * gmin(1.f, exp_det(sigmaDiv1<P2N>(-dot(dn, dn), sigNormal))) *
                gmin(1.f, exp_det(sigmaDiv1<P2D>(-dot(dp, dp), sigDepth))) * kGaussian5x5[2][2];
        } else if (di > 0 || (di == 0 && dj > 0)) {
            w = sW[eawFwdIndex(di, dj)][cp];
        } else {
            w = sW[eawFwdIndex(-di, -dj)][cp + di * kECC + dj];
        }
        if (__float_as_uint(w) != kEawSkip) {
            const int e = de + di * kEDC + dj;
            sum = sum + mk3(sC[0][e], sC[1][e], sC[2][e]) * w;
            weightSum += w;
        }
    }
    store3(colorOut, idxP, (weightSum == 0.f) ? colorP : sum / weightSum);
}

template <bool P2D>  // sigDepth + 1e-4f is a power of two (it is not for the reference's sigDepth = 1)
__global__ __launch_bounds__(256) void k_svgf_filter(float *__restrict__ colorOut, const float *__restrict__ colorIn,
                                                     float *__restrict__ varianceOut, const float *__restrict__ varianceIn,
                                                     const float *__restrict__ varFiltered, DenoiseGB gb, float sigDepth,
                                                     float sigNormal, float sigLuminance, DCamera cam, int level) {
    __shared__ float sC[3][kDN], sN[3][kDN], sP[3][kDN], sVar[kDN], sLum[kDN];
    __shared__ int sId[kDN];
    const DenoiseTile T = denoiseTile(cam.resx, cam.resy, level);
    for (int e = int(threadIdx.x); e < kDN; e += 256) {
        const int u = T.u0 + e % kDL - kDA, v = T.v0 + e / kDL - kDA;
        const int x = T.rx + T.step * u, y = T.ry + T.step * v;
        int id = kNoPixel;
        v3 c = mk3(0.f), n = mk3(0.f), q = mk3(0.f);
        float var = 0.f;
        if (u >= 0 && v >= 0 && x < cam.resx && y < cam.resy) {
            const int idx = x + y * cam.resx;
            id = gb.primId[idx];
            c = load3(colorIn, idx);
            n = load3(gb.normal, idx);
            q = load3(gb.position, idx);
            var = varianceIn[idx];
        }
        sId[e] = id;
        sC[0][e] = c.x; sC[1][e] = c.y; sC[2][e] = c.z;
        sN[0][e] = n.x; sN[1][e] = n.y; sN[2][e] = n.z;
        sP[0][e] = q.x; sP[1][e] = q.y; sP[2][e] = q.z;
        sVar[e] = var;
        sLum[e] = luminance(c);  // every tap of every neighbour used to recompute it
    }
    __syncthreads();
    const int lx = int(threadIdx.x & 15u), ly = int(threadIdx.x >> 4);
    const int x = T.rx + T.step * (T.u0 + lx), y = T.ry + T.step * (T.v0 + ly);
    if (x >= cam.resx || y >= cam.resy) return;
    const int idxP = x + y * cam.resx;
    const int ce = (ly + kDA) * kDL + lx + kDA;
    const v3 colorP = mk3(sC[0][ce], sC[1][ce], sC[2][ce]);
    if (sId[ce] <= -1) {
        store3(colorOut, idxP, colorP);
        varianceOut[idxP] = sVar[ce];
        return;
    }
    const v3 normalP = mk3(sN[0][ce], sN[1][ce], sN[2][ce]), posP = mk3(sP[0][ce], sP[1][ce], sP[2][ce]);
    const float lumP = sLum[ce];
    const float denom = sigLuminance * __builtin_sqrtf(gmax(varFiltered[idxP], 0.f)) + 1e-4f;  // the same for all 25 taps
    const float sigPos = sigDepth + 1e-4f;
    v3 colorSum = mk3(0.f);
    float varianceSum = 0.f, weightSum = 0.f, weight2Sum = 0.f;
#pragma unroll
    for (int tp = 0; tp < 13; tp++) {
        __builtin_amdgcn_sched_barrier(0);  // one tap pair at a time: without the fence the scheduler hoists all 260 LDS reads (214 VGPRs)
        constexpr TapPair kTaps[13] = {RD_PAIR(0), RD_PAIR(2), RD_PAIR(4), RD_PAIR(6), RD_PAIR(8), RD_PAIR(10), RD_PAIR(12),
                                       RD_PAIR(14), RD_PAIR(16), RD_PAIR(18), RD_PAIR(20), RD_PAIR(22), RD_PAIR(24)};
        const TapPair tap = kTaps[tp];
        const int e0 = ce + tap.e0, e1 = ce + tap.e1;
        const bool ok0 = sId[e0] != kNoPixel, ok1 = tap.second != 0 && sId[e1] != kNoPixel;  // inside the image (:120-122)
        const v3 c0 = mk3(sC[0][e0], sC[1][e0], sC[2][e0]), c1 = mk3(sC[0][e1], sC[1][e1], sC[2][e1]);
        const v3 n0 = mk3(sN[0][e0], sN[1][e0], sN[2][e0]), n1 = mk3(sN[0][e1], sN[1][e1], sN[2][e1]);
        const f2v dpx = splat2(posP.x) - f2v{sP[0][e0], sP[0][e1]}, dpy = splat2(posP.y) - f2v{sP[1][e0], sP[1][e1]},
                  dpz = splat2(posP.z) - f2v{sP[2][e0], sP[2][e1]};
        const f2v wPos = exp_det2(sigmaDiv2<P2D>(-dot2(dpx, dpy, dpz, dpx, dpy, dpz), sigPos));
        const f2v wNormal = f2v{pow_det(satDot(normalP, n0), sigNormal), pow_det(satDot(normalP, n1), sigNormal)} + splat2(1e-4f);
        const f2v dl = splat2(lumP) - f2v{sLum[e0], sLum[e1]};
        const f2v adl = f2v{fabs_(dl.x), fabs_(dl.y)};
        const f2v wColor = exp_det2(f2v{-adl.x / denom, -adl.y / denom}) + splat2(1e-4f);
        const f2v weight = wColor * wNormal * wPos * f2v{tap.g0, tap.g1};
        const f2v weight2 = weight * weight;
        if (ok0) {
            colorSum = colorSum + c0 * weight.x;
            varianceSum += sVar[e0] * weight2.x;
            weightSum += weight.x;
            weight2Sum += weight2.x;
        }
        if (ok1) {
            colorSum = colorSum + c1 * weight.y;
            varianceSum += sVar[e1] * weight2.y;
            weightSum += weight.y;
            weight2Sum += weight2.y;
        }
    }
    store3(colorOut, idxP, (weightSum < 1.1920928955078125e-7f) ? colorP : colorSum / weightSum);
    varianceOut[idxP] = (weight2Sum < 1.1920928955078125e-7f) ? sVar[ce] : varianceSum / weight2Sum;
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
