// radish_pt_amd/csrc/device/kernels_restir.h — ReSTIR DI as two launches.
//
// The reference fuses everything into ReSTIRDirectKernel (/root/reference/src/restir.cu:97-203) and separates
// "store my reservoir" from "gather my neighbours' reservoirs" with a block-scope __syncthreads() although the
// neighbours (radius 5) live in other 8x8 blocks — a cross-block race.  Here the kernel boundary is
// the barrier: pass 1 = primary hit, RIS, shadow ray, temporal merge, store; pass 2 = spatial merge, shade,
// x albedo, running mean.  Per-pixel state that the reference keeps in registers across its barrier travels in
// a 48-byte record.
#pragma once
#include "kernels_pt.h"

namespace rd {

struct LightLiSample {  // restir.h:95-99
    v3 Li, wi;
    float dist;
};
struct Reservoir {  // restir.h:10-92, 36 B when stored
    LightLiSample sample;
    int numSamples;
    float weight;
};
RD_DEV Reservoir emptyReservoir() { return Reservoir{{mk3(0.f), mk3(0.f), 0.f}, 0, 0.f}; }
RD_DEV Reservoir loadReservoir(const float *buf, long long i) {
    const float *p = buf + 9 * i;
    Reservoir r;
    r.sample.Li = mk3(p[0], p[1], p[2]);
    r.sample.wi = mk3(p[3], p[4], p[5]);
    r.sample.dist = p[6];
    r.numSamples = __float_as_int(p[7]);
    r.weight = p[8];
    return r;
}
RD_DEV void storeReservoir(float *buf, long long i, const Reservoir &r) {
    float *p = buf + 9 * i;
    p[0] = r.sample.Li.x; p[1] = r.sample.Li.y; p[2] = r.sample.Li.z;
    p[3] = r.sample.wi.x; p[4] = r.sample.wi.y; p[5] = r.sample.wi.z;
    p[6] = r.sample.dist;
    p[7] = __int_as_float(r.numSamples);
    p[8] = r.weight;
}
RD_DEV void resvUpdate(Reservoir &r, const LightLiSample &s, float newWeight, float rnd, bool faithful) {  // restir.h:17-24
    r.weight += newWeight;
    r.numSamples++;
    // restir.h:21 tests a float for truthiness; the corrected form is the usual `rand*W < w`.
    bool take = faithful ? ((rnd * r.weight / newWeight) != 0.f) : (rnd * r.weight < newWeight);
    if (take) r.sample = s;
}
RD_DEV bool resvInvalid(const Reservoir &r) { return isNanOrInf(r.weight) || r.weight < 0.f; }  // :42
RD_DEV void resvCheckValidity(Reservoir &r) {  // :44-49 → clear() :26-29 (keeps the stale sample)
    if (resvInvalid(r)) {
        r.weight = 0.f;
        r.numSamples = 0;
    }
}
RD_DEV void resvMerge(Reservoir &r, const Reservoir &rhs, float rnd) {  // :51-58
    r.weight += rhs.weight;
    r.numSamples += rhs.numSamples;
    if (rnd * r.weight < rhs.weight) r.sample = rhs.sample;
}
RD_DEV void resvPreClampedMerge(Reservoir &r, Reservoir rhs, float rnd, int M) {  // :69-77
    if (rhs.numSamples > 0 && rhs.numSamples > (M - 1) * r.numSamples && r.numSamples > 0) {
        rhs.weight *= static_cast<float>(M - 1) * r.numSamples / rhs.numSamples;
        rhs.numSamples = (M - 1) * r.numSamples;
    }
    resvMerge(r, rhs, rnd);
}

struct RestirArgs {
    float *reservoirOut;
    const float *reservoirIn;
    float *reservoirTemp;
    float4 *state;  // 3 x float4 per pixel
    // G-buffer of this frame / last frame (frame layout)
    const float *albedo, *normalCur, *normalLast, *depthCur;
    const int *motion, *primIdCur, *primIdLast;
    int gbWidth, gbHeight;
    int firstFrame, reuseMask, risCount, numSpatial, temporalClamp, faithfulRIS;
};

RD_DEV Reservoir findTemporalNeighbor(const RestirArgs &a, int idx) {  // restir.cu:19-40
    int primId = a.primIdCur[idx];
    int lastIdx = a.motion[idx];
    bool diff = false;
    if (lastIdx < 0) diff = true;
    else if (primId <= -1) diff = true;
    else if (a.primIdLast[lastIdx] != primId) diff = true;
    else {
        v3 norm = load3(a.normalCur, idx);
        v3 lastNorm = load3(a.normalLast, lastIdx);
        if (absDot(norm, lastNorm) < .1f) diff = true;
    }
    return diff ? emptyReservoir() : loadReservoir(a.reservoirIn, lastIdx);
}

RD_DEV Reservoir findSpatialNeighborDisk(const RestirArgs &a, int x, int y, v2 rnd) {  // restir.cu:42-80
    const float radius = 5.f;
    int W = a.gbWidth, H = a.gbHeight;
    int idx = y * W + x;
    v2 p = concentricSampleDisk(rnd.x, rnd.y) * radius;
    int px = (int)(float(x) + .5f + p.x);
    int py = (int)(float(y) + .5f + p.y);
    int pIdx = py * W + px;
    bool diff = false;
    if (px < 0 || px >= W || py < 0 || py >= H || (px == x && py == y)) diff = true;
    else if (a.primIdCur[pIdx] != a.primIdCur[idx]) diff = true;
    else {
        v3 norm = load3(a.normalCur, idx);
        v3 pNorm = load3(a.normalCur, pIdx);
        if (dot(norm, pNorm) < .1f) diff = true;
        float depth = a.depthCur[idx];
        float pDepth = a.depthCur[pIdx];
        if (fabs_(depth - pDepth) > depth * .1f) diff = true;
    }
    return diff ? emptyReservoir() : loadReservoir(a.reservoirTemp, pIdx);
}

// Shade + write (restir.cu:189-202).  `status`: >= 0 shade with this material id; -1 miss; -2 emitter.
// `missDirect`: what a primary miss leaves in `direct` (the env map sample, restir.cu:117-122); `metallic`/`roughness`:
// the textured values pass 1 shaded with (the reference keeps the whole Material in registers across its barrier).
RD_DEV void restirFinish(const DScene &s, const RestirArgs &a, int idx, int status, v3 norm, v3 wo, Reservoir reservoir,
                         Sampler &rng, int x, int y, bool doSpatial, float *directIllum, int iter, v3 missDirect,
                         float metallic, float roughness, int outIdx) {
    v3 direct = (status == -2) ? mk3(1.f) : missDirect;
    if (status >= 0) {
        Material material = loadMaterial(s.mats, status);
        material.baseColor = mk3(1.f);
        material.metallic = metallic;
        material.roughness = roughness;
        if (doSpatial) {
            Reservoir resvr = emptyReservoir();  // mergeSpatialNeighborDirect (:82-95)
            for (int i = 0; i < a.numSpatial; i++) {
                v2 r2 = sample2D(rng);
                Reservoir spatial = findSpatialNeighborDisk(a, x, y, r2);
                if (!resvInvalid(spatial)) resvMerge(resvr, spatial, rng.sample());
            }
            if (!resvInvalid(resvr) && !resvInvalid(reservoir)) resvMerge(reservoir, resvr, rng.sample());
        }
        LightLiSample smp = reservoir.sample;
        if (!resvInvalid(reservoir)) {
            v3 pHat = smp.Li * materialBSDF(material, norm, wo, smp.wi) * satDot(norm, smp.wi);  // restir.h:31-35
            float Wgt = reservoir.weight / (length(pHat) * static_cast<float>(reservoir.numSamples));  // :37-40
            direct = pHat * Wgt;
        }
        if (hasNanOrInf(direct)) direct = mk3(0.f);
    }
    direct = direct * load3(a.albedo, idx);
    storeRunningMean(directIllum, outIdx, direct, iter);  // outIdx: frame index, or the packed-tile index when world > 1
}

// apronBlocks > 0: the launch covers this rank's tiles plus an 8-pixel apron (mapPixelApron) — used with spatial reuse on
// a tile partition, where pass 2 needs the pass-1 reservoirs of pixels up to 5 px outside the rank's tiles; the apron
// pixels are computed redundantly by the rank that owns them, with identical results.
template <bool COUNT>
__global__ __launch_bounds__(256) void k_restir_pass1(DScene s, DCamera cam, PixelMap pm, int looper, int iter,
                                                      RestirArgs a, float *__restrict__ directIllum, int apronBlocks) {
    unsigned wg;
    const unsigned nBlocks = apronBlocks > 0 ? (unsigned)apronBlocks : (unsigned)pm.numBlocks;
    unsigned lane = threadIdx.x & 63u;
    // single-wave workgroups (64 threads, one 8x8 block each; the register bound stays that of 256 threads): a four-wave
    // workgroup keeps its slots until the slowest of its waves has ended — 2.89 -> 2.81 ms for both passes on the teapots config
    bool wgValid = xcdSwizzle(blockIdx.x, nBlocks, wg);
    unsigned blk = wgValid ? wg : 0xffffffffu / 64u;
    Pix px = apronBlocks > 0 ? mapPixelApron(pm, blk, lane) : mapPixel(pm, blk, lane);
    px.valid = px.valid && wgValid && blk < nBlocks;
    WalkStats ws{0, 0};
    unsigned nClosest = 0, nAny = 0, nHits = 0;
    const bool doSpatial = (a.reuseMask & 2) != 0;
    if (px.valid) {
        int idx = px.index;
        Sampler rng = makeSeededRandomEngine(looper, idx, 0, s.sobol);
        Ray ray = cameraSample(cam, px.x, px.y, sample4D(rng));
        HitRec h = traceClosest<COUNT>(s, ray, ws);
        nClosest++;
        int status = -1;
        Surface isec;
        isec.norm = mk3(0.f);
        isec.wo = mk3(0.f);
        Reservoir reservoir = emptyReservoir();
        v3 missDirect = mk3(0.f);
        float texMetallic = 0.f, texRoughness = 0.f;
        if (h.prim == -1 && hasEnvMap(s)) missDirect = envLookup(s, ray.d);  // :117-122
        if (h.prim != -1) {
            nHits++;
            fetchSurface(s, h.prim, h.bary, isec);
            Material material = texturedMaterial(s, isec);
            material.baseColor = mk3(1.f);  // :125
            texMetallic = material.metallic;
            texRoughness = material.roughness;
            if (material.type == Light) {
                status = -2;
            } else {
                status = isec.matId;
                isec.wo = -ray.d;
                bool deltaBSDF = (material.type == Dielectric);
                if (!deltaBSDF && dot(isec.norm, isec.wo) < 0.f) isec.norm = -isec.norm;
                for (int i = 0; i < a.risCount; ++i) {  // :139-156
                    v3 Li = mk3(0.f), wi = mk3(0.f);  // defined instead of uninitialised
                    float dist = 0.f;
                    v4 r4 = sample4D(rng);
                    float lightPdf = sampleDirectLightNoVisibility(s, isec.pos, r4, Li, wi, dist);
                    v3 bsdf = Li * materialBSDF(material, isec.norm, isec.wo, wi) * satDot(isec.norm, wi);
                    float weight = length(bsdf / lightPdf);
                    if (isNanOrInf(weight) || lightPdf <= 0.f) weight = 0.f;
                    resvUpdate(reservoir, LightLiSample{Li, wi, dist}, weight, rng.sample(), a.faithfulRIS != 0);
                }
                LightLiSample smp = reservoir.sample;
                nAny++;
                if (traceOccluded<COUNT>(s, isec.pos, isec.pos + smp.wi * smp.dist, ws)) reservoir.weight = 0.f;  // :158-163
                if (!a.firstFrame && (a.reuseMask & 1)) {  // :165-170
                    Reservoir temporal = findTemporalNeighbor(a, idx);
                    if (!resvInvalid(temporal)) resvPreClampedMerge(reservoir, temporal, rng.sample(), a.temporalClamp);
                }
                Reservoir tempReservoir = reservoir;
                if (doSpatial) {
                    resvCheckValidity(reservoir);
                    storeReservoir(a.reservoirTemp, idx, reservoir);  // :176-177
                }
                resvCheckValidity(tempReservoir);
                storeReservoir(a.reservoirOut, idx, tempReservoir);  // :186-187 (not read again this frame)
            }
        }
        if (doSpatial) {
            a.state[3 * (long long)idx + 0] = make_float4(isec.norm.x, isec.norm.y, isec.norm.z, isec.wo.x);
            a.state[3 * (long long)idx + 1] =
                make_float4(isec.wo.y, isec.wo.z, __uint_as_float(rng.scramble), __int_as_float(rng.ptr));
            // a miss carries its env-map colour in the norm/wo slots (unused for a miss)
            if (status == -1) a.state[3 * (long long)idx + 0] = make_float4(missDirect.x, missDirect.y, missDirect.z, 0.f);
            a.state[3 * (long long)idx + 2] = make_float4(__int_as_float(status), texMetallic, texRoughness, 0.f);
        } else {
            restirFinish(s, a, idx, status, isec.norm, isec.wo, reservoir, rng, px.x, px.y, false, directIllum, iter,
                         missDirect, texMetallic, texRoughness, px.out);
        }
    }
    if (COUNT) flushCounters(s.counters, nClosest, nAny, nHits, ws);
}

__global__ __launch_bounds__(256) void k_restir_pass2(DScene s, PixelMap pm, int iter, RestirArgs a,
                                                      float *__restrict__ directIllum) {
    unsigned wg;
    bool wgValid = xcdSwizzle(blockIdx.x, (unsigned)(pm.numBlocks + 3) >> 2, wg);
    unsigned lane = threadIdx.x & 63u;
    Pix px = mapPixel(pm, wgValid ? wg * 4u + (threadIdx.x >> 6) : 0xffffffffu / 64u, lane);
    if (!(px.valid && wgValid)) return;
    int idx = px.index;
    float4 s0 = a.state[3 * (long long)idx + 0], s1 = a.state[3 * (long long)idx + 1], s2 = a.state[3 * (long long)idx + 2];
    int status = __float_as_int(s2.x);
    Sampler rng{s.sobol, __float_as_uint(s1.z), __float_as_int(s1.w)};
    Reservoir reservoir = (status >= 0) ? loadReservoir(a.reservoirTemp, idx) : emptyReservoir();
    restirFinish(s, a, idx, status, mk3(s0.x, s0.y, s0.z), mk3(s0.w, s1.x, s1.y), reservoir, rng, px.x, px.y, true,
                 directIllum, iter, status == -1 ? mk3(s0.x, s0.y, s0.z) : mk3(0.f), s2.y, s2.z, px.out);
}

// =====================================================================================================================
// Pass 1 as five launches (the default since round 2): the two WALKS leave the pixel kernel.
//
//   k_restir_raygen   jittered primary ray of every pixel of the launch domain           -> ray list    (24 B / slot)
//   k_walk_packet     closest hit of the list, one 8x8 block per wave as a PACKET        -> hit records (16 B)
//                     (traverse.h, packetWalk; RDH_PT_NO_PACKETS: the lane-refill walker k_walk_pair / k_walk_persistent)
//   k_restir_ris      surface fetch, 32-candidate RIS from an LDS-resident light table   -> raw reservoir, shadow segment, state
//   k_walk_pair       any hit of the shadow segments (lane refill, sibling pairs)        -> occlusion flags (4 B)
//   k_restir_resolve  visibility, temporal merge, the two reservoir stores (restir.cu:158-187)
//
// In the fused kernel (k_restir_pass1 above, still there behind RDH_PT_RESTIR_FUSED) a pixel keeps its lane for the whole
// launch: every wave waits for its longest primary ray and again for its longest shadow ray, at the 4-5 waves per SIMD the
// RIS / BSDF code's registers allow (2.67 ms of the 3.69-ms config-4 frame in round 1, 2.0 ms of it the two walks).  The
// walk-only kernel refills lanes and runs at 8 waves per SIMD; the RIS kernel has no walk in it.  Every pixel still executes
// restir.cu:111-187 in order — same draws, same arithmetic — so images and reservoirs are bit-identical to the fused kernel
// and to the oracle.  A "slot" is the linear index block * 64 + lane of the launch domain (this rank's 8x8 blocks, or its
// apron blocks); everything handed from launch to launch is indexed by slot, not by pixel, because apron blocks of
// neighbouring own tiles can cover the same pixel twice.
// =====================================================================================================================
__global__ __launch_bounds__(256) void k_light_precompute(const LightRec *__restrict__ lights, LightPre *__restrict__ out, int n,
                                                          float sumLightPowerInv) {
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const float4 A = lights[i].a, B = lights[i].b, C = lights[i].c;
    const v3 v0 = mk3(A.x, A.y, A.z), v1 = mk3(A.w, B.x, B.y), v2_ = mk3(B.z, B.w, C.x);
    const v3 normal = triangleNormal(v0, v1, v2_);
    const float area = triangleArea(v0, v1, v2_);
    const float power = luminance(mk3(C.y, C.z, C.w)) / (area * 2.f * PI_F);  // scene.h:489
    out[i].a = A;
    out[i].b = B;
    out[i].c = C;
    out[i].d = make_float4(normal.x, normal.y, normal.z, power * sumLightPowerInv);
}

// Is this ray one the walker sets aside (k_walk_persistent<.., DEFER>)?  Its slab-test class, from the ray the walker will build.
RD_DEV bool restirRayIsLiteral(v3 a, v3 b, bool any) {
    Ray ray;
    if (any) {  // testOcclusion's set-up (scene.h:304-311)
        v3 dir = b - a;
        float dist = length(dir);
        dir = dir / dist;
        ray = makeOffsetedRay(a, dir);
    } else {
        ray = Ray{a, b};
    }
    return raySetAside(makeRaySlab(ray).cls);
}
constexpr int kRestirDeferCap = 256;  // == kWalkDeferCap (kernels_walk.h)

struct RestirSplit {  // per-slot scratch between the launches of pass 1
    float *rays;       // 6 floats: origin, direction (NaN origin.x: no ray)
    int4 *hits;        // rdh_hit of the primary ray
    float *segs;       // 6 floats: shadow segment x, y (NaN: none)
    int *occ;          // 1 = the shadow segment is occluded
    float *rawResv;    // 9 floats: the reservoir after RIS, before visibility
    float4 *st;        // 3 x float4, as RestirArgs::state
    int *deferCount;   // [0] literal-class primary rays, [1] literal-class shadow segments of this frame
    int *deferList;    // [0 .. cap) slots of the primary rays, [cap .. 2 cap) of the shadow segments
};

RD_DEV Pix restirPixel(const PixelMap &pm, unsigned blk, unsigned lane, int apronBlocks) {
    const unsigned nBlocks = apronBlocks > 0 ? (unsigned)apronBlocks : (unsigned)pm.numBlocks;
    Pix px = apronBlocks > 0 ? mapPixelApron(pm, blk, lane) : mapPixel(pm, blk, lane);
    px.valid = px.valid && blk < nBlocks;
    return px;
}

__global__ __launch_bounds__(256) void k_restir_raygen(DScene s, DCamera cam, PixelMap pm, int looper, int apronBlocks,
                                                       float *__restrict__ rays, int *__restrict__ deferCount,
                                                       int *__restrict__ deferList) {
    const unsigned blk = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const unsigned nBlocks = apronBlocks > 0 ? (unsigned)apronBlocks : (unsigned)pm.numBlocks;
    if (blk >= nBlocks) return;
    const Pix px = restirPixel(pm, blk, lane, apronBlocks);
    float *o = rays + 6ll * (blk * 64u + lane);
    if (!px.valid) {
        o[0] = __builtin_nanf("");
        return;
    }
    Sampler rng = makeSeededRandomEngine(looper, px.index, 0, s.sobol);
    const Ray ray = cameraSample(cam, px.x, px.y, sample4D(rng));  // restir.cu:111-113
    o[0] = ray.o.x; o[1] = ray.o.y; o[2] = ray.o.z;
    o[3] = ray.d.x; o[4] = ray.d.y; o[5] = ray.d.z;
    if (s.bvhSize != 0 && restirRayIsLiteral(ray.o, ray.d, false)) {
        const int at = atomicAdd(&deferCount[0], 1);
        if (at < kRestirDeferCap) deferList[at] = int(blk * 64u + lane);
    }
}

// sampleDirectLightNoVisibility (scene.h:458-492) with the light's record, plain normal and area pdf read from LDS.
RD_DEV float risCandidateStaged(const float4 *ldsLights, int lightId, v3 pos, float rz, float rw, v3 &radiance, v3 &wi, float &dist) {
    const float4 A = ldsLights[4 * lightId], B = ldsLights[4 * lightId + 1], C = ldsLights[4 * lightId + 2], D = ldsLights[4 * lightId + 3];
    const v3 v0 = mk3(A.x, A.y, A.z), v1 = mk3(A.w, B.x, B.y), v2_ = mk3(B.z, B.w, C.x);
    const v3 sampled = sampleTriangleUniform(v0, v1, v2_, rz, rw);
    const v3 normal = mk3(D.x, D.y, D.z);
    const v3 posToSampled = sampled - pos;
    if (dot(normal, posToSampled) > -1e-6f) return INVALID_PDF;
    radiance = mk3(C.y, C.z, C.w);
    // wi = normalize(posToSampled), dist = length(posToSampled) and pdfAreaToSolidAngle(D.w, pos, sampled, normal) share one dot
    // product, one square root and one reciprocal (bsdf.h, pdfAreaToSolidAngleFrom: why the bits are those of three separate
    // evaluations) — 36 of the ~290 VALU instructions of a candidate in this VALU-bound loop.
    const float d2 = dot(posToSampled, posToSampled);
    dist = __builtin_sqrtf(d2);
    wi = posToSampled * (1.f / dist);
    return pdfAreaToSolidAngleFrom(D.w, d2, normal, wi);
}

constexpr int kRisThreads = 512;
// STAGED: the alias table (8 B per entry) and the LightPre records (64 B per light) sit in LDS (1 026 lights: 74 KB; two
// 512-thread workgroups per CU), staged once per workgroup; the workgroups are persistent (grid-stride over the 8x8 blocks).
// Not STAGED (tables larger than kRisLdsBytes): the same loop on the global tables.
constexpr unsigned kRisLdsBytes = 76u * 1024u;
template <bool STAGED>
__global__ __launch_bounds__(kRisThreads) void k_restir_ris(DScene s, DCamera cam, PixelMap pm, int looper, RestirArgs a, int apronBlocks,
                                                            RestirSplit sp) {
    extern __shared__ float4 ldsRaw[];
    const int nLights = s.lightSamplerLength - (s.envSamplerLength != 0 ? 1 : 0);
    const float4 *ldsLights = ldsRaw;
    const AliasRec *ldsAlias = reinterpret_cast<const AliasRec *>(ldsRaw + 4 * nLights);
    if (STAGED) {
        const float4 *src = reinterpret_cast<const float4 *>(s.lightPre);
        for (int i = int(threadIdx.x); i < 4 * nLights; i += kRisThreads) ldsRaw[i] = src[i];
        AliasRec *dst = reinterpret_cast<AliasRec *>(ldsRaw + 4 * nLights);
        for (int i = int(threadIdx.x); i < s.lightSamplerLength; i += kRisThreads) dst[i] = s.lightAlias[i];
        __syncthreads();
    }
    const unsigned nBlocks = apronBlocks > 0 ? (unsigned)apronBlocks : (unsigned)pm.numBlocks;
    const unsigned lane = threadIdx.x & 63u;
    for (unsigned blk = blockIdx.x * (kRisThreads / 64) + (threadIdx.x >> 6); blk < nBlocks; blk += gridDim.x * (kRisThreads / 64)) {
        const Pix px = restirPixel(pm, blk, lane, apronBlocks);
        const long long slot = (long long)blk * 64 + lane;
        float *seg = sp.segs + 6 * slot;
        seg[0] = __builtin_nanf("");  // no shadow ray unless set below
        if (!px.valid) continue;
        Sampler rng = makeSeededRandomEngine(looper, px.index, 0, s.sobol);
        const Ray ray = cameraSample(cam, px.x, px.y, sample4D(rng));
        const int4 hr = sp.hits[slot];
        int status = -1;
        Surface isec;
        isec.norm = mk3(0.f);
        isec.wo = mk3(0.f);
        v3 missDirect = mk3(0.f);
        float texMetallic = 0.f, texRoughness = 0.f;
        if (hr.x == -1 && hasEnvMap(s)) missDirect = envLookup(s, ray.d);  // restir.cu:117-122
        if (hr.x != -1) {
            fetchSurface(s, hr.x, mk2(__int_as_float(hr.y), __int_as_float(hr.z)), isec);
            Material material = texturedMaterial(s, isec);
            material.baseColor = mk3(1.f);  // :125
            texMetallic = material.metallic;
            texRoughness = material.roughness;
            if (material.type == Light) {
                status = -2;
            } else {
                status = isec.matId;
                isec.wo = -ray.d;
                const bool deltaBSDF = (material.type == Dielectric);
                if (!deltaBSDF && dot(isec.norm, isec.wo) < 0.f) isec.norm = -isec.norm;
                Reservoir reservoir = emptyReservoir();
                for (int i = 0; i < a.risCount; ++i) {  // :139-156
                    v3 Li = mk3(0.f), wi = mk3(0.f);  // defined instead of uninitialised
                    float dist = 0.f;
                    const v4 r4 = sample4D(rng);
                    float lightPdf;
                    if (STAGED) {
                        lightPdf = INVALID_PDF;
                        if (s.lightSamplerLength != 0) {
                            const int length = s.lightSamplerLength;  // DevDiscreteSampler1D::sample, sampler.h:204-208
                            const int passId = imin(int(float(length) * r4.x), length - 1);
                            const AliasRec d = ldsAlias[passId];
                            const int lightId = (r4.y < d.prob) ? passId : d.failId;
                            if (lightId == length - 1 && s.envSamplerLength != 0)  // the environment map: not staged
                                lightPdf = sampleDirectLightNoVisibility(s, isec.pos, r4, Li, wi, dist);
                            else
                                lightPdf = risCandidateStaged(ldsLights, lightId, isec.pos, r4.z, r4.w, Li, wi, dist);
                        }
                    } else {
                        lightPdf = sampleDirectLightNoVisibility(s, isec.pos, r4, Li, wi, dist);
                    }
                    const v3 bsdf = Li * materialBSDF(material, isec.norm, isec.wo, wi) * satDot(isec.norm, wi);
                    float weight = length(bsdf / lightPdf);
                    if (isNanOrInf(weight) || lightPdf <= 0.f) weight = 0.f;
                    resvUpdate(reservoir, LightLiSample{Li, wi, dist}, weight, rng.sample(), a.faithfulRIS != 0);
                }
                const LightLiSample smp = reservoir.sample;
                const v3 target = isec.pos + smp.wi * smp.dist;  // the shadow segment of :160-163
                seg[0] = isec.pos.x; seg[1] = isec.pos.y; seg[2] = isec.pos.z;
                seg[3] = target.x; seg[4] = target.y; seg[5] = target.z;
                if (s.bvhSize != 0 && restirRayIsLiteral(isec.pos, target, true)) {
                    const int at = atomicAdd(&sp.deferCount[1], 1);
                    if (at < kRestirDeferCap) sp.deferList[kRestirDeferCap + at] = int(slot);
                }
                storeReservoir(sp.rawResv, slot, reservoir);
            }
        }
        sp.st[3 * slot + 0] = (status == -1) ? make_float4(missDirect.x, missDirect.y, missDirect.z, 0.f)
                                            : make_float4(isec.norm.x, isec.norm.y, isec.norm.z, isec.wo.x);
        sp.st[3 * slot + 1] = make_float4(isec.wo.y, isec.wo.z, __uint_as_float(rng.scramble), __int_as_float(rng.ptr));
        sp.st[3 * slot + 2] = make_float4(__int_as_float(status), texMetallic, texRoughness, 0.f);
    }
}

__global__ __launch_bounds__(256) void k_restir_resolve(DScene s, PixelMap pm, int iter, RestirArgs a, int apronBlocks, RestirSplit sp,
                                                        float *__restrict__ directIllum) {
    const unsigned blk = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const Pix px = restirPixel(pm, blk, lane, apronBlocks);
    if (!px.valid) return;
    const long long slot = (long long)blk * 64 + lane;
    const int idx = px.index;
    const bool doSpatial = (a.reuseMask & 2) != 0;
    const float4 s0 = sp.st[3 * slot + 0], s1 = sp.st[3 * slot + 1], s2 = sp.st[3 * slot + 2];
    const int status = __float_as_int(s2.x);
    Sampler rng{s.sobol, __float_as_uint(s1.z), __float_as_int(s1.w)};
    Reservoir reservoir = emptyReservoir();
    if (status >= 0) {
        reservoir = loadReservoir(sp.rawResv, slot);
        // :158-163.  A segment whose first float is NaN (a NaN hit position) is an empty slot to the walker: no flag was written for
        // it, and DevScene::testOcclusion of such a segment fails the root's box test — not occluded
        const float segX = sp.segs[6 * slot];
        if (segX == segX && sp.occ[slot] != 0) reservoir.weight = 0.f;
        if (!a.firstFrame && (a.reuseMask & 1)) {       // :165-170
            Reservoir temporal = findTemporalNeighbor(a, idx);
            if (!resvInvalid(temporal)) resvPreClampedMerge(reservoir, temporal, rng.sample(), a.temporalClamp);
        }
        Reservoir tempReservoir = reservoir;
        if (doSpatial) {
            resvCheckValidity(reservoir);
            storeReservoir(a.reservoirTemp, idx, reservoir);  // :176-177
        }
        resvCheckValidity(tempReservoir);
        storeReservoir(a.reservoirOut, idx, tempReservoir);  // :186-187
    }
    if (doSpatial) {
        a.state[3 * (long long)idx + 0] = s0;
        a.state[3 * (long long)idx + 1] = make_float4(s1.x, s1.y, __uint_as_float(rng.scramble), __int_as_float(rng.ptr));
        a.state[3 * (long long)idx + 2] = s2;
    } else {
        restirFinish(s, a, idx, status, mk3(s0.x, s0.y, s0.z), mk3(s0.w, s1.x, s1.y), reservoir, rng, px.x, px.y, false, directIllum,
                     iter, status == -1 ? mk3(s0.x, s0.y, s0.z) : mk3(0.f), s2.y, s2.z, px.out);
    }
}

}  // namespace rd
