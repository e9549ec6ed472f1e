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

}  // namespace rd
