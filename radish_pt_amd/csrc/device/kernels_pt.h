// radish_pt_amd/csrc/device/kernels_pt.h — one-lane-per-pixel kernels: ray batches, the megakernel path tracer,
// the direct-lighting tracer and the G-buffer.
//
// These keep the reference's kernel structure (singleKernelPT / PTDirectKernel / renderGBuffer: one thread per
// pixel, 8x8 pixels per wave64) and are the "naive path tracer, no sort" of BASELINE config 2; the queue-based
// pipeline is in kernels_wave.h and produces identical pixels.
#pragma once
#include "lights.h"

namespace rd {

struct Pix {
    int x, y;
    int index;  // y*W+x: the RNG key (makeSeededRandomEngine's `index`) — frame-global on every rank
    int out;    // where this pixel lives in the image arguments of this launch
    bool valid;
};

// 6-bit Morton decode: consecutive 8x8 blocks of a tile form 2x2, 4x4, ... squares (coherent rays per workgroup).
RD_DEV void morton6(unsigned m, unsigned &bx, unsigned &by) {
    bx = (m & 1) | ((m >> 1) & 2) | ((m >> 2) & 4);
    by = ((m >> 1) & 1) | ((m >> 2) & 2) | ((m >> 3) & 4);
}

// Workgroups are dealt round-robin over the 8 XCDs, each with its own L2 (MI355X_MICROARCH.md "Workgroup
// dispatch"): give XCD k the k-th contiguous eighth of the work list so neighbouring tiles share an L2.
// The grid is padded to 8*chunks workgroups; the function returns false for the padding.
RD_DEV bool xcdSwizzle(unsigned wg, unsigned numWork, unsigned &logical) {
    unsigned chunks = (numWork + 7u) >> 3;
    logical = (wg & 7u) * chunks + (wg >> 3);
    return logical < numWork;
}

RD_DEV Pix mapPixel(const PixelMap &pm, unsigned block, unsigned lane) {
    Pix p;
    unsigned bpe = (unsigned)pm.tile >> 3;  // 8x8 blocks per tile edge
    unsigned bpt = bpe * bpe;
    unsigned localTile = block / bpt;
    unsigned b = block - localTile * bpt;
    unsigned bx, by;
    if (bpe == 8u) {
        morton6(b, bx, by);
    } else {
        bx = b % bpe;
        by = b / bpe;
    }
    unsigned tileId = localTile * (unsigned)pm.world + (unsigned)pm.rank;
    unsigned tx = tileId % (unsigned)pm.tilesX, ty = tileId / (unsigned)pm.tilesX;
    unsigned lx = bx * 8u + (lane & 7u), ly = by * 8u + (lane >> 3);
    p.x = int(tx * (unsigned)pm.tile + lx);
    p.y = int(ty * (unsigned)pm.tile + ly);
    p.valid = (block < (unsigned)pm.numBlocks) && (tileId < (unsigned)pm.numTiles) && p.x < pm.W && p.y < pm.H;
    p.index = p.y * pm.W + p.x;
    p.out = pm.packed ? int(localTile * (unsigned)(pm.tile * pm.tile) + ly * (unsigned)pm.tile + lx) : p.index;
    return p;
}

// The pixel of this lane when the launch has one single-wave workgroup per 8x8 block (64 threads, gridBlocks(pm) workgroups):
// XCD x gets a contiguous range of blocks.  (Four-wave workgroups kept their slots until the slowest of their waves had ended.)
RD_DEV Pix mapWavePixel(const PixelMap &pm) {
    unsigned blk;
    const bool ok = xcdSwizzle(blockIdx.x, (unsigned)pm.numBlocks, blk);
    Pix px = mapPixel(pm, ok ? blk : 0xffffffffu / 64u, threadIdx.x & 63u);
    px.valid = px.valid && ok;
    return px;
}

RD_DEV v3 load3(const float *img, int i) { return mk3(img[3 * i], img[3 * i + 1], img[3 * i + 2]); }
RD_DEV void store3(float *img, int i, v3 v) {
    img[3 * i] = v.x;
    img[3 * i + 1] = v.y;
    img[3 * i + 2] = v.z;
}
RD_DEV void storeRunningMean(float *img, int i, v3 v, int iter) {  // pathtrace.cu:287-290
    store3(img, i, (load3(img, i) * float(iter) + v) / float(iter + 1));
}

// ---- ray batches (tests / roofline bench) ---------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(256) void k_trace_closest(DScene s, const float *__restrict__ rays, long long n,
                                                       int4 *__restrict__ hits) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    WalkStats ws{0, 0};
    bool valid = i < n;
    HitRec h;
    h.prim = -1;
    if (valid) {
        Ray r{mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5])};
        h = traceClosest<COUNT>(s, r, ws);
        bool hit = h.prim != -1;
        hits[i] = make_int4(h.prim, __float_as_int(hit ? h.bary.x : 0.f), __float_as_int(hit ? h.bary.y : 0.f),
                            __float_as_int(hit ? h.dist : 3.402823466e+38f));
    }
    if (COUNT) flushCounters(s.counters, valid ? 1u : 0u, 0u, (valid && h.prim != -1) ? 1u : 0u, ws);
}

template <bool COUNT>
__global__ __launch_bounds__(256) void k_trace_occluded(DScene s, const float *__restrict__ seg, long long n,
                                                        int *__restrict__ occluded) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    WalkStats ws{0, 0};
    bool valid = i < n;
    if (valid) {
        bool occ = traceOccluded<COUNT>(s, mk3(seg[6 * i], seg[6 * i + 1], seg[6 * i + 2]),
                                        mk3(seg[6 * i + 3], seg[6 * i + 4], seg[6 * i + 5]), ws);
        occluded[i] = occ ? 1 : 0;
    }
    if (COUNT) flushCounters(s.counters, 0u, valid ? 1u : 0u, 0u, ws);
}

// Optional ray dump of a frame (rdh_dump_rays): every ray the frame traces is appended to one of two lists, in the layout
// rdh_trace_closest / rdh_trace_occluded take — {origin, direction} per closest-hit ray, {x, y} per occlusion segment — so the
// walk-only kernel can be timed, and the CPU traversal baseline run, on exactly the frame's own rays (bench.py
// roofline.traversal_only).  Rays beyond a list's capacity are counted but not stored.
struct RayDump {
    float *closest, *any;            // nullptr: no dump
    long long capClosest, capAny;
    unsigned long long *count;       // [0] closest-hit rays, [1] occlusion segments
};
RD_DEV void dumpRay(const RayDump &d, int which, v3 a, v3 b) {
    const unsigned long long at = atomicAdd(&d.count[which], 1ull);
    float *dst = which ? d.any : d.closest;
    if ((long long)at < (which ? d.capAny : d.capClosest)) {
        dst[6 * at] = a.x; dst[6 * at + 1] = a.y; dst[6 * at + 2] = a.z;
        dst[6 * at + 3] = b.x; dst[6 * at + 4] = b.y; dst[6 * at + 5] = b.z;
    }
}

// ---- singleKernelPT (pathtrace.cu:149-291) ----------------------------------------------------------------
template <bool COUNT, bool DUMP = false>
__global__ __launch_bounds__(256) void k_path_trace_mega(DScene s, DCamera cam, PixelMap pm, int looper, int iter,
                                                         int maxDepth, float *__restrict__ directIllum,
                                                         float *__restrict__ indirectIllum, RayDump dump = RayDump{}) {
    Pix px = mapWavePixel(pm);  // single-wave workgroups: 64 threads, one 8x8 block each
    WalkStats ws{0, 0};
    unsigned nClosest = 0, nAny = 0, nHits = 0;

    if (px.valid) {
        v3 direct = mk3(0.f), indirect = mk3(0.f);
        Sampler rng = makeSeededRandomEngine(looper, px.index, 0, s.sobol);
        Ray ray = cameraSample(cam, px.x, px.y, sample4D(rng));
        if (DUMP) dumpRay(dump, 0, ray.o, ray.d);
        HitRec h = traceClosest<COUNT>(s, ray, ws);
        nClosest++;
        do {  // `goto WriteRadiance`
            if (h.prim == -1) {
                direct = mk3(1.f);
                break;
            }
            nHits++;
            Surface isec;
            fetchSurface(s, h.prim, h.bary, isec);
            Material material = texturedMaterial(s, isec);
            material.baseColor = mk3(1.f);  // DENOISER_DEMODULATE (:175-178)
            if (material.type == Light) {
                direct = mk3(1.f);
                break;
            }
            v3 throughput = mk3(1.f);
            isec.wo = -ray.d;
            for (int depth = 1; depth <= maxDepth; depth++) {
                bool deltaBSDF = (material.type == Dielectric);
                if (material.type != Dielectric && dot(isec.norm, isec.wo) < 0.f) isec.norm = -isec.norm;
                if (!deltaBSDF) {
                    v4 r4 = sample4D(rng);
                    float lightPdf = INVALID_PDF;
                    v3 radiance = mk3(0.f), wi = mk3(0.f);
                    if (s.lightSamplerLength != 0) {  // sampleDirectLight (scene.h:419-456)
                        LightPick lp = pickLightPoint(s, isec.pos, r4);
                        nAny++;
                        if (DUMP) dumpRay(dump, 1, isec.pos, lp.sampled);
                        bool occ = traceOccluded<COUNT>(s, isec.pos, lp.sampled, ws);
                        if (!occ) lightPdf = lightPdfUnoccluded(s, isec.pos, lp, radiance, wi);
                    }
                    if (lightPdf > 0.f) {
                        float BSDFPdf = materialPdf(material, isec.norm, isec.wo, wi);
                        v3 c = throughput * materialBSDF(material, isec.norm, isec.wo, wi) * radiance *
                               satDot(isec.norm, wi) / lightPdf * powerHeuristic(lightPdf, BSDFPdf);
                        if (depth == 1) direct = direct + c;
                        else indirect = indirect + c;
                    }
                }
                BSDFSample sample;
                sample.pdf = 0.f;
                materialSample(material, isec.norm, isec.wo, sample3D(rng), sample);
                if (sample.type == Invalid) break;
                else if (sample.pdf < 1e-8f) break;
                bool deltaSample = (sample.type & Specular) != 0;
                throughput = throughput * (sample.bsdf / sample.pdf * (deltaSample ? 1.f : absDot(isec.norm, sample.dir)));
                ray = makeOffsetedRay(isec.pos, sample.dir);
                v3 curPos = isec.pos;
                if (DUMP) dumpRay(dump, 0, ray.o, ray.d);
                h = traceClosest<COUNT>(s, ray, ws);
                nClosest++;
                if (h.prim == -1) {  // :232-247
                    if (hasEnvMap(s)) {
                        v3 radiance = envLookup(s, ray.d) * throughput;
                        float weight = deltaSample ? 1.f : powerHeuristic(sample.pdf, environmentMapPdf(s, ray.d));
                        indirect = indirect + radiance * weight;
                    }
                    break;
                }
                nHits++;
                fetchSurface(s, h.prim, h.bary, isec);
                isec.wo = -ray.d;
                material = texturedMaterial(s, isec);
                if (material.type == Light) {
                    if (dot(isec.norm, ray.d) < 0.f) break;  // SCENE_LIGHT_SINGLE_SIDED (:252-256)
                    v3 radiance = material.baseColor;
                    float weight = deltaSample
                                       ? 1.f
                                       : powerHeuristic(sample.pdf,
                                                        pdfAreaToSolidAngle(luminance(radiance) * s.sumLightPowerInv *
                                                                                getPrimitiveArea(s, isec.primId),
                                                                            curPos, isec.pos, isec.norm));
                    indirect = indirect + radiance * throughput * weight;
                    break;
                }
            }
        } while (false);
        if (hasNanOrInf(direct)) direct = mk3(0.f);
        if (hasNanOrInf(indirect)) indirect = mk3(0.f);
        direct = HDRToLDR(direct);
        indirect = HDRToLDR(indirect);
        storeRunningMean(directIllum, px.out, direct, iter);
        storeRunningMean(indirectIllum, px.out, indirect, iter);
    }
    if (COUNT) flushCounters(s.counters, nClosest, nAny, nHits, ws);
}

// ---- PTDirectKernel (pathtrace.cu:293-345) ------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(256) void k_path_trace_direct(DScene s, DCamera cam, PixelMap pm, int looper, int iter,
                                                           float *__restrict__ directIllum) {
    Pix px = mapWavePixel(pm);  // single-wave workgroups: 64 threads, one 8x8 block each
    WalkStats ws{0, 0};
    unsigned nClosest = 0, nAny = 0, nHits = 0;
    if (px.valid) {
        v3 direct = mk3(0.f);
        Sampler rng = makeSeededRandomEngine(looper, px.index, 0, s.sobol);
        Ray ray = cameraSample(cam, px.x, px.y, sample4D(rng));
        HitRec h = traceClosest<COUNT>(s, ray, ws);
        nClosest++;
        do {
            if (h.prim == -1) {  // :307-312
                if (hasEnvMap(s)) direct = envLookup(s, ray.d);
                break;
            }
            nHits++;
            Surface isec;
            fetchSurface(s, h.prim, h.bary, isec);
            Material material = texturedMaterial(s, isec);
            if (material.type == Light) {
                direct = material.baseColor;
                break;
            }
            isec.wo = -ray.d;
            bool deltaBSDF = (material.type == Dielectric);
            if (!deltaBSDF && dot(isec.norm, isec.wo) < 0.f) isec.norm = -isec.norm;
            if (!deltaBSDF) {
                v4 r4 = sample4D(rng);
                float lightPdf = INVALID_PDF;
                v3 Li = mk3(0.f), wi = mk3(0.f);
                if (s.lightSamplerLength != 0) {
                    LightPick lp = pickLightPoint(s, isec.pos, r4);
                    nAny++;
                    bool occ = traceOccluded<COUNT>(s, isec.pos, lp.sampled, ws);
                    if (!occ) lightPdf = lightPdfUnoccluded(s, isec.pos, lp, Li, wi);
                }
                if (lightPdf > 0.f)
                    direct = Li * materialBSDF(material, isec.norm, isec.wo, wi) * satDot(isec.norm, wi) / lightPdf;
            }
        } while (false);
        storeRunningMean(directIllum, px.out, direct, iter);
    }
    if (COUNT) flushCounters(s.counters, nClosest, nAny, nHits, ws);
}

// ---- renderGBuffer (gBuffer.cu:3-76) --------------------------------------------------------------------------
struct GBufPtrs {
    float *albedo, *normal;
    int *motion;
    float *depth;
    int *primId;
    int width, height;
};
template <bool COUNT>
__global__ __launch_bounds__(256) void k_gbuffer(DScene s, DCamera cam, DCamera lastCam, PixelMap pm, GBufPtrs gb) {
    Pix px = mapWavePixel(pm);  // single-wave workgroups: 64 threads, one 8x8 block each
    WalkStats ws{0, 0};
    unsigned nClosest = 0, nHits = 0;
    if (px.valid) {
        int idx = px.index;
        float aspect = float(cam.resx) / float(cam.resy);
        v2 pixelsize = {1.f / float(cam.resx), 1.f / float(cam.resy)};
        v2 scr = mk2(float(px.x), float(px.y)) * pixelsize;
        v2 ruv = scr + pixelsize * mk2(0.5f, 0.5f);
        ruv = {1.f - ruv.x * 2.f, 1.f - ruv.y * 2.f};
        v3 pLens = mk3(0.f);
        v2 f = (ruv * mk2(aspect, 1.f)) * cam.tanFovY;
        v3 pFocus = mk3(f.x, f.y, 1.f) * cam.focalDist;
        v3 dir = pFocus - pLens;
        Ray ray;
        ray.o = cam.position + cam.right * pLens.x + cam.up * pLens.y;
        ray.d = normalize(mul(m3{cam.right, cam.up, cam.view}, dir));
        HitRec h = traceClosest<COUNT>(s, ray, ws);
        nClosest++;
        if (h.prim != -1) {
            nHits++;
            Surface isec;
            fetchSurface(s, h.prim, h.bary, isec);
            int matId = isec.matId;
            if (loadMaterial(s.mats, isec.matId).type == Light) matId = -2;  // NullPrimitive - 1 (:33-37, untextured record)
            Material material = texturedMaterial(s, isec);            // :44 (may perturb isec.norm)
            store3(gb.albedo, idx, material.baseColor);
            store3(gb.normal, idx, isec.norm);
            gb.primId[idx] = matId;
            gb.depth[idx] = length(ray.o - isec.pos);  // glm::distance(pos, origin) = length(origin - pos)
            v2 ndc = cameraRasterUV(lastCam, isec.pos);  // Camera::getRasterCoord (sceneStructs.h:45-48)
            int lx = (int)(float(lastCam.resx) * ndc.x), ly = (int)(float(lastCam.resy) * ndc.y);
            gb.motion[idx] = (lx >= 0 && lx < gb.width && ly >= 0 && ly < gb.height) ? ly * cam.resx + lx : -1;
        } else {
            store3(gb.albedo, idx, hasEnvMap(s) ? envLookup(s, ray.d) : mk3(0.f));  // :61-66
            store3(gb.normal, idx, mk3(0.f));
            gb.primId[idx] = -1;
            gb.depth[idx] = 1.f;
            gb.motion[idx] = 0;
        }
    }
    if (COUNT) flushCounters(s.counters, nClosest, 0u, nHits, ws);
}

// ---- tile reassembly after the all-gather (no reference counterpart) ---------------------------------------------
// `channels` 32-bit words per pixel: 3 for images, 9 for DirectReservoir records.
__global__ __launch_bounds__(256) void k_untile(const float *__restrict__ gathered, float *__restrict__ frame, int W,
                                                int H, int tile, int tilesX, int numTiles, int world, int tilesPerRank,
                                                int channels) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)W * H;
    if (i >= total) return;
    int x = int(i % W), y = int(i / W);
    int tx = x / tile, ty = y / tile;
    int tileId = ty * tilesX + tx;
    int rank = tileId % world, localTile = tileId / world;
    long long src = ((long long)rank * tilesPerRank + localTile) * (long long)(tile * tile) + (y - ty * tile) * tile + (x - tx * tile);
    for (int c = 0; c < channels; c++) frame[channels * i + c] = gathered[channels * src + c];
}
// frame layout → this rank's packed tile buffer (the send side of the reservoir exchange)
__global__ __launch_bounds__(256) void k_pack_tiles(const float *__restrict__ frame, float *__restrict__ packed, PixelMap pm,
                                                    int channels) {
    unsigned lane = threadIdx.x & 63u;
    unsigned block = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (block >= (unsigned)pm.numBlocks) return;
    PixelMap q = pm;
    q.packed = 1;
    Pix px = mapPixel(q, block, lane);
    if (!px.valid) return;
    for (int c = 0; c < channels; c++) packed[(long long)channels * px.out + c] = frame[(long long)channels * px.index + c];
}

// ---- G-buffer exchange on a tile partition -------------------------------------------------------------------------------
// Each rank renders the G-buffer records of ITS tiles (frame layout); k_gbuf_pack gathers them into one 9-float record per
// pixel of the rank's packed tile buffer — albedo.xyz, normal.xyz, motion, depth, primId (ints carried as their bits) — the
// send side of one all-gather; k_gbuf_unpack scatters the gathered records of every rank into the frame-layout planes, so
// that every rank holds the whole frame's G-buffer (ReSTIR's temporal lookup follows motion vectors to arbitrary pixels).
__global__ __launch_bounds__(256) void k_gbuf_pack(const float *__restrict__ albedo, const float *__restrict__ normal,
                                                   const int *__restrict__ motion, const float *__restrict__ depth,
                                                   const int *__restrict__ primId, float *__restrict__ packed, PixelMap pm) {
    unsigned lane = threadIdx.x & 63u;
    unsigned block = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (block >= (unsigned)pm.numBlocks) return;
    PixelMap q = pm;
    q.packed = 1;
    Pix px = mapPixel(q, block, lane);
    if (!px.valid) return;
    float *o = packed + 9ll * px.out;
    const long long i = px.index;
    o[0] = albedo[3 * i]; o[1] = albedo[3 * i + 1]; o[2] = albedo[3 * i + 2];
    o[3] = normal[3 * i]; o[4] = normal[3 * i + 1]; o[5] = normal[3 * i + 2];
    o[6] = __int_as_float(motion[i]);
    o[7] = depth[i];
    o[8] = __int_as_float(primId[i]);
}
__global__ __launch_bounds__(256) void k_gbuf_unpack(const float *__restrict__ gathered, float *__restrict__ albedo,
                                                     float *__restrict__ normal, int *__restrict__ motion,
                                                     float *__restrict__ depth, int *__restrict__ primId, int W, int H, int tile,
                                                     int tilesX, int world, int tilesPerRank) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)W * H) return;
    int x = int(i % W), y = int(i / W);
    int tx = x / tile, ty = y / tile;
    int tileId = ty * tilesX + tx;
    int rank = tileId % world, localTile = tileId / world;
    long long src = ((long long)rank * tilesPerRank + localTile) * (long long)(tile * tile) + (y - ty * tile) * tile + (x - tx * tile);
    const float *r = gathered + 9ll * src;
    albedo[3 * i] = r[0]; albedo[3 * i + 1] = r[1]; albedo[3 * i + 2] = r[2];
    normal[3 * i] = r[3]; normal[3 * i + 1] = r[4]; normal[3 * i + 2] = r[5];
    motion[i] = __float_as_int(r[6]);
    depth[i] = r[7];
    primId[i] = __float_as_int(r[8]);
}

// This rank's tiles grown by one 8-pixel block on every side (>= the 5-pixel radius of ReSTIR's spatial reuse,
// restir.cu:45): the pixels whose pass-1 reservoirs this rank's pass 2 may gather.  `out` is unused (frame layout).
RD_DEV Pix mapPixelApron(const PixelMap &pm, unsigned block, unsigned lane) {
    Pix p;
    unsigned bpe = ((unsigned)pm.tile >> 3) + 2u;
    unsigned bpt = bpe * bpe;
    unsigned localTile = block / bpt;
    unsigned b = block - localTile * bpt;
    unsigned bx = b % bpe, by = b / bpe;
    unsigned tileId = localTile * (unsigned)pm.world + (unsigned)pm.rank;
    int tx = int(tileId % (unsigned)pm.tilesX), ty = int(tileId / (unsigned)pm.tilesX);
    p.x = tx * pm.tile - 8 + int(bx * 8u + (lane & 7u));
    p.y = ty * pm.tile - 8 + int(by * 8u + (lane >> 3));
    p.valid = (localTile < (unsigned)pm.tilesPerRank) && (tileId < (unsigned)pm.numTiles) && p.x >= 0 && p.y >= 0 &&
              p.x < pm.W && p.y < pm.H;
    p.index = p.y * pm.W + p.x;
    p.out = p.index;
    return p;
}

}  // namespace rd
