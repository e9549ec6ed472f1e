// radish_pt_amd/csrc/device/kernels_persist.h — pathTrace as ONE persistent launch: a per-lane state machine with
// lane refill ("wavefront in registers").
//
// Why: node visits per ray are heavy-tailed on the reference's threaded BVH (Cornell stand-in: median 1, mean 34,
// p99 ≈ 950, max ≈ 2 800 dependent steps).  A kernel boundary per bounce (kernels_wave.h) makes every bounce wait for
// its longest ray (≈0.4–0.6 ms each, nine times per frame), and a one-lane-per-pixel megakernel makes every wave wait
// for its longest path.  Here a lane that finishes a ray shades it and carries on; a lane whose path ends takes the
// next pixel; the frame's critical path is the longest single PATH, and lanes stay busy until the frame runs dry.
//
// Every lane executes exactly singleKernelPT's sequence for its pixel (/root/reference/src/pathtrace.cu:149-291):
// primary ray → [NEE sample → shadow ray → BSDF sample → extension ray]* with the same RNG draws and the same order of
// additions, so the result is bit-identical to the megakernel and to the queue pipeline.  Wave-level scheduling only
// decides WHEN a lane's next step runs:
//   box steps   while enough lanes are walking,
//   leaf tests  when >= 1/RD_PT_LEAF_DEN of the tracing lanes are parked on a leaf,
//   shading     when >= RD_SHADE_MIN 64ths of the busy lanes wait for it (or nothing else can run),
//   raygen      when >= RD_PT_REFILL_MIN lanes are idle and pixels remain.
// The three thresholds were re-tuned together at the end of round 1 (profiles/r01_f_experiment_thresholds.txt): 16 / 16 / 4
// -> 32 / 6 / 3 is 4.8 % on the Cornell frame (3.34 -> 3.18 ms) and on the teapots frame (10.24 -> 9.75 ms), +1 % on the
// 1-M-triangle 4K frame; one at a time each gave 1.5-2.4 %.  (Later shading batches are fuller and cost the same ~10 k clocks;
// idle lanes are refilled sooner; triangles are tested a little earlier.)
// Path state that is live only across phases (throughput, accumulators, the pending extension ray, …) sits in LDS,
// 84 B per lane, so the traversal loop runs on the walker's registers alone.
#pragma once
#include "kernels_pt.h"
#include "wg_trace.h"

namespace rd {

#ifndef RD_SHADE_MIN
#define RD_SHADE_MIN 32
#endif
#ifndef RD_PT_REFILL_MIN
#define RD_PT_REFILL_MIN 6
#endif
#ifndef RD_PT_LEAF_DEN  // k_pt_persistent only; the one-ray-per-lane kernels and the G-buffer keep RD_LEAF_NUM / RD_LEAF_DEN = 1 / 4
#define RD_PT_LEAF_DEN 3
#endif
// ... and once the pixel supply has ended (64ths of the busy lanes).  In the drain a wave serialises the shading calls of its
// remaining paths (≈10 k clocks each, whatever the lane count); with RD_SHADE_MIN at a quarter, waiting for half of them
// in the drain measured 1 % faster on the Cornell frame, 4 / 8 64ths 3 % / 1.5 % slower; with RD_SHADE_MIN = 32 the two coincide.
#ifndef RD_SHADE_MIN_DRAIN
#define RD_SHADE_MIN_DRAIN 32
#endif
#ifndef RD_PIX_REFILL_MIN
#define RD_PIX_REFILL_MIN 16
#endif

constexpr int kDeferCap = 256;  // rays k_gbuffer_literal takes: one per workgroup, one round
struct PersistCounters {
    int blockHead;  // next 8x8 pixel block beyond the static first round (see k_pt_persistent)
    int deferCount;  // literal-class primary rays of the frame (k_gbuffer_find_literal); above kDeferCap none is set aside
#ifdef RD_PERSIST_STAMPS  // diagnostic build only: when each wave started, ran out of pixels, and ended (wall clock);
    // [3] wall-clock ticks it spent tracing literal-class rays whole, [4] how many such rays
    unsigned long long stamp[5][4096];
#endif
#ifdef RD_PERSIST_PHASES  // diagnostic build only: where the waves' time goes (s_memtime ticks summed over waves) and how full
    // the wave is in each phase: [0..5] ticks in raygen / literal+coop / box loop / leaf / retire / shade, [6] total;
    // [8] box wave-steps, [9] box lane-steps, [10] leaf calls, [11] leaf lanes, [12] shade calls, [13] shade lanes,
    // [14] raygen calls, [15] raygen lanes
    unsigned long long phase[16];
#endif
    int deferred[kDeferCap];  // pixel indices; NOT cleared between launches (the host clears up to here)
};

constexpr int PS_IDLE = 0, PS_TRACE = 1, PS_SHADE = 2;

// ---- longest-paths-first block order ---------------------------------------------------------------------------
// A persistent launch ends one path-latency after the pixel supply runs dry (measured: supply dry at ~1.9 ms, last
// wave out at ~3.8 ms on the Cornell frame), so the blocks holding long paths must start first.  Each launch records,
// per 8x8 block, the largest number of box steps any of its paths took; the next launch visits blocks in descending
// order of that figure (frames of an animation are coherent; the first frame uses plain order).  Pure scheduling:
// which lane renders a pixel, and when, cannot change its value.
constexpr int kCostBuckets = 64;
// One workgroup of 1 024 threads (a counting sort of ~32 000 blocks into 64 buckets).  The LDS counters are bumped once per
// wave and bucket, not once per block: neighbouring blocks cost about the same, so a wave's 64 blocks fall into a handful of
// buckets, and 64 lanes adding to ONE LDS word serialise (the first version did that from 256 threads: 104 us per frame, 3 %
// of the Cornell frame, on the stream between two persistent launches; this one: see profiles/).
// `ema`: running estimate of each block's cost over the frames so far (the per-launch figure is ONE sample of the longest of
// 64 random paths — noisy; blocks differ systematically by what they look at).  ema' = (RD_EMA_KEEP * ema + cost) / (RD_EMA_KEEP + 1).
#ifndef RD_EMA_KEEP
#define RD_EMA_KEEP 7
#endif
constexpr int kScheduleThreads = 1024;
__global__ __launch_bounds__(kScheduleThreads) void k_persist_schedule(unsigned *__restrict__ cost, unsigned *__restrict__ ema,
                                                                       int *__restrict__ order, int n) {
    __shared__ int hist[kCostBuckets], base[kCostBuckets];
    const int t = int(threadIdx.x);
    const int lane = t & 63;
    if (t < kCostBuckets) hist[t] = 0;
    __syncthreads();
    auto bucketOf = [](unsigned c) {  // quarter-octave buckets, descending: bucket 0 = most expensive
        int b = 0;
        if (c > 0) {
            int lg = 31 - __clz((int)c);                         // floor(log2 c)
            int frac = lg >= 2 ? int((c >> (lg - 2)) & 3u) : 0;  // next two bits
            b = lg * 4 + frac + 1;
        }
        if (b > kCostBuckets - 1) b = kCostBuckets - 1;
        return kCostBuckets - 1 - b;
    };
    // add `1` per lane of the wave to counter[bucket] with one atomic per distinct bucket; returns the lane's slot
    auto waveAdd = [&](int *counter, int bucket, bool valid) {
        int slot = 0;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int L = __ffsll((long long)todo) - 1;
            const int b0 = __shfl(bucket, L, 64);
            const unsigned long long same = __ballot(valid && bucket == b0);
            int first = 0;
            if (lane == L) first = atomicAdd(&counter[b0], __popcll(same));
            first = __shfl(first, L, 64);
            if (valid && bucket == b0) slot = first + __popcll(same & ((1ull << lane) - 1ull));
            todo &= ~same;
        }
        return slot;
    };
    // four blocks per thread and round: the four pairs of loads are in flight together (one workgroup cannot hide a
    // dependent global load per round behind anything else)
    for (int i0 = 0; i0 < n; i0 += 4 * kScheduleThreads) {
        unsigned old[4], c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0 + j * kScheduleThreads + t;
            old[j] = i < n ? ema[i] : 0u;
            c[j] = i < n ? cost[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0 + j * kScheduleThreads + t;
            const bool valid = i < n;
            const unsigned e = old[j] == 0u ? c[j] : (RD_EMA_KEEP * old[j] + c[j] + RD_EMA_KEEP / 2) / (RD_EMA_KEEP + 1);
            if (valid) {
                ema[i] = e;
                cost[i] = 0u;
            }
            waveAdd(hist, bucketOf(e), valid);
        }
    }
    __syncthreads();
    if (t < 64) {  // exclusive prefix sum of the 64 bucket sizes
        int v = hist[t], incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (t >= d) incl += up;
        }
        base[t] = incl - v;
    }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 4 * kScheduleThreads) {  // order inside a bucket: ties are equally expensive
        unsigned e[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0 + j * kScheduleThreads + t;
            e[j] = i < n ? ema[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0 + j * kScheduleThreads + t;
            const bool valid = i < n;
            const int pos = waveAdd(base, bucketOf(e[j]), valid);
            if (valid) order[pos] = i;
        }
    }
}

#ifndef RD_PERSIST_WAVES
#define RD_PERSIST_WAVES 1
#endif
#ifdef RD_PERSIST_PHASES
#ifdef RD_PERSIST_PHASES_DRAIN  // count only what happens after this wave has found the pixel supply dry
#define PH_ON exhausted
#else
#define PH_ON true
#endif
#define PH_MARK(i) do { unsigned long long _n = __builtin_amdgcn_s_memtime(); if (PH_ON) phT[i] += _n - phLast; phLast = _n; } while (0)
#define PH_COUNT(i, mask) do { if (PH_ON) { phT[i] += 1ull; phT[(i) + 1] += (unsigned long long)__popcll(mask); } } while (0)
#else
#define PH_MARK(i) do { } while (0)
#define PH_COUNT(i, mask) do { } while (0)
#endif
// PAIRS: the per-lane walks go over the sibling pairs (DScene::pairs; traverse.h, pairStep) instead of the six threaded arrays:
// same triangle tests in the same order, same counters; chosen by the host for big scenes (radish_hip.hip, usePairs).
#ifndef RD_PERSIST_PAIR_WAVES
#define RD_PERSIST_PAIR_WAVES 3  // the pair variant is held to three waves per SIMD (173 VGPRs otherwise: two)
#endif
template <bool COUNT, bool PAIRS = false>
__global__ __launch_bounds__(64, PAIRS && !COUNT ? RD_PERSIST_PAIR_WAVES : RD_PERSIST_WAVES) void k_pt_persistent(DScene s, DCamera cam, PixelMap pm, int looper, int iter, int maxDepth,
                                                       float *__restrict__ directIllum, float *__restrict__ indirectIllum,
                                                       PersistCounters *pc, const int *__restrict__ blockOrder,
                                                       unsigned *__restrict__ blockCost, int2 *__restrict__ pairOvf = nullptr,
                                                       int pairOvfDepth = 0) {
    // ---- LDS-resident path state (SoA: lane-consecutive, conflict-free) ----
    // One wave per workgroup: nothing here needs a workgroup barrier, and a finished wave frees its CU slot at once
    // (with 4-wave workgroups the slot stays taken until the slowest of the four has drained its last path).
    __shared__ float sThr[3][64], sAccD[3][64], sAccI[3][64], sCur[3][64];
    __shared__ float sExtO[3][64], sExtD[3][64], sExtPdf[64], sNee[4][64];
    __shared__ int sFlags[64];  // bit0: extension ray pending, bit1: that sample was specular
    __shared__ int2 sStack[PAIRS ? kPairLds * 64 : 1];  // PAIRS: the lanes' stacks (traverse.h, pairPush)
    int2 *const ovf = PAIRS ? pairOvf + (size_t)blockIdx.x * (size_t)pairOvfDepth * 64 : nullptr;
    const int t = int(threadIdx.x);
    const int lane = t & 63;
    const int end = s.bvhSize;
    const int numBlocks = pm.numBlocks;
    // xcdSwizzle is a bijection on the group range padded to a multiple of 8: walk the padded range, skip the padding
    const int paddedBlocks = blockOrder ? numBlocks : int((((unsigned)(numBlocks + 3) / 4u + 7u) / 8u) * 8u * 4u);

    WalkStats ws{0, 0};
    unsigned nClosest = 0, nAny = 0, nHits = 0;
#ifdef RD_PERSIST_PHASES
    unsigned long long phT[16] = {0};
    unsigned long long phLast = __builtin_amdgcn_s_memtime();
    const unsigned long long phStart = phLast;
#endif
#ifdef RD_PERSIST_STAMPS
    const int gw = int(blockIdx.x);
    bool stampedDry = false;
    unsigned long long litTicks = 0, litRays = 0;
    if (lane == 0 && gw < 4096) pc->stamp[0][gw] = wall_clock64();
#endif

    // wave-uniform pixel reservation: one 8x8 block at a time; the first is static (wave g takes block g)
    int curBlock = int(blockIdx.x);
    int slotNext = 0;  // next unassigned pixel slot (0..63) of curBlock
    const int gridWavesN = int(gridDim.x);
    bool exhausted = curBlock >= paddedBlocks;

    // per-lane registers
    int state = PS_IDLE;
    int outIdx = 0;
    int myBlock = 0;         // 8x8 block this lane's pixel belongs to (index into blockCost)
    unsigned pathSteps = 0;  // box steps of the current path
    int k = 0;  // index of the hit the current extension ray leads to (0 = primary)
    uint32_t rngScramble = 0;
    int rngPtr = 0;
    // the ray being walked: origin / reciprocal direction as register pairs (packed slab test), direction, slab-test class
    RaySlabPk rp;
    rp.oxy = rp.ozw = rp.ixy = rp.izw = f2v{0.f, 0.f};
    v3 rayD = mk3(0.f);
    int rayCls = 0;
    auto slab = [&]() {  // the RaySlab view of it (same registers)
        RaySlab r;
        r.o = mk3(rp.oxy.x, rp.oxy.y, rp.ozw.x);
        r.d = rayD;
        r.inv = mk3(rp.ixy.x, rp.ixy.y, rp.izw.x);
        r.cls = rayCls;
        return r;
    };
    unsigned ordOfs = 0;  // byte offset of this ray's ordering inside the node allocation (layouts.h: one allocation)
    const char *nodeBase = reinterpret_cast<const char *>(s.nodes[0]);
    const unsigned ordStride = (unsigned)(s.bvhSize + 1) * (unsigned)sizeof(NodeRec);
    unsigned stepMark = 0, waveSteps = 0;  // wave-uniform count of box-loop iterations: the path's cost is how many it was in flight for
    int node = end, pending = -1;  // PAIRS: `node` is the pair to enter next (pairStep's `cur`), `ordOfs` the ordering itself
    PairStack stack{0, 0};
    float tmax = 0.f;
    int hitPrim = -1;
    v2 hitBary = mk2(0.f, 0.f);
    bool isShadow = false, occluded = false;
    // is this lane's walk over?  (threaded: the position has reached the end of the array; pairs: nothing to enter, nothing stacked)
    auto walkOver = [&]() { return PAIRS ? (node == kPairNone && stack.sp == 0) : (node == end); };

    auto startTrace = [&](const Ray &ray, float limit, bool shadow) {
        {
            const RaySlab r = makeRaySlab(ray);
            rp = packSlab(r);
            rayD = r.d;
            rayCls = r.cls;
        }
        ordOfs = (unsigned)getMTBVHId(-ray.d) * (PAIRS ? 1u : ordStride);
        stepMark = waveSteps;
        node = 0;
        pending = -1;
        hitPrim = -1;
        occluded = false;
        tmax = limit;
        isShadow = shadow;
        state = PS_TRACE;
        if (PAIRS) {
            stack = PairStack{0, 0};
            if (rayCls == 0 || end == 0) pairStart<COUNT>(s, slab(), limit, node, pending, ws);  // the root's box
            else node = kPairFresh;  // a literal-class ray: traced whole below
        }
    };
    auto startShadow = [&](v3 x, v3 y) {  // DevScene::testOcclusion's ray set-up (scene.h:304-311)
        v3 dir = y - x;
        float dist = length(dir);
        dir = dir / dist;
        nAny++;
        startTrace(makeOffsetedRay(x, dir), dist - 1e-4f, true);
    };
    auto finishPixel = [&]() {  // pathtrace.cu:279-290
        v3 direct = mk3(sAccD[0][t], sAccD[1][t], sAccD[2][t]);
        v3 indirect = mk3(sAccI[0][t], sAccI[1][t], sAccI[2][t]);
        if (hasNanOrInf(direct)) direct = mk3(0.f);
        if (hasNanOrInf(indirect)) indirect = mk3(0.f);
        direct = HDRToLDR(direct);
        indirect = HDRToLDR(indirect);
        storeRunningMean(directIllum, outIdx, direct, iter);
        storeRunningMean(indirectIllum, outIdx, indirect, iter);
        atomicMax(&blockCost[myBlock], pathSteps);
        state = PS_IDLE;
    };
    auto startExtensionOrFinish = [&]() {
        if (sFlags[t] & 1) {
            Ray ray{mk3(sExtO[0][t], sExtO[1][t], sExtO[2][t]), mk3(sExtD[0][t], sExtD[1][t], sExtD[2][t])};
            nClosest++;
            startTrace(ray, 3.402823466e+38f, false);
        } else {
            finishPixel();
        }
    };

    for (;;) {
        // ---------------- raygen for idle lanes ----------------
        unsigned long long idleM = __ballot(state == PS_IDLE);
        int nIdle = __popcll(idleM);
        if (!exhausted && nIdle >= RD_PT_REFILL_MIN) {
            PH_COUNT(14, idleM);
            int myRank = __popcll(idleM & laneMaskLt());
            int taken = 0;
            while (taken < nIdle && !exhausted) {
                if (slotNext == 64) {
                    int b = 0;
                    if (lane == 0) b = atomicAdd(&pc->blockHead, 1);
                    curBlock = __shfl(b, 0, 64) + gridWavesN;
                    slotNext = 0;
                    if (curBlock >= paddedBlocks) {
                        exhausted = true;
                        break;
                    }
                }
                int avail = 64 - slotNext;
                int give = (nIdle - taken) < avail ? (nIdle - taken) : avail;
                if (state == PS_IDLE && myRank >= taken && myRank < taken + give) {
                    // block to render: the scheduled order if there is one, else XCD-aware plain order (workgroup-sized
                    // groups of 4 blocks, as in the one-shot kernels)
                    unsigned blk;
                    bool ok = true;
                    if (blockOrder) {
                        blk = (unsigned)blockOrder[curBlock];
                    } else {
                        unsigned wgLogical;
                        ok = xcdSwizzle((unsigned)curBlock >> 2, (unsigned)(numBlocks + 3) >> 2, wgLogical);
                        blk = ok ? wgLogical * 4u + ((unsigned)curBlock & 3u) : 0xffffffffu / 64u;
                    }
                    Pix px = mapPixel(pm, blk, (unsigned)(slotNext + (myRank - taken)));
                    if (px.valid && ok) {
                        outIdx = px.out;
                        myBlock = int(blk);
                        pathSteps = 0;
                        Sampler rng = makeSeededRandomEngine(looper, px.index, 0, s.sobol);
                        Ray ray = cameraSample(cam, px.x, px.y, sample4D(rng));
                        rngScramble = rng.scramble;
                        rngPtr = rng.ptr;
                        sAccD[0][t] = sAccD[1][t] = sAccD[2][t] = 0.f;
                        sAccI[0][t] = sAccI[1][t] = sAccI[2][t] = 0.f;
                        k = 0;
                        nClosest++;
                        startTrace(ray, 3.402823466e+38f, false);
#ifndef RD_NO_ROOT_SHORTCUT
                        if (PAIRS) {  // the root has been tested by startTrace
                            if (rayCls == 0 && end != 0 && node == kPairNone && pending < 0) {
                                sAccD[0][t] = sAccD[1][t] = sAccD[2][t] = 1.f;
                                finishPixel();
                            }
                        } else
                        // A primary ray that misses the root box is finished here and now: the walk would visit the root, miss
                        // it, follow its link to the end of the array, and the shading step would write direct = 1
                        // (pathtrace.cu:169-172).  More than half of the Cornell frame's pixels are such rays; taking them
                        // through the trace / shade phases costs the wave's other lanes two phase changes each.
                        if (rayCls == 0 && end != 0) {
                            const NodeRec *root = reinterpret_cast<const NodeRec *>(nodeBase + ordOfs);
                            const float4 lo = root->lo_prim, hi = root->hi_next;
                            float boundDist;
                            const bool boundHit = aabbFastPk(lo, hi, rp, boundDist);
                            if (!(boundHit && boundDist < tmax) && __float_as_int(hi.w) == end) {
                                if (COUNT) ws.nodes++;
                                sAccD[0][t] = sAccD[1][t] = sAccD[2][t] = 1.f;
                                finishPixel();
                            }
                        }
#endif
                    }
                }
                slotNext += give;
                taken += give;
            }
        }
        if (__ballot(state != PS_IDLE) == 0ull) {
            if (exhausted) break;
            continue;
        }
#ifdef RD_PERSIST_STAMPS
        if (exhausted && !stampedDry) {
            stampedDry = true;
            if (lane == 0 && gw < 4096) pc->stamp[1][gw] = wall_clock64();
        }
#endif

        PH_MARK(0);
        // ---------------- literal-class rays: traced whole by the whole wave (traverse.h, coopTraceWhole) ----------------
        {
            unsigned long long lit = __ballot(PAIRS ? (state == PS_TRACE && node == kPairFresh)
                                                    : (state == PS_TRACE && rayCls != 0 && node == 0 && pending < 0 && end != 0));
#ifdef RD_PERSIST_STAMPS
            const unsigned long long litT0 = wall_clock64();
            litRays += (unsigned long long)__popcll(lit);
#endif
            while (lit) {
                const int L = __ffsll((long long)lit) - 1;
                lit &= lit - 1ull;
                const bool shadowL = readlaneI(isShadow ? 1 : 0, L) != 0;
                const NodeRec *un = reinterpret_cast<const NodeRec *>(nodeBase + (unsigned)readlaneI((int)ordOfs, L) * (PAIRS ? ordStride : 1u));
                const RaySlab ur = readlaneRay(slab(), L);
                const float lim = readlaneF(tmax, L);
                CoopTrace ct = shadowL ? coopTraceWhole<true>(s, un, ur, lim) : coopTraceWhole<false>(s, un, ur, lim);
                if (lane == L) {
                    hitPrim = ct.hitPrim;
                    hitBary = ct.bary;
                    tmax = ct.tmax;
                    occluded = ct.found;
                    node = PAIRS ? kPairNone : end;
                    if (COUNT) {
                        ws.nodes += ct.nodes;
                        ws.tris += ct.tris;
                    }
                }
            }
#ifdef RD_PERSIST_STAMPS
            litTicks += wall_clock64() - litT0;
#endif
        }
        PH_MARK(1);
        // ---------------- box steps ----------------
        // Run until a quarter of the lanes that entered the loop have stopped walking (parked on a leaf or finished
        // their ray): one ballot + popcount per step is the whole scheduling cost.
        if (PAIRS) {
            bool busy = state == PS_TRACE && pending < 0 && !walkOver();
            const int nStart = __popcll(__ballot(busy));
            if (nStart > 0) {
                const int minWalk = (nStart * (RD_PT_LEAF_DEN - 1) + RD_PT_LEAF_DEN - 1) / RD_PT_LEAF_DEN;
                const RaySlab rs = slab();
                do {
                    PH_COUNT(8, __ballot(busy));
                    waveSteps++;
                    pairPopOne<COUNT>(sStack, lane, ovf, tmax, busy, node, stack, pending, ws);
                    if (busy && node >= 0) {
                        pairStep<COUNT>(s.pairs, sStack, lane, ovf, rs, tmax, (int)ordOfs, node, stack, pending, ws);
                        busy = pending < 0 && (node >= 0 || stack.sp > 0);
                    }
                } while (__popcll(ballotb(busy)) >= (minWalk > 1 ? minWalk : 1));
            }
        } else {
            bool walking = state == PS_TRACE && pending < 0 && node != end;
            int nStart = __popcll(__ballot(walking));
            if (nStart == 1) {
                // a lone walker (the drain of the frame): the whole wave tests 64 boxes ahead for it (coopWalk)
                const int L = __ffsll((long long)__ballot(walking)) - 1;
                CoopResult cr = coopWalk(reinterpret_cast<const NodeRec *>(nodeBase + (unsigned)readlaneI((int)ordOfs, L)),
                                         readlaneI(node, L), end, readlaneRay(slab(), L), readlaneF(tmax, L), RD_COOP_WINDOWS);
                waveSteps += 8u;
                if (lane == L) {
                    node = cr.node;
                    pending = cr.pending;
                    if (COUNT) ws.nodes += cr.visited;
                }
            } else if (nStart > 0) {
                const int minWalk = (nStart * (RD_PT_LEAF_DEN - 1) + RD_PT_LEAF_DEN - 1) / RD_PT_LEAF_DEN;
                do {
                    PH_COUNT(8, __ballot(walking));
                    waveSteps++;
                    if (walking) {
                        // one VALU for the address: base (SGPRs) + 32-bit byte offset
                        const NodeRec *rec = reinterpret_cast<const NodeRec *>(nodeBase + (ordOfs + ((unsigned)node << 5)));
                        float4 lo = rec->lo_prim;
                        float4 hi = rec->hi_next;
                        float boundDist;
                        if (COUNT) ws.nodes++;
                        bool boundHit = aabbFastPk(lo, hi, rp, boundDist);  // every walker here is of class 0 (the others were traced whole above)
                        if (boundHit && boundDist < tmax) {
                            pending = __float_as_int(lo.w);
                            node++;
                        } else {
                            node = __float_as_int(hi.w);
                        }
                        walking = pending < 0 && node != end;
                    }
                } while (__popcll(ballotb(walking)) >= (minWalk > 1 ? minWalk : 1));
                // (Requesting both successors ahead of the box test was measured and rejected: the L1 is busy ~80 % of
                //  the launch — TCP_GATE_EN — and doubling its requests cost 7 % in the bulk and gained nothing in the
                //  drain; DESIGN.md §7.  Round 3 tried the general form for SMALL launches — a rank's share of the frame on
                //  4-8 GPUs, where the launch lasts one path latency: 2 / 4 / 8 records of the threaded order per round trip,
                //  walked through without waiting again, bit-exact — and it lost there too (2.58 -> 3.38 / 3.53 / 6.41 ms per
                //  rank at 8 ranks, profiles/r03_c_*): a draining wave is bound by instruction issue shared with its SIMD's
                //  other waves, not by the round trip.)
            }
        }
        PH_MARK(2);
        // ---------------- leaf tests of parked lanes ----------------
        if (__ballot(state == PS_TRACE && pending >= 0) != 0ull) PH_COUNT(10, __ballot(state == PS_TRACE && pending >= 0));
        if (state == PS_TRACE && pending >= 0) {
            TriVerts tv = loadTri(s.tris, pending);
            float dist;
            v2 bary;
            if (COUNT) ws.tris++;
            bool hit = intersectTriangle(slab(), tv.a, tv.b, tv.c, bary, dist);
            if (hit && dist < tmax) {
                if (isShadow) {
                    occluded = true;
                    node = PAIRS ? kPairNone : end;
                    stack = PairStack{0, 0};
                } else {
                    hitPrim = pending;
                    tmax = dist;
                    hitBary = bary;
                }
            }
            pending = -1;
        }
        PH_MARK(3);
        // ---------------- retire finished traces ----------------
        if (state == PS_TRACE && pending < 0 && walkOver()) {
            pathSteps += waveSteps - stepMark;
            if (isShadow) {
                float nw = sNee[3][t];
                if (!occluded && nw >= 0.f) {  // the `+=` of pathtrace.cu:201-207
                    if (nw == 0.f) {
                        sAccD[0][t] += sNee[0][t]; sAccD[1][t] += sNee[1][t]; sAccD[2][t] += sNee[2][t];
                    } else {
                        sAccI[0][t] += sNee[0][t]; sAccI[1][t] += sNee[1][t]; sAccI[2][t] += sNee[2][t];
                    }
                }
                startExtensionOrFinish();
            } else {
                state = PS_SHADE;
            }
        }
        PH_MARK(4);
        // ---------------- shading ----------------
        unsigned long long shadeM = __ballot(state == PS_SHADE);
        if (shadeM != 0ull) {
            // Shade once a quarter of the busy lanes wait for it (16 of 64 in the bulk of the frame; in the drain, with a
            // handful of lanes left, a lane must not wait for every other lane's ray to end before it may continue).
            int nBusy = __popcll(__ballot(state != PS_IDLE));
            if (__popcll(shadeM) * 64 >= nBusy * (exhausted ? RD_SHADE_MIN_DRAIN : RD_SHADE_MIN)) {
                PH_COUNT(12, shadeM);
                if (state == PS_SHADE) {
                    v3 rayDir = rayD;
                    bool terminate = true;  // set false once a shadow or extension ray is started
                    do {
                        if (hitPrim == -1) {  // miss: pathtrace.cu:169-172 (primary), :232-247 (later)
                            if (k == 0) {
                                sAccD[0][t] = sAccD[1][t] = sAccD[2][t] = 1.f;
                            } else if (hasEnvMap(s)) {
                                v3 radiance = envLookup(s, rayDir) * mk3(sThr[0][t], sThr[1][t], sThr[2][t]);
                                float weight = (sFlags[t] & 2) ? 1.f : powerHeuristic(sExtPdf[t], environmentMapPdf(s, rayDir));
                                v3 add = radiance * weight;
                                sAccI[0][t] += add.x; sAccI[1][t] += add.y; sAccI[2][t] += add.z;
                            }
                            break;
                        }
                        nHits++;
                        Surface isec;
                        fetchSurface(s, hitPrim, hitBary, isec);
                        Material material = texturedMaterial(s, isec);
                        v3 throughput;
                        if (k == 0) {
                            material.baseColor = mk3(1.f);  // DENOISER_DEMODULATE (:175-178)
                            if (material.type == Light) {   // :179-182
                                sAccD[0][t] = sAccD[1][t] = sAccD[2][t] = 1.f;
                                break;
                            }
                            throughput = mk3(1.f);
                        } else {
                            throughput = mk3(sThr[0][t], sThr[1][t], sThr[2][t]);
                            if (material.type == Light) {  // :251-271
                                if (dot(isec.norm, rayDir) < 0.f) break;
                                v3 radiance = material.baseColor;
                                bool deltaSample = (sFlags[t] & 2) != 0;
                                v3 curPos = mk3(sCur[0][t], sCur[1][t], sCur[2][t]);
                                float weight = deltaSample
                                                   ? 1.f
                                                   : powerHeuristic(sExtPdf[t],
                                                                    pdfAreaToSolidAngle(luminance(radiance) * s.sumLightPowerInv *
                                                                                            getPrimitiveArea(s, isec.primId),
                                                                                        curPos, isec.pos, isec.norm));
                                v3 add = radiance * throughput * weight;
                                sAccI[0][t] += add.x; sAccI[1][t] += add.y; sAccI[2][t] += add.z;
                                break;
                            }
                        }
                        const int depth = k + 1;
                        if (depth > maxDepth) break;  // loop bound of pathtrace.cu:187
                        isec.wo = -rayDir;
                        bool deltaBSDF = (material.type == Dielectric);
                        if (material.type != Dielectric && dot(isec.norm, isec.wo) < 0.f) isec.norm = -isec.norm;
                        Sampler rng{s.sobol, rngScramble, rngPtr};
                        bool emitShadow = false;
                        v3 shadowTarget = mk3(0.f);
                        if (!deltaBSDF) {  // NEE (:195-208)
                            v4 r4 = sample4D(rng);
                            if (s.lightSamplerLength != 0) {
                                LightPick lp = pickLightPoint(s, isec.pos, r4);
                                v3 radiance = mk3(0.f), wi = mk3(0.f);
                                float lightPdf = lightPdfUnoccluded(s, isec.pos, lp, radiance, wi);
                                v3 cc = mk3(0.f);
                                float nw = -1.f;
                                if (lightPdf > 0.f) {
                                    float BSDFPdf = materialPdf(material, isec.norm, isec.wo, wi);
                                    cc = throughput * materialBSDF(material, isec.norm, isec.wo, wi) * radiance *
                                         satDot(isec.norm, wi) / lightPdf * powerHeuristic(lightPdf, BSDFPdf);
                                    nw = depth == 1 ? 0.f : 1.f;
                                }
                                sNee[0][t] = cc.x; sNee[1][t] = cc.y; sNee[2][t] = cc.z; sNee[3][t] = nw;
                                shadowTarget = lp.sampled;
                                emitShadow = true;  // traced even if nothing can be added
                            }
                        }
                        sCur[0][t] = isec.pos.x; sCur[1][t] = isec.pos.y; sCur[2][t] = isec.pos.z;
                        BSDFSample sample;
                        sample.pdf = 0.f;
                        materialSample(material, isec.norm, isec.wo, sample3D(rng), sample);
                        rngScramble = rng.scramble;
                        rngPtr = rng.ptr;
                        int flags = 0;
                        if (!(sample.type == Invalid) && !(sample.pdf < 1e-8f)) {
                            bool deltaSample = (sample.type & Specular) != 0;
                            throughput = throughput * (sample.bsdf / sample.pdf * (deltaSample ? 1.f : absDot(isec.norm, sample.dir)));
                            Ray ray = makeOffsetedRay(isec.pos, sample.dir);
                            sExtO[0][t] = ray.o.x; sExtO[1][t] = ray.o.y; sExtO[2][t] = ray.o.z;
                            sExtD[0][t] = ray.d.x; sExtD[1][t] = ray.d.y; sExtD[2][t] = ray.d.z;
                            sExtPdf[t] = sample.pdf;
                            sThr[0][t] = throughput.x; sThr[1][t] = throughput.y; sThr[2][t] = throughput.z;
                            flags = 1 | (deltaSample ? 2 : 0);
                        }
                        sFlags[t] = flags;
                        k = depth;
                        terminate = false;
                        if (emitShadow) startShadow(isec.pos, shadowTarget);
                        else startExtensionOrFinish();
                    } while (false);
                    if (terminate) finishPixel();
                }
            }
        }
        PH_MARK(5);
    }
#ifdef RD_PERSIST_PHASES
#ifdef RD_PERSIST_PHASES_DRAIN
    phT[6] = phT[0] + phT[1] + phT[2] + phT[3] + phT[4] + phT[5];
    (void)phStart;
#else
    phT[6] = __builtin_amdgcn_s_memtime() - phStart;
#endif
    if (lane == 0)
        for (int i = 0; i < 16; i++) atomicAdd(&pc->phase[i], phT[i]);
#endif
#ifdef RD_PERSIST_STAMPS
    if (lane == 0 && gw < 4096) {
        pc->stamp[2][gw] = wall_clock64();
        pc->stamp[3][gw] = litTicks;
        pc->stamp[4][gw] = litRays;
    }
#endif
    if (COUNT) flushCounters(s.counters, nClosest, nAny, nHits, ws);
}

// ---- renderGBuffer as a persistent launch (gBuffer.cu:3-76) -------------------------------------------------------------
// k_gbuffer (kernels_pt.h) binds a pixel to a lane for the whole launch, so every wave waits for the longest of its 64
// primary rays (measured on the teapots scene: 2.97 ms for 2.07 M rays, a quarter of k_pt_persistent's rate).  Here lanes
// are refilled: a lane whose ray has ended writes its G-buffer record (batched: once 16 lanes wait) and takes the next
// pixel.  Same centre ray, same walk, same record per pixel as k_gbuffer.
// One pixel's G-buffer record (gBuffer.cu:28-67), from the primary ray's closest hit.
RD_DEV void gbufStore(const DScene &s, const DCamera &cam, const DCamera &lastCam, const GBufPtrs &gb, int idx, const RaySlab &rs,
                      int hitPrim, v2 hitBary) {
    if (hitPrim != -1) {
        Surface isec;
        fetchSurface(s, hitPrim, hitBary, isec);
        int matId = isec.matId;
        if (loadMaterial(s.mats, isec.matId).type == Light) matId = -2;  // NullPrimitive - 1 (gBuffer.cu:33-37)
        Material material = texturedMaterial(s, isec);                  // :44 (may perturb isec.norm)
        store3(gb.albedo, idx, material.baseColor);
        store3(gb.normal, idx, isec.norm);
        gb.primId[idx] = matId;
        gb.depth[idx] = length(rs.o - isec.pos);
        v2 ndc = cameraRasterUV(lastCam, isec.pos);
        int lx = (int)(float(lastCam.resx) * ndc.x), ly = (int)(float(lastCam.resy) * ndc.y);
        gb.motion[idx] = (lx >= 0 && lx < gb.width && ly >= 0 && ly < gb.height) ? ly * cam.resx + lx : -1;
    } else {
        store3(gb.albedo, idx, hasEnvMap(s) ? envLookup(s, rs.d) : mk3(0.f));  // :61-66
        store3(gb.normal, idx, mk3(0.f));
        gb.primId[idx] = -1;
        gb.depth[idx] = 1.f;
        gb.motion[idx] = 0;
    }
}

// The un-jittered centre ray of pixel (x, y) (gBuffer.cu:11-26).
RD_DEV Ray gbufPrimaryRay(const DCamera &cam, int x, int y) {
    float aspect = float(cam.resx) / float(cam.resy);
    v2 pixelsize = {1.f / float(cam.resx), 1.f / float(cam.resy)};
    v2 scr = mk2(float(x), float(y)) * pixelsize;
    v2 ruv = scr + pixelsize * mk2(0.5f, 0.5f);
    ruv = {1.f - ruv.x * 2.f, 1.f - ruv.y * 2.f};
    v3 pLens = mk3(0.f);
    v2 f = (ruv * mk2(aspect, 1.f)) * cam.tanFovY;
    v3 pFocus = mk3(f.x, f.y, 1.f) * cam.focalDist;
    v3 dir = pFocus - pLens;
    Ray ray;
    ray.o = cam.position + cam.right * pLens.x + cam.up * pLens.y;
    ray.d = normalize(mul(m3{cam.right, cam.up, cam.view}, dir));
    return ray;
}

#ifndef RD_GB_FINISH_MIN
#define RD_GB_FINISH_MIN 16
#endif
// PAIRS: the primary rays walk the sibling pairs (traverse.h, pairStep).  Coherent rays gain nothing from it while the threaded arrays
// sit in the caches (teapots, 201 k nodes: 0.78 ms either way) — the host picks it for trees whose six threaded arrays outgrow the
// Infinity Cache (radish_hip.hip, rdh_gbuffer_render).
template <bool COUNT, bool DEFER, bool PAIRS = false>
__global__ __launch_bounds__(64, PAIRS && !COUNT ? 4 : 1) void k_gbuffer_persistent(DScene s, DCamera cam, DCamera lastCam, PixelMap pm, GBufPtrs gb,
                                                           PersistCounters *pc, int2 *__restrict__ pairOvf = nullptr, int pairOvfDepth = 0) {
    __shared__ int2 sStack[PAIRS ? kPairLds * 64 : 1];  // PAIRS: the lanes' stacks (traverse.h, pairPush)
    int2 *const ovf = PAIRS ? pairOvf + (size_t)blockIdx.x * (size_t)pairOvfDepth * 64 : nullptr;
    const int lane = int(threadIdx.x) & 63;
    const int end = s.bvhSize;
    const int numBlocks = pm.numBlocks;
    const int paddedBlocks = int((((unsigned)(numBlocks + 3) / 4u + 7u) / 8u) * 8u * 4u);
    WalkStats ws{0, 0};
    unsigned nClosest = 0, nHits = 0;
    int curBlock = int(blockIdx.x);
    int slotNext = 0;
    const int gridWavesN = int(gridDim.x);
    bool exhausted = curBlock >= paddedBlocks;

    constexpr int G_IDLE = 0, G_TRACE = 1, G_DONE = 2;
    int state = G_IDLE;
    int pixIdx = 0;
    RaySlab rs;
    rs.o = rs.d = rs.inv = mk3(0.f);
    rs.cls = 0;
    const NodeRec *nodes = s.nodes[0];
    int node = end, pending = -1;  // PAIRS: `node` is the pair to enter next (pairStep's `cur`)
    int ord = 0;
    PairStack stack{0, 0};
    float tmax = 0.f;
    int hitPrim = -1;
    v2 hitBary = mk2(0.f, 0.f);
    const bool deferAll = DEFER && pc->deferCount <= kDeferCap;  // written by k_gbuffer_find_literal, earlier in the stream
    auto walkOver = [&]() { return PAIRS ? (node == kPairNone && stack.sp == 0) : (node == end); };

    for (;;) {
        // ---------------- new pixels for idle lanes ----------------
        const unsigned long long idleM = __ballot(state == G_IDLE);
        const int nIdle = __popcll(idleM);
        if (!exhausted && nIdle >= RD_PIX_REFILL_MIN) {
            const int myRank = __popcll(idleM & laneMaskLt());
            int taken = 0;
            while (taken < nIdle && !exhausted) {
                if (slotNext == 64) {
                    int b = 0;
                    if (lane == 0) b = atomicAdd(&pc->blockHead, 1);
                    curBlock = __shfl(b, 0, 64) + gridWavesN;
                    slotNext = 0;
                    if (curBlock >= paddedBlocks) {
                        exhausted = true;
                        break;
                    }
                }
                const int avail = 64 - slotNext;
                const int give = (nIdle - taken) < avail ? (nIdle - taken) : avail;
                if (state == G_IDLE && myRank >= taken && myRank < taken + give) {
                    unsigned wgLogical;
                    const bool ok = xcdSwizzle((unsigned)curBlock >> 2, (unsigned)(numBlocks + 3) >> 2, wgLogical);
                    const unsigned blk = ok ? wgLogical * 4u + ((unsigned)curBlock & 3u) : 0xffffffffu / 64u;
                    Pix px = mapPixel(pm, blk, (unsigned)(slotNext + (myRank - taken)));
                    if (px.valid && ok) {
                        pixIdx = px.index;
                        Ray ray = gbufPrimaryRay(cam, px.x, px.y);
                        rs = makeRaySlab(ray);
                        nodes = s.nodes[getMTBVHId(-ray.d)];
                        node = 0;
                        pending = -1;
                        hitPrim = -1;
                        tmax = 3.402823466e+38f;
                        if (DEFER && raySetAside(rs.cls) && end != 0 && deferAll) {
                            // a literal-class ray: k_gbuffer_literal has it (found by k_gbuffer_find_literal)
                        } else {
                            nClosest++;
                            state = G_TRACE;
                            if (PAIRS) {
                                ord = getMTBVHId(-ray.d);
                                stack = PairStack{0, 0};
                                if (rs.cls == 0 || end == 0) pairStart<COUNT>(s, rs, tmax, node, pending, ws);  // the root's box
                                else node = kPairFresh;  // a literal-class ray: traced whole below
                            }
                        }
                    }
                }
                slotNext += give;
                taken += give;
            }
        }
        if (__ballot(state != G_IDLE) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---------------- literal-class rays: traced whole by the whole wave ----------------
        {
            unsigned long long lit = __ballot(PAIRS ? (state == G_TRACE && node == kPairFresh)
                                                    : (state == G_TRACE && rs.cls != 0 && node == 0 && pending < 0 && end != 0));
            while (lit) {
                const int L = __ffsll((long long)lit) - 1;
                lit &= lit - 1ull;
                CoopTrace ct = coopTraceWhole<false>(s, readlanePtr(nodes, L), readlaneRay(rs, L), readlaneF(tmax, L));
                if (lane == L) {
                    hitPrim = ct.hitPrim;
                    hitBary = ct.bary;
                    tmax = ct.tmax;
                    node = PAIRS ? kPairNone : end;
                    if (COUNT) {
                        ws.nodes += ct.nodes;
                        ws.tris += ct.tris;
                    }
                }
            }
        }
        // ---------------- box steps ----------------
        if (PAIRS) {
            bool busy = state == G_TRACE && pending < 0 && !walkOver();
            const int nStart = __popcll(__ballot(busy));
            if (nStart > 0) {
                const int minWalk = (nStart * (RD_LEAF_DEN - RD_LEAF_NUM) + RD_LEAF_DEN - 1) / RD_LEAF_DEN;
                do {
                    pairPopOne<COUNT>(sStack, lane, ovf, tmax, busy, node, stack, pending, ws);
                    if (busy && node >= 0) {
                        pairStep<COUNT>(s.pairs, sStack, lane, ovf, rs, tmax, ord, node, stack, pending, ws);
                        busy = pending < 0 && (node >= 0 || stack.sp > 0);
                    }
                } while (__popcll(__ballot(busy)) >= (minWalk > 1 ? minWalk : 1));
            }
        } else {
            bool walking = state == G_TRACE && pending < 0 && node != end;
            const int nStart = __popcll(__ballot(walking));
            if (nStart == 1) {
                const int L = __ffsll((long long)__ballot(walking)) - 1;
                CoopResult cr = coopWalk(readlanePtr(nodes, L), readlaneI(node, L), end, readlaneRay(rs, L), readlaneF(tmax, L),
                                         RD_COOP_WINDOWS);
                if (lane == L) {
                    node = cr.node;
                    pending = cr.pending;
                    if (COUNT) ws.nodes += cr.visited;
                }
            } else if (nStart > 0) {
                const int minWalk = (nStart * (RD_LEAF_DEN - RD_LEAF_NUM) + RD_LEAF_DEN - 1) / RD_LEAF_DEN;
                do {
                    if (walking) {
                        float4 lo = nodes[node].lo_prim;
                        float4 hi = nodes[node].hi_next;
                        float boundDist;
                        if (COUNT) ws.nodes++;
                        bool boundHit = aabbFast(lo, hi, rs, boundDist);
                        if (boundHit && boundDist < tmax) {
                            pending = __float_as_int(lo.w);
                            node++;
                        } else {
                            node = __float_as_int(hi.w);
                        }
                        walking = pending < 0 && node != end;
                    }
                } while (__popcll(__ballot(walking)) >= (minWalk > 1 ? minWalk : 1));
            }
        }
        // ---------------- leaf tests ----------------
        if (state == G_TRACE && pending >= 0) {
            TriVerts tv = loadTri(s.tris, pending);
            float dist;
            v2 bary;
            if (COUNT) ws.tris++;
            bool hit = intersectTriangle(rs, tv.a, tv.b, tv.c, bary, dist);
            if (hit && dist < tmax) {
                hitPrim = pending;
                tmax = dist;
                hitBary = bary;
            }
            pending = -1;
        }
        if (state == G_TRACE && pending < 0 && walkOver()) state = G_DONE;
        // ---------------- write the records of finished pixels ----------------
        {
            const unsigned long long doneM = __ballot(state == G_DONE);
            const int nBusy = __popcll(__ballot(state != G_IDLE));
            if (doneM != 0ull && __popcll(doneM) * 64 >= nBusy * RD_GB_FINISH_MIN) {
                if (state == G_DONE) {
                    if (hitPrim != -1) nHits++;
                    gbufStore(s, cam, lastCam, gb, pixIdx, rs, hitPrim, hitBary);
                    state = G_IDLE;
                }
            }
        }
    }
    if (COUNT) flushCounters(s.counters, nClosest, 0u, nHits, ws);
}

// ---- renderGBuffer with the primary rays of an 8x8 block walked as a PACKET (traverse.h, packetWalk) ---------------------------------
// Centre rays of neighbouring pixels make nearly the same walk (teapots camera: 89 visits per ray, 107 different nodes per block), so
// a wave takes one block, walks the union once with one uniform node load per visit, and writes its 64 records.  No lane refill,
// no persistent grid: a wave's work is one short walk.  Same centre ray, same visits, same record per pixel as k_gbuffer.
template <bool COUNT, bool DEFER>
__global__ __launch_bounds__(256) void k_gbuffer_packet(DScene s, DCamera cam, DCamera lastCam, PixelMap pm, GBufPtrs gb, const PersistCounters *pc,
                                                        int budget) {
    const int lane = int(threadIdx.x) & 63;
    const unsigned blk = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (blk >= (unsigned)pm.numBlocks) return;
    const int end = s.bvhSize;
    const Pix px = mapPixel(pm, blk, (unsigned)lane);
    WalkStats ws{0, 0};
    unsigned nClosest = 0, nHits = 0;
    const bool deferAll = DEFER && pc->deferCount <= kDeferCap;  // written by k_gbuffer_find_literal, earlier in the stream
    Ray ray{mk3(0.f), mk3(0.f, 0.f, 1.f)};
    if (px.valid) ray = gbufPrimaryRay(cam, px.x, px.y);
    const RaySlab rs = makeRaySlab(ray);
    const int ord = getMTBVHId(-ray.d);
    float tmax = 3.402823466e+38f;
    int hitPrim = -1;
    v2 hitBary = mk2(0.f, 0.f);
    const bool traced = px.valid && !(DEFER && raySetAside(rs.cls) && end != 0 && deferAll);  // else: k_gbuffer_literal has this pixel
    if (traced) nClosest++;
    unsigned long long lit = __ballot(traced && rs.cls != 0 && end != 0);
    while (lit) {  // literal-class rays kept here: traced whole by the wave, one after the other
        const int L = __ffsll((long long)lit) - 1;
        lit &= lit - 1ull;
        const NodeRec *nodes = s.nodes[0] + (size_t)readlaneI(ord, L) * (size_t)(end + 1);
        CoopTrace ct = coopTraceWhole<false>(s, nodes, readlaneRay(rs, L), readlaneF(tmax, L));
        if (lane == L) {
            hitPrim = ct.hitPrim;
            hitBary = ct.bary;
            tmax = ct.tmax;
            if (COUNT) {
                ws.nodes += ct.nodes;
                ws.tris += ct.tris;
            }
        }
    }
    packetWalkAll<COUNT>(s, traced && rs.cls == 0 && end != 0, ord, rs, tmax, hitPrim, hitBary, ws, budget);
    if (traced) {
        if (hitPrim != -1) nHits++;
        gbufStore(s, cam, lastCam, gb, px.index, rs, hitPrim, hitBary);
    }
    if (COUNT) flushCounters(s.counters, nClosest, 0u, nHits, ws);
}

// Literal-class primary rays are traced apart, each by a whole 1 024-thread workgroup (wg_trace.h), in a launch that runs
// BESIDE k_gbuffer_persistent on a second stream: traced in place by one wave, one such ray takes 1.2 ms on the teapots scene,
// twice the rest of the pass.  The ray's class depends on the camera and the pixel only, so a first tiny kernel lists them;
// when there are more than kDeferCap (an axis-aligned camera makes a whole pixel row such rays) none is set aside — many
// one-wave traces in parallel are the better use of the chip then.
__global__ __launch_bounds__(256) void k_gbuffer_find_literal(DScene s, DCamera cam, PixelMap pm, PersistCounters *pc) {
    const int idx = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= cam.resx * cam.resy || s.bvhSize == 0) return;
    // on a tile partition only the rays of this rank's tiles (tile t belongs to rank t % world)
    if (pm.world > 1 && (((idx / cam.resx) / pm.tile) * pm.tilesX + (idx % cam.resx) / pm.tile) % pm.world != pm.rank) return;
    Ray ray = gbufPrimaryRay(cam, idx % cam.resx, idx / cam.resx);
    if (raySetAside(makeRaySlab(ray).cls)) {
        const int at = atomicAdd(&pc->deferCount, 1);
        if (at < kDeferCap) pc->deferred[at] = idx;
    }
}

template <bool COUNT>
__global__ __launch_bounds__(kWgTraceThreads) void k_gbuffer_literal(DScene s, DCamera cam, DCamera lastCam, GBufPtrs gb,
                                                                      const PersistCounters *pc) {
    __shared__ WgTraceShared sh;
    const int n = pc->deferCount;
    if (n > kDeferCap) return;  // k_gbuffer_persistent traces them all in place
    for (int i = int(blockIdx.x); i < n; i += int(gridDim.x)) {
        const int idx = pc->deferred[i];
        Ray ray = gbufPrimaryRay(cam, idx % cam.resx, idx / cam.resx);
        RaySlab rs = makeRaySlab(ray);
        CoopTrace ct = wgTraceWhole<false>(s, s.nodes[getMTBVHId(-ray.d)], rs, 3.402823466e+38f, sh);
        if (threadIdx.x == 0) {
            gbufStore(s, cam, lastCam, gb, idx, rs, ct.hitPrim, ct.bary);
            if (COUNT) {
                atomicAdd(&s.counters->closestRays, 1ull);
                atomicAdd(&s.counters->nodeVisits, (unsigned long long)ct.nodes);
                atomicAdd(&s.counters->triTests, (unsigned long long)ct.tris);
                if (ct.hitPrim != -1) atomicAdd(&s.counters->closestHits, 1ull);
            }
        }
    }
}

}  // namespace rd
