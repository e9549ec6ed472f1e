// radish_pt_amd/csrc/device/kernels_wave.h — the wavefront path tracer: raygen → [trace → shade]* → finish.
//
// The reference has no counterpart (it is a megakernel); the contract is that every pixel receives
// exactly the value singleKernelPT (/root/reference/src/pathtrace.cu:149-291) would give it.  That holds because
//   * the RNG stream depends only on (looper, pixel index, draw count) and its 8-byte state rides with the path;
//   * each path owns its accumulators (no cross-path atomics), and launches are stream-ordered so that the
//     additions to `direct`/`indirect` happen in the megakernel's order (NEE of bounce d, then the emitter hit
//     of bounce d);
//   * all arithmetic is the same device functions the megakernel calls.
//
// Launch structure per frame (maxDepth = D):  memset(counters) · raygen · for k = 0..D { trace(k) · shade(k) } · finish
// (optionally as three sub-frames on three streams, see k_wf_raygen; the material sort happens inside shade, see k_wf_shade)
//   trace(k)  = any-hit for the shadow rays emitted by shade(k-1)  +  closest-hit for the rays of bounce k
//   shade(k)  = surface fetch for hit k, emitter/miss termination, then bounce body k+1 (NEE sample → shadow
//               queue, BSDF sample → next ray queue) with wave64 ballot compaction
// trace and shade are persistent kernels: a fixed grid of waves pulls 64-item packets from device-side queue
// heads (one returning atomic per packet), so no queue length ever travels to the host.
#pragma once
#include "kernels_pt.h"
#include "wg_trace.h"

namespace rd {

constexpr int kWfLitCap = 8192;   // per stage; rays beyond it stay in the ordinary queues
constexpr int kWfLitShadow = 0x40000000;
constexpr int kMaxWaveDepth = 32;  // 4 + 7*depth Sobol dimensions <= 200 → depth <= 28

// Every counter sits on its own 128-byte line: same-address returning atomics serialise at ~12 ns each, and counters that share
// a line share that queue — the four class counters of k_wf_classify, side by side in one line, made it the slowest kernel of
// the sorted pipeline (0.87 ms per launch on the teapots frame).
struct alignas(128) WaveCounter {
    int v;
    int pad[31];
};
struct WaveCounters {
    WaveCounter rayCount[kMaxWaveDepth + 2];     // rays of bounce k (written by raygen / shade(k-1))
    WaveCounter shadowCount[kMaxWaveDepth + 2];  // shadow rays emitted by shade(k)
    WaveCounter litCount[kMaxWaveDepth + 2];     // literal-class rays (both kinds) that shade(k-1) set aside for trace(k)
    WaveCounter traceHead[kMaxWaveDepth + 2];
    WaveCounter shadeHead[kMaxWaveDepth + 2][4];
};

struct WaveWorkspace {
    float4 *ro;       // ray origin.xyz (offset applied), w = pdf of the BSDF sample that made this ray
    float4 *rd;       // ray direction.xyz, w = 1 if that sample was specular (deltaSample)
    float4 *thr;      // throughput.xyz
    float4 *prevPos;  // position the ray left from (curPos, pathtrace.cu:227) = origin of the pending shadow ray
    float4 *accD;     // direct.xyz
    float4 *accI;     // indirect.xyz
    float4 *nee;      // pending NEE contribution.xyz, w: 0 → direct, 1 → indirect, -1 → nothing to add
    float4 *sht;      // shadow-ray target.xyz
    uint2 *rng;       // {scramble, ptr}
    int4 *hit;        // {primId, bary.x, bary.y, shading class of the hit's material}
    int *rayq[2];
    int *shadowq;
    int *litq[2];     // path slots of rays that look literal-class (bit 30: a shadow ray); trace(k) starts them first, one per wave
    int litCap;       // entries a stage may hold: min(kWfLitCap, 64 * waves of the trace grid)
    int *treeOvf;     // k_wf_trace<.., true>: its lanes' stacks beyond their LDS rings, [wave of the grid][treeOvfDepth][64] int2 (traverse.h, pairPush)
    int treeOvfDepth;
    WaveCounters *ctr;
};


// Append `item` for every lane with pred==true: one atomic per wave, order inside the wave preserved.
RD_DEV void waveAppend(bool pred, int item, int *queue, int *count) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(count, __popcll(mask));
    base = __shfl(base, leader, 64);
    if (pred) queue[base + __popcll(mask & laneMaskLt())] = item;
}

// The same for the 64-record chunks of one k_wf_shade packet at once: ONE returning atomic (they serialise chip-wide per
// counter at ~12 ns; k_wf_shade's two appends per 64 records were most of its time).
constexpr int kShadeChunks = 4;  // k_wf_shade: 64-record chunks per packet
constexpr int kShadePacket = 64 * kShadeChunks;
RD_DEV void waveAppendN(const bool (&pred)[kShadeChunks], const int (&item)[kShadeChunks], int *queue, int *count) {
    unsigned long long m[kShadeChunks];
    int total = 0;
#pragma unroll
    for (int j = 0; j < kShadeChunks; j++) {
        m[j] = __ballot(pred[j]);
        total += __popcll(m[j]);
    }
    if (total == 0) return;
    int base = 0;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(count, total);
    base = __shfl(base, 0, 64);
#pragma unroll
    for (int j = 0; j < kShadeChunks; j++) {
        if (pred[j]) queue[base + __popcll(m[j] & laneMaskLt())] = item[j];
        base += __popcll(m[j]);
    }
}

// Work distribution for the persistent kernels.  Same-address returning atomics serialise chip-wide at ~12 ns each
// (MI355X_MICROARCH.md "dequeue"/"fanin"), so (a) every wave's FIRST packet is static — wave g takes items
// [g*kPacket, (g+1)*kPacket) — and the shared head only hands out what lies beyond gridWaves*kPacket, and (b) a pull
// reserves kPacket = 128 items.  Heads are zeroed once per frame by the counters memset.
constexpr int kPacket = 128;
RD_DEV int globalWave() { return int(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); }
RD_DEV int gridWaves() { return int(gridDim.x * (blockDim.x >> 6)); }
RD_DEV int wavePull(int *head, int packet = kPacket) {
    int base = 0;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(head, packet);
    return __shfl(base, 0, 64) + gridWaves() * packet;
}
// (Round 3 measured GUIDED self-scheduling — after the static packet a pull reserves an even share of half the items still
// unreserved, 32..128 for k_wf_trace, 64..packet for k_wf_shade, since a stage hands every wave only two or three fixed packets —
// and rejected it: teapots 9.07 -> 9.08 ms sorted / 8.90 -> 8.95 unsorted with three sub-frames, one pipeline 11.2 -> 11.8; the
// extra returning atomics cost what the finer tail saves, and with three pipelines a stage's tail is filled by the others anyway.
// profiles/r03_f_experiment_guided_scheduling.txt)

// ---- raygen ---------------------------------------------------------------------------------------------------
// Sub-frames: the wavefront pipeline can run as H independent pipelines over interleaved block sets (block b belongs to
// sub-frame b % H), each with its own workspace and stream, so that one sub-frame's stage tails are filled by the other's
// stage bodies.  `part` / `parts` = h / H; path slots are local to the sub-frame.
RD_DEV unsigned subFrameBlocks(const PixelMap &pm, int part, int parts) { return (unsigned)(pm.numBlocks - part + parts - 1) / (unsigned)parts; }
__global__ __launch_bounds__(256) void k_wf_raygen(DScene s, DCamera cam, PixelMap pm, WaveWorkspace w, int looper, int part, int parts) {
    unsigned wg;
    const unsigned nLocal = subFrameBlocks(pm, part, parts);
    bool wgValid = xcdSwizzle(blockIdx.x, (nLocal + 3u) >> 2, wg);
    unsigned lane = threadIdx.x & 63u;
    unsigned blockLocal = wgValid ? wg * 4u + (threadIdx.x >> 6) : 0xffffffffu / 64u;
    const bool inRange = wgValid && blockLocal < nLocal;
    Pix px = mapPixel(pm, inRange ? blockLocal * (unsigned)parts + (unsigned)part : 0xffffffffu / 64u, lane);
    bool valid = px.valid && inRange;
    int p = int(blockLocal * 64u + lane);  // path slot = work index (fixed for the frame)
    bool listed = false;
    if (valid) {
        Sampler rng = makeSeededRandomEngine(looper, px.index, 0, s.sobol);
        Ray ray = cameraSample(cam, px.x, px.y, sample4D(rng));
        w.ro[p] = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.f);
        w.rd[p] = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.f);
        w.accD[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        w.accI[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        w.rng[p] = make_uint2(rng.scramble, (unsigned)rng.ptr);
        // Primary rays that LOOK literal-class (the hint k_wf_shade uses for the rays it emits) go on stage 0's first list, so that a
        // workgroup of trace(0) traces them before anything else instead of one wave meeting them late (stage 0 was where the last
        // launches above 1 ms were: profiles/r03_w_teapots_trace_stage_time_histogram.txt).  A handful per frame: one atomic each.
        if (fabs_(ray.d.x) < 1.1e-6f || fabs_(ray.d.y) < 1.1e-6f || fabs_(ray.d.z) < 1.1e-6f) {
            const int at = atomicAdd(&w.ctr->litCount[0].v, 1);
            if (at < w.litCap) {
                w.litq[0][at] = p;
                listed = true;
            }
        }
    }
    // Bounce 0's queue is the identity over this launch's slots (-1 marks pixels outside the frame; -2 - slot: on the first list, as
    // k_wf_shade marks such rays): no compaction.
    if (inRange) w.rayq[0][p] = valid ? (listed ? -2 - p : p) : -1;
    if (blockIdx.x == 0 && threadIdx.x == 0) w.ctr->rayCount[0].v = int(nLocal) * 64;
}

// ---- trace(k): shadow rays of bounce k (from shade(k-1)) + closest hits of bounce k ---------------------------
// Persistent waves with LANE REFILL: a wave reserves 64 consecutive work items per atomic and hands them to lanes
// one by one as their previous ray finishes, so a wave never idles behind its longest ray (rays of one packet differ
// several-fold in visit count after the first bounce).  Shadow (any-hit) and extension (closest-hit) rays run through
// one walker; a lane's kind only decides how it terminates.
#ifndef RD_REFILL_MIN
#define RD_REFILL_MIN 16  // refill once at least this many lanes are idle
#endif
#ifndef RD_WF_FINISH_MIN  // retire finished rays once they are this many 64ths of the wave's busy lanes (0: every iteration).  With
#define RD_WF_FINISH_MIN 16  // three sub-frames, teapots / Cornell: 0 -> 9.31 / 4.02 ms, 8 -> 9.20 / 3.99, 16 -> 9.05 / 3.93, 24 -> 9.08 / 3.98
#endif
#ifndef RD_WF_COOP_LONE  // a lone walking lane is walked by the whole wave (coopWalk): 0.5-1 % on either scene
#define RD_WF_COOP_LONE 1
#endif

// PAIRS: the per-lane walks go over the sibling pairs (DScene::pairs: ONE 64-byte record per inner node for all six orderings, both
// children fetched and tested per round trip, the far one re-checked when the walk reaches it; traverse.h, pairStep) instead of the
// six threaded arrays — same triangle tests in the same order, same counters.  Chosen by the host for big scenes (radish_hip.hip,
// usePairs): a sixth of the node footprint and half the dependent round trips (profiles/r03_h_*).
#ifndef RD_WF_PAIR_WAVES
#define RD_WF_PAIR_WAVES 1  // waves per SIMD the pair variant is held to (1: whatever its registers allow)
#endif
template <bool COUNT, bool PAIRS = false>
__global__ __launch_bounds__(256, PAIRS && !COUNT ? RD_WF_PAIR_WAVES : 1) void k_wf_trace(DScene s, WaveWorkspace w, int k) {
    WaveCounters *c = w.ctr;
    const int nShadow = (k > 0) ? c->shadowCount[k - 1].v : 0;
    const int nRay = c->rayCount[k].v;
    const int total = nShadow + nRay;
    const int *rayq = w.rayq[k & 1];
    WalkStats ws{0, 0};
    unsigned nClosest = 0, nAny = 0, nHits = 0;
    const int end = s.bvhSize;
    const int END = end;  // `node` of a finished walk (!PAIRS)

    // wave-uniform reservation of work items; the first one is static (see wavePull)
    int resNext = globalWave() * kPacket, resEnd = resNext + kPacket;
    bool exhausted = resNext >= total;
    if (resEnd > total) resEnd = total;

    // per-lane walker state
    bool alive = false;
    bool isShadow = false;
    int p = -1;
    RaySlab rs;
    rs.o = rs.d = rs.inv = mk3(0.f);
    rs.cls = 0;
    const NodeRec *nodes = s.nodes[0];  // !PAIRS: the ray's threaded array
    int ord = 0, cur = kPairNone;  // PAIRS: the ray's ordering, the pair it enters next, its stack (pairStep)
    PairStack sp{0, 0};
    int node = END, pending = -1;
    float tmax = 0.f;
    int hitPrim = -1;
    v2 hitBary = mk2(0.f, 0.f);
    bool occluded = false;

    // Rays that shade(k-1) found to look literal-class (they wander through the tree for thousands of box steps; one wave needs 90 us
    // (median) to 1.5 ms for one of them on the teapots scene) start FIRST: picked up with a late packet one of them is what the stage
    // ends on (measured ceiling: 0.5-0.95 ms of the teapots frame).  Round 3: the workgroup traces them TOGETHER — its four waves are
    // at the same point only here, before each goes its own way with lane refill — 256 records per round trip with the parallel
    // resolve of wg_trace.h instead of 64 with a scalar one (coopTraceWhole): entry `it` of the list belongs to workgroup it % grid.
    // The list is a HINT (shade classifies by the direction it emits, this kernel by the ray it builds): an entry whose ray turns out
    // ordinary is left to a lane of the workgroup (first round below), as before.
    // (PAIRS: the four waves' stack rows take over the workgroup trace's LDS once the list is done)
    constexpr int kStackEntries = PAIRS ? 4 * kPairLds * 64 : 1;
    __shared__ union WfTraceLds {
        WgTraceSharedT<256> wg;
        int2 stack[kStackEntries];
    } lds;
    WgTraceSharedT<256> &wgsh = lds.wg;
    int2 *const stk = lds.stack + (PAIRS ? int(threadIdx.x >> 6) * kPairLds * 64 : 0);
    int2 *const ovf = PAIRS ? reinterpret_cast<int2 *>(w.treeOvf) + (size_t)globalWave() * (size_t)w.treeOvfDepth * 64 : nullptr;
    const int nLit = c->litCount[k].v < w.litCap ? c->litCount[k].v : w.litCap;
    for (int it = int(blockIdx.x); it < nLit; it += int(gridDim.x)) {  // uniform over the workgroup
        const int e = w.litq[k & 1][it];
        const bool sh = (e & kWfLitShadow) != 0;
        const int pp = e & (kWfLitShadow - 1);
        Ray ray;
        float lim;
        if (sh) {
            const float4 x4 = w.prevPos[pp], y4 = w.sht[pp];
            const v3 x = mk3(x4.x, x4.y, x4.z), y = mk3(y4.x, y4.y, y4.z);
            v3 dir = y - x;  // DevScene::testOcclusion's ray set-up (scene.h:304-311)
            const float dist = length(dir);
            dir = dir / dist;
            lim = dist - 1e-4f;
            ray = makeOffsetedRay(x, dir);
        } else {
            const float4 o = w.ro[pp], d = w.rd[pp];
            ray = Ray{mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z)};
            lim = 3.402823466e+38f;
        }
        const RaySlab urs = makeRaySlab(ray);
        if (urs.cls == 0 || end == 0) continue;  // the hint did not hold: a lane takes this entry below
        const NodeRec *un = s.nodes[getMTBVHId(-ray.d)];
        const CoopTrace ct = sh ? wgTraceWhole<true, 256>(s, un, urs, lim, wgsh) : wgTraceWhole<false, 256>(s, un, urs, lim, wgsh);
        if (threadIdx.x == 0) {  // the record of the finished ray (as in the retire step of the loop below)
            if (sh) {
                nAny++;
                const float4 n = w.nee[pp];
                if (!ct.found && n.w >= 0.f) {  // the addition sampleDirectLight's caller makes (pathtrace.cu:201-207)
                    float4 *acc = n.w == 0.f ? w.accD : w.accI;
                    const float4 a = acc[pp];
                    acc[pp] = make_float4(a.x + n.x, a.y + n.y, a.z + n.z, 0.f);
                }
            } else {
                nClosest++;
                const bool hit = ct.hitPrim != -1;
                if (hit) nHits++;
                w.hit[pp] = make_int4(ct.hitPrim, __float_as_int(hit ? ct.bary.x : 0.f), __float_as_int(hit ? ct.bary.y : 0.f),
                                      hit ? int(s.primClass[ct.hitPrim]) : 0);
            }
            if (COUNT) {
                ws.nodes += ct.nodes;
                ws.tris += ct.tris;
            }
        }
    }
    if (PAIRS) __syncthreads();  // every wave is done with the workgroup trace's LDS
    bool firstRound = true;
    bool fromList = false;  // this lane's item of the first round came from the list: skip it if the workgroup has traced it above
    for (;;) {
        // ---- hand new items to idle lanes ----
        bool doStart = false;  // this lane takes an item in this round: kind and path slot go straight into isShadow / p
        fromList = false;
        if (firstRound) {
            firstRound = false;
            // the workgroup's j-th entry (j = 4 * lane + wave of the workgroup: at most 256 per workgroup) — see above
            const int it = int(blockIdx.x) + (int(threadIdx.x & 63u) * 4 + int(threadIdx.x >> 6)) * int(gridDim.x);
            if (it < nLit) {
                const int e = w.litq[k & 1][it];
                doStart = true;
                fromList = true;
                isShadow = (e & kWfLitShadow) != 0;
                p = e & (kWfLitShadow - 1);
            }
        }
        unsigned long long idle = __ballot(!alive && !doStart && p < 0);
        int nIdle = __popcll(idle);
        if (!exhausted && nIdle >= RD_REFILL_MIN) {
            int myRank = __popcll(idle & laneMaskLt());
            int taken = 0;  // items handed out so far in this refill (wave-uniform)
            while (taken < nIdle && !exhausted) {
                if (resNext == resEnd) {
                    resNext = wavePull(&c->traceHead[k].v);
                    resEnd = resNext + kPacket;
                    if (resNext >= total) {
                        exhausted = true;
                        break;
                    }
                    if (resEnd > total) resEnd = total;
                }
                int avail = resEnd - resNext;
                int give = (nIdle - taken) < avail ? (nIdle - taken) : avail;
                if (!alive && !doStart && p < 0 && myRank >= taken && myRank < taken + give) {
                    const int item = resNext + (myRank - taken);
                    doStart = true;
                    isShadow = item < nShadow;
                    p = isShadow ? w.shadowq[item] : rayq[item - nShadow];
                }
                resNext += give;
                taken += give;
            }
        }
        if (doStart) {
            Ray ray;
            if (isShadow) {
                float4 x4 = w.prevPos[p], y4 = w.sht[p];
                v3 x = mk3(x4.x, x4.y, x4.z), y = mk3(y4.x, y4.y, y4.z);
                v3 dir = y - x;  // DevScene::testOcclusion's ray set-up (scene.h:304-311)
                float dist = length(dir);
                dir = dir / dist;
                tmax = dist - 1e-4f;
                ray = makeOffsetedRay(x, dir);
                nAny++;
            } else if (p >= 0) {
                float4 o = w.ro[p], d = w.rd[p];
                ray = Ray{mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z)};
                tmax = 3.402823466e+38f;
                nClosest++;
            } else {
                ray = Ray{mk3(0.f), mk3(0.f, 0.f, 1.f)};  // off-frame slot of bounce 0: nothing to trace
            }
            if (p >= 0) {
                rs = makeRaySlab(ray);
                pending = -1;
                if (PAIRS) {
                    ord = getMTBVHId(-ray.d);
                    sp = PairStack{0, 0};
                    if (rs.cls == 0 || end == 0) pairStart<COUNT>(s, rs, tmax, cur, pending, ws);  // the root's box
                    else cur = kPairFresh;  // a literal-class ray: traced whole below
                    alive = true;  // also when the root's box is missed: the lane goes through the loop once and is retired below
                } else {
                    nodes = s.nodes[getMTBVHId(-ray.d)];
                    node = 0;
                    alive = node != END;
                }
                hitPrim = -1;
                occluded = false;
                if (fromList && rs.cls != 0 && end != 0) {  // traced (and counted, and retired) by the workgroup above
                    if (isShadow) nAny--;
                    else nClosest--;
                    alive = false;
                    p = -1;
                }
            }
        }
        unsigned long long am = __ballot(alive);
        if (am == 0ull) {
            if (exhausted) break;
            continue;
        }

        // ---- literal-class rays: traced whole by the whole wave (traverse.h, coopTraceWhole) ----
        {
            unsigned long long lit = __ballot(PAIRS ? (alive && cur == kPairFresh) : (alive && rs.cls != 0 && node == 0 && pending < 0));
            while (lit) {
                const int L = __ffsll((long long)lit) - 1;
                lit &= lit - 1ull;
                const bool shadowL = readlaneI(isShadow ? 1 : 0, L) != 0;
                const NodeRec *un = PAIRS ? s.nodes[0] + (size_t)readlaneI(ord, L) * (size_t)(end + 1) : readlanePtr(nodes, L);
                const RaySlab ur = readlaneRay(rs, L);
                const float lim = readlaneF(tmax, L);
                CoopTrace ct = shadowL ? coopTraceWhole<true>(s, un, ur, lim) : coopTraceWhole<false>(s, un, ur, lim);
                if (int(threadIdx.x & 63u) == L) {
                    hitPrim = ct.hitPrim;
                    hitBary = ct.bary;
                    tmax = ct.tmax;
                    occluded = ct.found;
                    node = END;
                    cur = kPairNone;
                    alive = false;
                    if (COUNT) {
                        ws.nodes += ct.nodes;
                        ws.tris += ct.tris;
                    }
                }
            }
        }
        // ---- box steps until a quarter of the lanes that entered the loop have stopped walking (parked on a leaf or finished):
        // one ballot + popcount per step, as in k_walk_persistent; every walker here is of class 0 (the others were traced whole).
        // (Until the end of round 2 this loop took two ballots per step, tested "enough lanes alive" as well and went through the
        // class dispatch of boxTest: the teapots frame 10.0 -> 9.3 ms, the Cornell frame 4.6 -> 4.0 ms with three sub-frames.) ----
        if (PAIRS) {
            bool busy = alive && pending < 0;  // cur >= 0 or entries on the stack
            const int nStart = __popcll(__ballot(busy));
            if (nStart > 0) {
                const int minWalk = (nStart * (RD_LEAF_DEN - RD_LEAF_NUM) + RD_LEAF_DEN - 1) / RD_LEAF_DEN;
                const int lane = int(threadIdx.x & 63u);
                do {
                    pairPopOne<COUNT>(stk, lane, ovf, tmax, busy, cur, sp, pending, ws);
                    if (busy && cur >= 0) {
                        pairStep<COUNT>(s.pairs, stk, lane, ovf, rs, tmax, ord, cur, sp, pending, ws);
                        busy = pending < 0 && (cur >= 0 || sp.sp > 0);
                    }
                } while (__popcll(__ballot(busy)) >= (minWalk > 1 ? minWalk : 1));
                alive = alive && (pending >= 0 || cur >= 0 || sp.sp > 0);
            }
        } else
        {
            bool walking = alive && pending < 0;
            const int nStart = __popcll(__ballot(walking));
#if RD_WF_COOP_LONE
            if (nStart == 1) {  // a lone walker (the end of a stage): the whole wave tests 64 boxes ahead for it
                const int L = __ffsll((long long)__ballot(walking)) - 1;
                CoopResult cr = coopWalk(readlanePtr(nodes, L), readlaneI(node, L), end, readlaneRay(rs, L), readlaneF(tmax, L), RD_COOP_WINDOWS);
                if (int(threadIdx.x & 63u) == L) {
                    node = cr.node;
                    pending = cr.pending;
                    if (COUNT) ws.nodes += cr.visited;
                    alive = (node != end) || pending >= 0;
                }
            } else
#endif
            if (nStart > 0) {
                const int minWalk = (nStart * (RD_LEAF_DEN - RD_LEAF_NUM) + RD_LEAF_DEN - 1) / RD_LEAF_DEN;
                do {
                    if (walking) {
                        if (COUNT) ws.nodes++;
                        float4 lo = nodes[node].lo_prim;
                        float4 hi = nodes[node].hi_next;
                        float boundDist;
                        bool boundHit = aabbFast(lo, hi, rs, boundDist);
                        if (boundHit && boundDist < tmax) {
                            pending = __float_as_int(lo.w);
                            node++;
                        } else {
                            node = __float_as_int(hi.w);
                        }
                        walking = pending < 0 && node != END;
                        alive = (node != END) || pending >= 0;
                    }
                } while (__popcll(__ballot(walking)) >= (minWalk > 1 ? minWalk : 1));
            }
        }
        // ---- triangle tests of the parked lanes ----
        if (alive && pending >= 0) {
            TriVerts t = loadTri(s.tris, pending);
            float dist;
            v2 bary;
            if (COUNT) ws.tris++;
            bool hit = intersectTriangle(rs, t.a, t.b, t.c, bary, dist);
            if (hit && dist < tmax) {
                if (isShadow) {
                    occluded = true;
                    node = END;
                    cur = kPairNone;
                    sp = PairStack{0, 0};
                } else {
                    hitPrim = pending;
                    tmax = dist;
                    hitBary = bary;
                }
            }
            pending = -1;
            alive = PAIRS ? (cur >= 0 || sp.sp > 0) : (node != END);
        }
        // ---- retire finished lanes: together, once done lanes * 64 >= (walking + done) lanes * RD_WF_FINISH_MIN (the records are
        // dependent read-modify-writes; a lane that has finished keeps its path slot in p until then and is not refilled) ----
        const unsigned long long doneM = __ballot(!alive && p >= 0);
        if (doneM != 0ull && __popcll(doneM) * 64 >= (__popcll(doneM) + __popcll(__ballot(alive))) * RD_WF_FINISH_MIN)
        if (!alive && p >= 0) {
            if (isShadow) {
                float4 n = w.nee[p];
                if (!occluded && n.w >= 0.f) {  // the addition sampleDirectLight's caller makes (pathtrace.cu:201-207)
                    if (n.w == 0.f) {
                        float4 a = w.accD[p];
                        w.accD[p] = make_float4(a.x + n.x, a.y + n.y, a.z + n.z, 0.f);
                    } else {
                        float4 a = w.accI[p];
                        w.accI[p] = make_float4(a.x + n.x, a.y + n.y, a.z + n.z, 0.f);
                    }
                }
            } else {
                bool hit = hitPrim != -1;
                if (hit) nHits++;
                // .w: the shading class of the hit (shade(k)'s material sort bins by it); the distance is not needed again
                w.hit[p] = make_int4(hitPrim, __float_as_int(hit ? hitBary.x : 0.f), __float_as_int(hit ? hitBary.y : 0.f),
                                     hit ? int(s.primClass[hitPrim]) : 0);
            }
            p = -1;
        }
    }
    if (COUNT) flushCounters(s.counters, nClosest, nAny, nHits, ws);
}

// ---- shade(k): hit k → terminate or run bounce body k+1 ----------------------------------------------------------
// sorted != 0 (material sort, BASELINE config 3): the wave bins the 1 024 records of its packet by BSDF class BEFORE it
// shades them — class of every record (trace(k) left it in the hit record: one gather), a counting
// sort by ballots into a 4-KB LDS table of the wave, then the 16 chunks are shaded in class order: at most three of the
// sixteen chunks straddle a class boundary, the others run one branch of the material switch.  Round 1 had a separate pass over
// the hit records for this (k_wf_classify: four global class queues, 0.2 ms per bounce on the teapots frame — the sorted
// pipeline was 1.8 ms SLOWER than the unsorted one); the binning now costs no launch, no global queue and no extra read of the
// hit records.  Classes: 0 = terminal (miss / emitter / other), 1 = Lambertian, 2 = metallic workflow, 3 = dielectric.
constexpr int kSortChunks = 16;                  // chunks of 64 records per sorted packet
constexpr int kSortPacket = 64 * kSortChunks;    // records a wave bins at a time
#ifndef RD_WF_SHADE_WAVES
#define RD_WF_SHADE_WAVES 1  // waves per SIMD k_wf_shade is held to (1: whatever its 155 VGPRs allow = 3)
#endif
__global__ __launch_bounds__(256, RD_WF_SHADE_WAVES) void k_wf_shade(DScene s, WaveWorkspace w, int k, int maxDepth, int sorted) {
    __shared__ int sSorted[4][kSortPacket];  // per wave of the workgroup: the packet's path slots in class order
    WaveCounters *c = w.ctr;
    const int lane = int(threadIdx.x & 63u);
    int *mySorted = sSorted[threadIdx.x >> 6];
    {
        const int q = 0;
        const int n = c->rayCount[k].v;
        const int *hitq = w.rayq[k & 1];
        // queue entries of the packet's chunks, held until its last chunk has been shaded, then appended together
        int pendP[kShadeChunks], pendQ[kShadeChunks];  // path slot; its ray-queue entry (-2 - slot: trace(k+1) has it on its first list)
        bool pendShadow[kShadeChunks], pendRay[kShadeChunks];
#pragma unroll
        for (int j = 0; j < kShadeChunks; j++) {
            pendP[j] = pendQ[j] = -1;
            pendShadow[j] = pendRay[j] = false;
        }
        // unsorted: packets of kShadePacket = 256 records.  sorted: 256 .. kSortPacket = 1 024 records (multiples of 256), as large
        // as still leaves every wave about two packets — with one big packet per wave there is no dynamic balancing left and a
        // short queue (the Cornell frame after bounce 1) would keep nine waves in ten idle while the others shade 16 chunks each
        int packet = kShadePacket;
        if (sorted) {
            const int perWave = n / (2 * gridWaves());
            packet = ((perWave + kShadePacket - 1) / kShadePacket) * kShadePacket;
            packet = packet < kShadePacket ? kShadePacket : (packet > kSortPacket ? kSortPacket : packet);
        }
        int packetEnd = 0;  // sorted: records of the current packet that exist (<= kSortPacket)
        for (int base = globalWave() * packet, sub = 0, fresh = 1;;) {
            if (sub == packet) {
                base = wavePull(&c->shadeHead[k][q].v, packet);
                sub = 0;
                fresh = 1;
            }
            if (base + sub >= n) break;
            if (sorted && fresh) {  // bin this packet: counting sort of up to 1 024 records by class
                fresh = 0;
                packetEnd = (n - base) < packet ? (n - base) : packet;
                int pp[kSortChunks], cls[kSortChunks];
#pragma unroll
                for (int j = 0; j < kSortChunks; j++) {
                    const int i = base + j * 64 + lane;
                    pp[j] = (j * 64 + lane < packetEnd) ? hitq[i] : -1;
                    if (pp[j] <= -2) pp[j] = -2 - pp[j];  // a ray trace(k) started first (literal-class list): same path slot
                }
#pragma unroll
                for (int j = 0; j < kSortChunks; j++) {
                    // class of the hit, as trace(k) recorded it; off-frame slots of bounce 0 (p = -1) sort with the terminal
                    // class and are skipped below
                    cls[j] = (pp[j] >= 0) ? w.hit[pp[j]].w : 0;
                }
                int at = 0;  // running position in class order (wave-uniform)
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
#pragma unroll
                    for (int j = 0; j < kSortChunks; j++) {
                        const bool in = (j * 64 + lane < packetEnd) && cls[j] == qq;
                        const unsigned long long m = __ballot(in);
                        if (in) mySorted[at + __popcll(m & laneMaskLt())] = pp[j];
                        at += __popcll(m);
                    }
                }
                __builtin_amdgcn_wave_barrier();  // one wave writes and then reads its own table: LDS executes a wave's ops in order
            }
            int item = base + sub + lane;
            const int subNow = sub;
            sub += 64;
            bool active = sorted ? (subNow + lane < packetEnd) : (item < n);
            bool emitShadow = false, emitRay = false, litShadow = false, litRay = false, rayFirst = false;
            int p = -1;
            if (active) {
                p = sorted ? mySorted[subNow + lane] : hitq[item];
                if (p <= -2) p = -2 - p;  // (see the binning above)
                active = p >= 0;
            }
            if (active) {
                int4 h = w.hit[p];
                float4 rdw = w.rd[p];
                v3 rayDir = mk3(rdw.x, rdw.y, rdw.z);
                do {
                    if (h.x == -1) {  // miss: primary → direct = 1 (pathtrace.cu:169-172); later → env map (:232-247)
                        if (k == 0) {
                            w.accD[p] = make_float4(1.f, 1.f, 1.f, 0.f);
                        } else if (hasEnvMap(s)) {
                            float4 t = w.thr[p], o = w.ro[p];
                            v3 radiance = envLookup(s, rayDir) * mk3(t.x, t.y, t.z);
                            float weight = (rdw.w != 0.f) ? 1.f : powerHeuristic(o.w, environmentMapPdf(s, rayDir));
                            v3 add = radiance * weight;
                            float4 a = w.accI[p];
                            w.accI[p] = make_float4(a.x + add.x, a.y + add.y, a.z + add.z, 0.f);
                        }
                        break;
                    }
                    Surface isec;
                    fetchSurface(s, h.x, mk2(__int_as_float(h.y), __int_as_float(h.z)), isec);
                    Material material = texturedMaterial(s, isec);
                    v3 throughput;
                    if (k == 0) {
                        material.baseColor = mk3(1.f);  // DENOISER_DEMODULATE (:175-178)
                        if (material.type == Light) {   // :179-182
                            w.accD[p] = make_float4(1.f, 1.f, 1.f, 0.f);
                            break;
                        }
                        throughput = mk3(1.f);
                    } else {
                        float4 t = w.thr[p];
                        throughput = mk3(t.x, t.y, t.z);
                        if (material.type == Light) {  // :251-271
                            if (dot(isec.norm, rayDir) < 0.f) break;
                            v3 radiance = material.baseColor;
                            float4 o = w.ro[p], cp = w.prevPos[p];
                            bool deltaSample = rdw.w != 0.f;
                            float weight = deltaSample
                                               ? 1.f
                                               : powerHeuristic(o.w, pdfAreaToSolidAngle(luminance(radiance) * s.sumLightPowerInv *
                                                                                            getPrimitiveArea(s, isec.primId),
                                                                                        mk3(cp.x, cp.y, cp.z), isec.pos, isec.norm));
                            v3 add = radiance * throughput * weight;
                            float4 a = w.accI[p];
                            w.accI[p] = make_float4(a.x + add.x, a.y + add.y, a.z + add.z, 0.f);
                            break;
                        }
                    }
                    const int depth = k + 1;
                    if (depth > maxDepth) break;  // loop bound of pathtrace.cu:187
                    isec.wo = -rayDir;
                    bool deltaBSDF = (material.type == Dielectric);
                    if (material.type != Dielectric && dot(isec.norm, isec.wo) < 0.f) isec.norm = -isec.norm;
                    uint2 rs = w.rng[p];
                    Sampler rng{s.sobol, rs.x, (int)rs.y};
                    if (!deltaBSDF) {  // NEE (:195-208); the shadow ray itself is traced by trace(k+1)
                        v4 r4 = sample4D(rng);
                        if (s.lightSamplerLength != 0) {
                            LightPick lp = pickLightPoint(s, isec.pos, r4);
                            v3 radiance = mk3(0.f), wi = mk3(0.f);
                            float lightPdf = lightPdfUnoccluded(s, isec.pos, lp, radiance, wi);
                            float4 n = make_float4(0.f, 0.f, 0.f, -1.f);
                            if (lightPdf > 0.f) {
                                float BSDFPdf = materialPdf(material, isec.norm, isec.wo, wi);
                                v3 cc = throughput * materialBSDF(material, isec.norm, isec.wo, wi) * radiance *
                                        satDot(isec.norm, wi) / lightPdf * powerHeuristic(lightPdf, BSDFPdf);
                                n = make_float4(cc.x, cc.y, cc.z, depth == 1 ? 0.f : 1.f);
                            }
                            w.nee[p] = n;
                            w.sht[p] = make_float4(lp.sampled.x, lp.sampled.y, lp.sampled.z, 0.f);
                            emitShadow = true;  // traced even when nothing can be added: sampleDirectLight tests
                                                // occlusion before the single-sided rejection
                            {  // a hint only (trace classifies the ray it builds): some component below 1.1e-6 of the length
                                const v3 dl = lp.sampled - isec.pos;
                                const float l2 = dot(dl, dl) * 1.21e-12f;
                                litShadow = dl.x * dl.x < l2 || dl.y * dl.y < l2 || dl.z * dl.z < l2;
                            }
                        }
                    }
                    w.prevPos[p] = make_float4(isec.pos.x, isec.pos.y, isec.pos.z, 0.f);
                    BSDFSample sample;
                    sample.pdf = 0.f;
                    materialSample(material, isec.norm, isec.wo, sample3D(rng), sample);
                    if (sample.type == Invalid) break;
                    else if (sample.pdf < 1e-8f) break;
                    bool deltaSample = (sample.type & Specular) != 0;
                    throughput = throughput * (sample.bsdf / sample.pdf * (deltaSample ? 1.f : absDot(isec.norm, sample.dir)));
                    Ray ray = makeOffsetedRay(isec.pos, sample.dir);
                    w.ro[p] = make_float4(ray.o.x, ray.o.y, ray.o.z, sample.pdf);
                    w.rd[p] = make_float4(ray.d.x, ray.d.y, ray.d.z, deltaSample ? 1.f : 0.f);
                    w.thr[p] = make_float4(throughput.x, throughput.y, throughput.z, 0.f);
                    w.rng[p] = make_uint2(rng.scramble, (unsigned)rng.ptr);
                    emitRay = true;
                    litRay = fabs_(ray.d.x) < 1.1e-6f || fabs_(ray.d.y) < 1.1e-6f || fabs_(ray.d.z) < 1.1e-6f;
                } while (false);
            }
            if (__ballot(litShadow || litRay) != 0ull) {  // rare: a few hundred rays per stage
                int *lq = w.litq[(k + 1) & 1];
                int *lc = &c->litCount[k + 1].v;
                for (int kind = 0; kind < 2; kind++) {
                    const bool pred = kind == 0 ? litShadow : litRay;
                    const unsigned long long m = __ballot(pred);
                    if (m == 0ull) continue;
                    int at = 0;
                    if (lane == 0) at = atomicAdd(lc, __popcll(m));
                    at = __shfl(at, 0, 64) + __popcll(m & laneMaskLt());
                    if (pred && at < w.litCap) {  // beyond the list's capacity the ray stays in its ordinary queue
                        lq[at] = kind == 0 ? (p | kWfLitShadow) : p;
                        if (kind == 0) emitShadow = false;  // nobody else reads the shadow queue
                        else rayFirst = true;  // the ray queue is also shade(k+1)'s list of hits: the entry stays, marked as traced
                    }
                }
            }
            {
                const int j = ((sub >> 6) - 1) & (kShadeChunks - 1);  // chunk just shaded, within its group of four (wave-uniform)
#pragma unroll
                for (int jj = 0; jj < kShadeChunks; jj++)
                    if (jj == j) {
                        pendP[jj] = p;
                        pendQ[jj] = rayFirst ? -2 - p : p;
                        pendShadow[jj] = emitShadow;
                        pendRay[jj] = emitRay;
                    }
            }
            // every fourth chunk, and at the last chunk of the queue (the loop then leaves or pulls the next packet): append
            if ((sub & (kShadePacket - 1)) == 0 || base + sub >= n) {
                waveAppendN(pendShadow, pendP, w.shadowq, &c->shadowCount[k].v);
                waveAppendN(pendRay, pendQ, w.rayq[(k + 1) & 1], &c->rayCount[k + 1].v);
#pragma unroll
                for (int jj = 0; jj < kShadeChunks; jj++) pendShadow[jj] = pendRay[jj] = false;
            }
        }
    }
}

// ---- finish: NaN scrub, HDRToLDR, running mean (pathtrace.cu:279-290) ------------------------------------------
__global__ __launch_bounds__(256) void k_wf_finish(PixelMap pm, WaveWorkspace w, int iter, float *__restrict__ directIllum,
                                                   float *__restrict__ indirectIllum, int part, int parts) {
    unsigned wg;
    const unsigned nLocal = subFrameBlocks(pm, part, parts);
    bool wgValid = xcdSwizzle(blockIdx.x, (nLocal + 3u) >> 2, wg);
    unsigned lane = threadIdx.x & 63u;
    unsigned blockLocal = wgValid ? wg * 4u + (threadIdx.x >> 6) : 0xffffffffu / 64u;
    const bool inRange = wgValid && blockLocal < nLocal;
    Pix px = mapPixel(pm, inRange ? blockLocal * (unsigned)parts + (unsigned)part : 0xffffffffu / 64u, lane);
    if (!(px.valid && inRange)) return;
    int p = int(blockLocal * 64u + lane);
    float4 a = w.accD[p], b = w.accI[p];
    v3 direct = mk3(a.x, a.y, a.z), indirect = mk3(b.x, b.y, b.z);
    if (hasNanOrInf(direct)) direct = mk3(0.f);
    if (hasNanOrInf(indirect)) indirect = mk3(0.f);
    direct = HDRToLDR(direct);
    indirect = HDRToLDR(indirect);
    storeRunningMean(directIllum, px.out, direct, iter);
    storeRunningMean(indirectIllum, px.out, indirect, iter);
}

}  // namespace rd
