// radish_pt_amd/csrc/device/kernels_walk.h — a persistent WALK-ONLY kernel over a ray batch: box steps, leaf tests, retire, and
// nothing else in its register allocation.
//
// k_trace_closest / k_trace_occluded (kernels_pt.h) give every ray a lane for the whole launch, so each wave waits for its
// longest ray.  Here a lane whose ray has ended writes its record and takes the next ray of the batch (the lane-refill
// structure of k_gbuffer_persistent, without a surface fetch).  DevScene::intersect (scene.h:262-301) /
// DevScene::testOcclusion (:303-334) per lane, exactly as walkRay (traverse.h) performs them: same decisions, same counters.
// Rays are handed out in chunks of 64 (the first chunk of a wave is its index, further ones come from an atomic counter).
#pragma once
#include "kernels_persist.h"

namespace rd {

#ifndef RD_WALK_REFILL_MIN
#define RD_WALK_REFILL_MIN 16  // refill once this many lanes are idle
#endif
#ifndef RD_WALK_PAIR_FETCH
#define RD_WALK_PAIR_FETCH 0  // 1: lane pairs share the two requests of a box step (traverse.h, fetchNodePaired)
#endif
#ifndef RD_WALK_FINISH_MIN
#define RD_WALK_FINISH_MIN 8  // retire once done lanes * 64 >= busy lanes * this
#endif

// DEFER (ReSTIR's ray lists): literal-class rays — one wave needs ~1 ms for one of them on the teapots scene, which is then the
// duration of the whole launch — have been listed by the kernel that wrote the rays (at most kWalkDeferCap of them, else none is
// set aside) and are traced by k_trace_wg_list, one 1 024-thread workgroup each, on a second stream beside this launch.
constexpr int kWalkDeferCap = 256;
template <bool COUNT, bool ANY, bool DEFER = false>
__global__ __launch_bounds__(64) void k_walk_persistent(DScene s, const float *__restrict__ rays, long long n, int4 *__restrict__ hits,
                                                        int *__restrict__ occluded, PersistCounters *pc,
                                                        const int *__restrict__ deferCount = nullptr, int slotList = 0) {
    const int lane = int(threadIdx.x) & 63;
    const int end = s.bvhSize;
    const long long chunks = (n + 63) / 64;
    WalkStats ws{0, 0};
    unsigned nRays = 0, nHits = 0;
    long long curChunk = (long long)blockIdx.x;
    int slotNext = 0;
    const int gridWavesN = int(gridDim.x);
    bool exhausted = curChunk >= chunks;

    constexpr int W_IDLE = 0, W_TRACE = 1, W_DONE = 2;
    int state = W_IDLE;
    long long rayIdx = 0;
    RaySlab rs;
    rs.o = rs.d = rs.inv = mk3(0.f);
    rs.cls = 0;
    const NodeRec *nodes = s.nodes[0];
#if RD_WALK_PAIR_FETCH
    const char *nodeBase = reinterpret_cast<const char *>(s.nodes[0]);  // all six orderings are one allocation (layouts.h)
    const unsigned ordStride = (unsigned)(s.bvhSize + 1) * (unsigned)sizeof(NodeRec);
    unsigned ordOfs = 0;
    const bool oddLane = (lane & 1) != 0;
#endif
    int node = end, pending = -1;
    float tmax = 0.f;
    int hitPrim = -1;
    v2 hitBary = mk2(0.f, 0.f);
    bool found = false;
    const bool deferAll = DEFER && *deferCount <= kWalkDeferCap;  // written by the kernel that produced the list, earlier in the stream

    for (;;) {
        // ---------------- new rays for idle lanes ----------------
        const unsigned long long idleM = __ballot(state == W_IDLE);
        const int nIdle = __popcll(idleM);
        if (!exhausted && nIdle >= RD_WALK_REFILL_MIN) {
            const int myRank = __popcll(idleM & laneMaskLt());
            int taken = 0;
            while (taken < nIdle && !exhausted) {
                if (slotNext == 64) {
                    int b = 0;
                    if (lane == 0) b = atomicAdd(&pc->blockHead, 1);
                    curChunk = (long long)__shfl(b, 0, 64) + gridWavesN;
                    slotNext = 0;
                    if (curChunk >= chunks) {
                        exhausted = true;
                        break;
                    }
                }
                const int avail = 64 - slotNext;
                const int give = (nIdle - taken) < avail ? (nIdle - taken) : avail;
                if (state == W_IDLE && myRank >= taken && myRank < taken + give) {
                    const long long i = curChunk * 64 + slotNext + (myRank - taken);
                    // slotList (ReSTIR's per-slot ray lists only): a NaN in the first float marks an EMPTY SLOT — no ray, no record,
                    // nothing counted (the consumer, k_restir_resolve, does not read the record of such a slot either).  The public
                    // ray-batch entries (rdh_trace_closest / rdh_trace_occluded) pass slotList = 0: a caller's ray with a NaN origin
                    // is a ray like any other (it fails the root's box test, as in DevScene::intersect) and is counted.
                    if (i < n && (slotList == 0 || rays[6 * i] == rays[6 * i])) {
                        rayIdx = i;
                        const v3 a = mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
                        const v3 b = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
                        Ray ray;
                        if (ANY) {  // testOcclusion(a, b) (scene.h:303-315)
                            v3 dir = b - a;
                            float dist = length(dir);
                            dir = dir / dist;
                            ray = makeOffsetedRay(a, dir);
                            tmax = dist - 1e-4f;
                        } else {
                            ray = Ray{a, b};
                            tmax = 3.402823466e+38f;
                        }
                        rs = makeRaySlab(ray);
                        nodes = s.nodes[getMTBVHId(-ray.d)];
#if RD_WALK_PAIR_FETCH
                        ordOfs = (unsigned)getMTBVHId(-ray.d) * ordStride;
#endif
                        node = 0;
                        pending = -1;
                        hitPrim = -1;
                        hitBary = mk2(0.f, 0.f);
                        found = false;
                        if (!(DEFER && deferAll && raySetAside(rs.cls) && end != 0)) {  // else: k_trace_wg_list has this ray
                            nRays++;
                            state = W_TRACE;
                        }
                    }
                }
                slotNext += give;
                taken += give;
            }
        }
        if (__ballot(state != W_IDLE) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---------------- literal-class rays: traced whole by the whole wave ----------------
        {
            unsigned long long lit = __ballot(state == W_TRACE && rs.cls != 0 && node == 0 && pending < 0 && end != 0);
            while (lit) {
                const int L = __ffsll((long long)lit) - 1;
                lit &= lit - 1ull;
                CoopTrace ct = coopTraceWhole<ANY>(s, readlanePtr(nodes, L), readlaneRay(rs, L), readlaneF(tmax, L));
                if (lane == L) {
                    hitPrim = ct.hitPrim;
                    hitBary = ct.bary;
                    tmax = ct.tmax;
                    found = ct.found;
                    node = end;
                    if (COUNT) {
                        ws.nodes += ct.nodes;
                        ws.tris += ct.tris;
                    }
                }
            }
        }
        // ---------------- box steps ----------------
        {
            bool walking = state == W_TRACE && pending < 0 && node != end;
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
#if RD_WALK_PAIR_FETCH
                    float4 lo, hi;  // the record's two corners, in either order (see fetchNodePaired)
                    int recPrim, recNext;
                    fetchNodePaired(nodeBase, ordOfs + ((unsigned)node << 5), walking, oddLane, lo, hi, recPrim, recNext);
#endif
                    if (walking) {
#if !RD_WALK_PAIR_FETCH
                        float4 lo = nodes[node].lo_prim;
                        float4 hi = nodes[node].hi_next;
                        const int recPrim = __float_as_int(lo.w), recNext = __float_as_int(hi.w);
#endif
                        float boundDist;
                        if (COUNT) ws.nodes++;
                        bool boundHit = aabbFast(lo, hi, rs, boundDist);
                        if (boundHit && boundDist < tmax) {
                            pending = recPrim;
                            node++;
                        } else {
                            node = recNext;
                        }
                        walking = pending < 0 && node != end;
                    }
                } while (__popcll(__ballot(walking)) >= (minWalk > 1 ? minWalk : 1));
            }
        }
        // ---------------- leaf tests ----------------
        if (state == W_TRACE && pending >= 0) {
            TriVerts tv = loadTri(s.tris, pending);
            float dist;
            v2 bary;
            if (COUNT) ws.tris++;
            bool hit = intersectTriangle(rs, tv.a, tv.b, tv.c, bary, dist);
            if (hit && dist < tmax) {
                if (ANY) {
                    found = true;
                    node = end;
                } else {
                    hitPrim = pending;
                    tmax = dist;
                    hitBary = bary;
                }
            }
            pending = -1;
        }
        if (state == W_TRACE && pending < 0 && node == end) state = W_DONE;
        // ---------------- records of finished rays ----------------
        {
            const unsigned long long doneM = __ballot(state == W_DONE);
            const int nBusy = __popcll(__ballot(state != W_IDLE));
            if (doneM != 0ull && __popcll(doneM) * 64 >= nBusy * RD_WALK_FINISH_MIN) {
                if (state == W_DONE) {
                    if (ANY) {
                        occluded[rayIdx] = found ? 1 : 0;
                    } else {
                        const bool hit = hitPrim != -1;
                        if (hit) nHits++;
                        hits[rayIdx] = make_int4(hitPrim, __float_as_int(hit ? hitBary.x : 0.f), __float_as_int(hit ? hitBary.y : 0.f),
                                                 __float_as_int(hit ? tmax : 3.402823466e+38f));
                    }
                    state = W_IDLE;
                }
            }
        }
    }
    if (COUNT) flushCounters(s.counters, ANY ? 0u : nRays, ANY ? nRays : 0u, nHits, ws);
}

// ---- closest hits of a COHERENT ray list — consecutive rays start side by side, e.g. the primary rays of 8x8 pixel blocks in slot
// order — one 64-ray chunk per wave, walked as a PACKET (traverse.h, packetWalk): one uniform node load per visit for the whole wave
// instead of one L1 request per lane.  Same records, same counters as the lane-refill walkers; on rays that do not travel together
// it is merely slow (a wave then visits the union of 64 unrelated walks), so the caller says when a list is coherent.
// Literal-class rays: set aside for k_trace_wg_list when the producer listed them (DEFER), else traced whole by the wave, in place.
template <bool COUNT, bool DEFER>
__global__ __launch_bounds__(256) void k_walk_packet(DScene s, const float *__restrict__ rays, long long n, int4 *__restrict__ hits,
                                                     const int *__restrict__ deferCount, int slotList, int budget) {
    const int lane = int(threadIdx.x) & 63;
    const long long i = ((long long)blockIdx.x * 4 + (long long)(threadIdx.x >> 6)) * 64 + lane;
    const int end = s.bvhSize;
    WalkStats ws{0, 0};
    unsigned nRays = 0, nHits = 0;
    const bool deferAll = DEFER && *deferCount <= kWalkDeferCap;
    const bool have = i < n && (slotList == 0 || rays[6 * i] == rays[6 * i]);  // see k_walk_persistent
    Ray ray{mk3(0.f), mk3(0.f, 0.f, 1.f)};
    if (have) ray = Ray{mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5])};
    const RaySlab rs = makeRaySlab(ray);
    const int ord = getMTBVHId(-ray.d);
    float tmax = 3.402823466e+38f;
    int hitPrim = -1;
    v2 hitBary = mk2(0.f, 0.f);
    const bool traced = have && !(DEFER && deferAll && raySetAside(rs.cls) && end != 0);  // else: k_trace_wg_list has this ray
    if (traced) nRays++;
    unsigned long long lit = __ballot(traced && rs.cls != 0 && end != 0);
    while (lit) {
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
        const bool hit = hitPrim != -1;
        if (hit) nHits++;
        hits[i] = make_int4(hitPrim, __float_as_int(hit ? hitBary.x : 0.f), __float_as_int(hit ? hitBary.y : 0.f),
                            __float_as_int(hit ? tmax : 3.402823466e+38f));
    }
    if (COUNT) flushCounters(s.counters, nRays, 0u, nHits, ws);
}

// ---- the same walker over SIBLING PAIRS (DScene::pairs; traverse.h, pairStep) ---------------------------------------------------
template <bool COUNT, bool ANY, bool DEFER = false>
__global__ __launch_bounds__(64, COUNT ? 1 : 7) void k_walk_pair(DScene s, const float *__restrict__ rays, long long n, int4 *__restrict__ hits,
                                                  int *__restrict__ occluded, PersistCounters *pc, int2 *__restrict__ overflow,
                                                  int overflowDepth, const int *__restrict__ deferCount = nullptr, int slotList = 0) {
    __shared__ int2 stk[kPairLds * 64];
    const int lane = int(threadIdx.x) & 63;
    const int end = s.bvhSize;
    const long long chunks = (n + 63) / 64;
    WalkStats ws{0, 0};
    unsigned nRays = 0, nHits = 0;
    long long curChunk = (long long)blockIdx.x;
    int slotNext = 0;
    const int gridWavesN = int(gridDim.x);
    bool exhausted = curChunk >= chunks;
    int2 *const ovf = overflow + (size_t)blockIdx.x * (size_t)overflowDepth * 64;

    constexpr int W_IDLE = 0, W_TRACE = 1, W_DONE = 2;
    int state = W_IDLE;
    long long rayIdx = 0;
    RaySlab rs;
    rs.o = rs.d = rs.inv = mk3(0.f);
    rs.cls = 0;
    int ord = 0;
    int cur = kPairNone, pending = -1;
    PairStack sp{0, 0};
    float tmax = 0.f;
    int hitPrim = -1;
    v2 hitBary = mk2(0.f, 0.f);
    bool found = false;
    const bool deferAll = DEFER && *deferCount <= kWalkDeferCap;

    for (;;) {
        // ---------------- new rays for idle lanes ----------------
        const unsigned long long idleM = __ballot(state == W_IDLE);
        const int nIdle = __popcll(idleM);
        if (!exhausted && nIdle >= RD_WALK_REFILL_MIN) {
            const int myRank = __popcll(idleM & laneMaskLt());
            int taken = 0;
            while (taken < nIdle && !exhausted) {
                if (slotNext == 64) {
                    int b = 0;
                    if (lane == 0) b = atomicAdd(&pc->blockHead, 1);
                    curChunk = (long long)__shfl(b, 0, 64) + gridWavesN;
                    slotNext = 0;
                    if (curChunk >= chunks) {
                        exhausted = true;
                        break;
                    }
                }
                const int avail = 64 - slotNext;
                const int give = (nIdle - taken) < avail ? (nIdle - taken) : avail;
                if (state == W_IDLE && myRank >= taken && myRank < taken + give) {
                    const long long i = curChunk * 64 + slotNext + (myRank - taken);
                    if (i < n && (slotList == 0 || rays[6 * i] == rays[6 * i])) {  // see k_walk_persistent
                        rayIdx = i;
                        const v3 a = mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
                        const v3 b = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
                        Ray ray;
                        if (ANY) {  // testOcclusion(a, b) (scene.h:303-315)
                            v3 dir = b - a;
                            float dist = length(dir);
                            dir = dir / dist;
                            ray = makeOffsetedRay(a, dir);
                            tmax = dist - 1e-4f;
                        } else {
                            ray = Ray{a, b};
                            tmax = 3.402823466e+38f;
                        }
                        rs = makeRaySlab(ray);
                        ord = getMTBVHId(-ray.d);
                        sp = PairStack{0, 0};
                        hitPrim = -1;
                        hitBary = mk2(0.f, 0.f);
                        found = false;
                        if (!(DEFER && deferAll && raySetAside(rs.cls) && end != 0)) {  // else: k_trace_wg_list has this ray
                            nRays++;
                            state = W_TRACE;
                            if (rs.cls == 0 || end == 0) {
                                pairStart<COUNT>(s, rs, tmax, cur, pending, ws);
                            } else {
                                cur = kPairFresh;
                                pending = -1;
                            }
                        }
                    }
                }
                slotNext += give;
                taken += give;
            }
        }
        if (__ballot(state != W_IDLE) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---------------- literal-class rays: traced whole by the whole wave, over the threaded array of their ordering ----------------
        {
            unsigned long long lit = __ballot(state == W_TRACE && cur == kPairFresh);
            while (lit) {
                const int L = __ffsll((long long)lit) - 1;
                lit &= lit - 1ull;
                const NodeRec *nodes = s.nodes[0] + (size_t)readlaneI(ord, L) * (size_t)(end + 1);
                CoopTrace ct = coopTraceWhole<ANY>(s, nodes, readlaneRay(rs, L), readlaneF(tmax, L));
                if (lane == L) {
                    hitPrim = ct.hitPrim;
                    hitBary = ct.bary;
                    tmax = ct.tmax;
                    found = ct.found;
                    cur = kPairNone;
                    if (COUNT) {
                        ws.nodes += ct.nodes;
                        ws.tris += ct.tris;
                    }
                }
            }
        }
        // ---------------- box steps ----------------
        {
            bool busy = state == W_TRACE && pending < 0 && (cur >= 0 || sp.sp > 0);
            const int nStart = __popcll(__ballot(busy));
            if (nStart > 0) {
                const int minWalk = (nStart * (RD_LEAF_DEN - RD_LEAF_NUM) + RD_LEAF_DEN - 1) / RD_LEAF_DEN;
                do {
                    pairPopOne<COUNT>(stk, lane, ovf, tmax, busy, cur, sp, pending, ws);
                    if (busy && cur >= 0) {
                        pairStep<COUNT>(s.pairs, stk, lane, ovf, rs, tmax, ord, cur, sp, pending, ws);
                        busy = pending < 0 && (cur >= 0 || sp.sp > 0);
                    }
                } while (__popcll(__ballot(busy)) >= (minWalk > 1 ? minWalk : 1));
            }
        }
        // ---------------- leaf tests ----------------
        if (state == W_TRACE && pending >= 0) {
            TriVerts tv = loadTri(s.tris, pending);
            float dist;
            v2 bary;
            if (COUNT) ws.tris++;
            bool hit = intersectTriangle(rs, tv.a, tv.b, tv.c, bary, dist);
            if (hit && dist < tmax) {
                if (ANY) {
                    found = true;
                    cur = kPairNone;
                    sp = PairStack{0, 0};
                } else {
                    hitPrim = pending;
                    tmax = dist;
                    hitBary = bary;
                }
            }
            pending = -1;
        }
        if (state == W_TRACE && pending < 0 && cur == kPairNone && sp.sp == 0) state = W_DONE;
        // ---------------- records of finished rays ----------------
        {
            const unsigned long long doneM = __ballot(state == W_DONE);
            const int nBusy = __popcll(__ballot(state != W_IDLE));
            if (doneM != 0ull && __popcll(doneM) * 64 >= nBusy * RD_WALK_FINISH_MIN) {
                if (state == W_DONE) {
                    if (ANY) {
                        occluded[rayIdx] = found ? 1 : 0;
                    } else {
                        const bool hit = hitPrim != -1;
                        if (hit) nHits++;
                        hits[rayIdx] = make_int4(hitPrim, __float_as_int(hit ? hitBary.x : 0.f), __float_as_int(hit ? hitBary.y : 0.f),
                                                 __float_as_int(hit ? tmax : 3.402823466e+38f));
                    }
                    state = W_IDLE;
                }
            }
        }
    }
    if (COUNT) flushCounters(s.counters, ANY ? 0u : nRays, ANY ? nRays : 0u, nHits, ws);
}

// One WORKGROUP per ray (wg_trace.h) over a ray batch: RDH_PT_WG_PER_RAY on the ray-batch entries.  This is how the G-buffer
// traces its literal-class rays; as an entry of its own it lets the tests put every kind of ray
// through wgTraceWhole (closest and any-hit) and compare records and counters with the oracle.  Far slower than the walkers
// for ordinary rays: 16 waves per ray.
template <bool COUNT, bool ANY>
__global__ __launch_bounds__(kWgTraceThreads) void k_trace_wg(DScene s, const float *__restrict__ rays, long long n, int4 *__restrict__ hits,
                                                              int *__restrict__ occluded) {
    __shared__ WgTraceShared sh;
    for (long long i = (long long)blockIdx.x; i < n; i += (long long)gridDim.x) {
        const v3 a = mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
        const v3 b = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        Ray ray;
        float tmax;
        if (ANY) {  // testOcclusion(a, b) (scene.h:303-315)
            v3 dir = b - a;
            float dist = length(dir);
            dir = dir / dist;
            ray = makeOffsetedRay(a, dir);
            tmax = dist - 1e-4f;
        } else {
            ray = Ray{a, b};
            tmax = 3.402823466e+38f;
        }
        const RaySlab rs = makeRaySlab(ray);
        CoopTrace ct = wgTraceWhole<ANY>(s, s.nodes[getMTBVHId(-ray.d)], rs, tmax, sh);
        if (threadIdx.x == 0) {
            if (ANY) {
                occluded[i] = ct.found ? 1 : 0;
            } else {
                const bool hit = ct.hitPrim != -1;
                hits[i] = make_int4(ct.hitPrim, __float_as_int(hit ? ct.bary.x : 0.f), __float_as_int(hit ? ct.bary.y : 0.f),
                                    __float_as_int(hit ? ct.tmax : 3.402823466e+38f));
            }
            if (COUNT) {
                atomicAdd(ANY ? &s.counters->anyRays : &s.counters->closestRays, 1ull);
                atomicAdd(&s.counters->nodeVisits, (unsigned long long)ct.nodes);
                atomicAdd(&s.counters->triTests, (unsigned long long)ct.tris);
                if (!ANY && ct.hitPrim != -1) atomicAdd(&s.counters->closestHits, 1ull);
            }
        }
    }
}

// k_trace_wg over an index list (the literal-class rays a producer kernel set aside): list[0 .. *count) are indices into `rays`.
template <bool COUNT, bool ANY>
__global__ __launch_bounds__(kWgTraceThreads) void k_trace_wg_list(DScene s, const float *__restrict__ rays, const int *__restrict__ list,
                                                                   const int *__restrict__ count, int4 *__restrict__ hits,
                                                                   int *__restrict__ occluded) {
    __shared__ WgTraceShared sh;
    const int n = *count;
    if (n > kWalkDeferCap) return;  // the walker traces them all in place
    for (int k = int(blockIdx.x); k < n; k += int(gridDim.x)) {
        const long long i = list[k];
        const v3 a = mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
        const v3 b = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        Ray ray;
        float tmax;
        if (ANY) {
            v3 dir = b - a;
            float dist = length(dir);
            dir = dir / dist;
            ray = makeOffsetedRay(a, dir);
            tmax = dist - 1e-4f;
        } else {
            ray = Ray{a, b};
            tmax = 3.402823466e+38f;
        }
        const RaySlab rs = makeRaySlab(ray);
        CoopTrace ct = wgTraceWhole<ANY>(s, s.nodes[getMTBVHId(-ray.d)], rs, tmax, sh);
        if (threadIdx.x == 0) {
            if (ANY) {
                occluded[i] = ct.found ? 1 : 0;
            } else {
                const bool hit = ct.hitPrim != -1;
                hits[i] = make_int4(ct.hitPrim, __float_as_int(hit ? ct.bary.x : 0.f), __float_as_int(hit ? ct.bary.y : 0.f),
                                    __float_as_int(hit ? ct.tmax : 3.402823466e+38f));
            }
            if (COUNT) {
                atomicAdd(ANY ? &s.counters->anyRays : &s.counters->closestRays, 1ull);
                atomicAdd(&s.counters->nodeVisits, (unsigned long long)ct.nodes);
                atomicAdd(&s.counters->triTests, (unsigned long long)ct.tris);
                if (!ANY && ct.hitPrim != -1) atomicAdd(&s.counters->closestHits, 1ull);
            }
        }
    }
}
}  // namespace rd
