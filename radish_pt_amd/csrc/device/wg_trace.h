// radish_pt_amd/csrc/device/wg_trace.h — one ray traced by a WHOLE WORKGROUP (1 024 threads).
//
// Literal-class rays (a direction component that is tiny, axis-parallel or non-finite: the reference's box test ignores
// one slab for them, bvh.h:138-148) wander through ~10 % of the tree.  Traced by one wave (coopTraceWhole, traverse.h) the
// worst primary ray of the teapots frame needs 605 windows of 64 records, almost every one a dependent fetch: 1.2 ms — and
// that ONE ray is then the duration of the whole 2-M-ray G-buffer pass.  Kernels that can set such a ray aside (one ray per
// pixel and no path to continue: the G-buffer) hand it to this routine in a second, tiny launch: thread i tests record
// base + i of a 1 024-record window (box, and the leaf's triangle straight away); the walk through the window is then
// resolved in parallel: every record's single step (descend / stop on an accepted triangle / follow the miss link) is composed
// by pointer jumping inside each wave, and wave 0 hops across the waves' results.  (A serial resolve by wave 0 — one LDS
// round trip per miss link — was measured first: slower than the one-wave trace.)
// An accepted hit shrinks tmax: every thread re-evaluates its `boundDist < closestDist` / `dist < closestDist` bits and wave
// 0 continues.  Same decisions, same counters as the sequential walk.
// ALL threads of the workgroup must call this with identical arguments (uniform control flow: it contains barriers).
#pragma once
#include "traverse.h"

namespace rd {

#ifndef RD_WG_TRACE_THREADS
#define RD_WG_TRACE_THREADS 1024  // threads = records per window of the workgroup-per-ray launches (256 / 512 measured: profiles/r03_z3_*)
#endif
constexpr int kWgTraceThreads = RD_WG_TRACE_THREADS;
constexpr int kWgTraceWords = kWgTraceThreads / 64;

// THREADS = workgroup size = records per window: 1 024 for the launches that give a ray a workgroup of its own (G-buffer, ReSTIR's
// set-aside lists), 256 inside k_wf_trace, whose 4-wave workgroups trace the stage's literal-class list together before their waves
// go their own ways (kernels_wave.h).
template <int THREADS>
struct WgTraceSharedT {
    int4 jump[THREADS];  // {where the walk is after the records it passes from here, boxes visited, triangles tested, -}
    int prim[THREADS];
    float dist[THREADS], bu[THREADS], bv[THREADS];
    float tmax, baryU, baryV;
    int node, cur, hitPrim, found, state;  // state: 0 window left (node set), 1 hit accepted (cur set, jump table stale)
    unsigned nodes, tris;
};
using WgTraceShared = WgTraceSharedT<kWgTraceThreads>;

constexpr int kWgAccept = 0x40000000;  // jump target: "stopped on record (target & 0xffff), whose triangle is accepted"

template <bool ANY, int THREADS = kWgTraceThreads>
RD_DEV CoopTrace wgTraceWhole(const DScene &s, const NodeRec *nodes, const RaySlab &u, float tLimit, WgTraceSharedT<THREADS> &sh) {
    constexpr int kWgTraceThreads = THREADS;  // (shadows the launch constant: everything below is per window of THREADS records)
    const int tid = int(threadIdx.x);
    const int lane = tid & 63, w = tid >> 6;
    const int end = s.bvhSize;
    if (tid == 0) {
        sh.node = 0;
        sh.tmax = tLimit;
        sh.hitPrim = -1;
        sh.baryU = sh.baryV = 0.f;
        sh.found = 0;
        sh.nodes = sh.tris = 0u;
    }
    __syncthreads();
    for (;;) {
        const int base = sh.node;
        if (base == end || (ANY && sh.found)) break;  // uniform: read after a barrier
        float tm = sh.tmax;
        const int nvalid = (end - base) < kWgTraceThreads ? (end - base) : kWgTraceThreads;
        const bool valid = tid < nvalid;
        float4 lo = make_float4(0.f, 0.f, 0.f, __int_as_float(-1)), hi = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            lo = nodes[base + tid].lo_prim;
            hi = nodes[base + tid].hi_next;
        }
        float t = 0.f;
        bool bh;
        if (u.cls == 0) bh = aabbFast(lo, hi, u, t);
        else bh = aabbIntersect(mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), u, t);
        bh = bh && valid;
        const int prim = __float_as_int(lo.w);
        const bool isLeaf = valid && prim >= 0;
        bool triHit = false;
        float dist = 0.f;
        v2 bary = mk2(0.f, 0.f);
        if (isLeaf && bh && t < tm) {  // speculative: tmax can only shrink, so this covers every leaf the walk can accept
            TriVerts tv = loadTri(s.tris, prim);
            triHit = intersectTriangle(u, tv.a, tv.b, tv.c, bary, dist);
        }
        const int link = __float_as_int(hi.w) - base;  // miss link, relative to the window (always > tid)
        sh.prim[tid] = prim;
        sh.dist[tid] = dist;
        sh.bu[tid] = bary.x;
        sh.bv[tid] = bary.y;
        if (tid == 0) sh.cur = 0;
        for (;;) {
            // One step of the walk from every record: descend (next record), stop on an accepted triangle, or follow the
            // miss link.  Then pointer jumping inside the wave's 64 records (6 doublings, shuffles only), so that every
            // record knows where the walk that enters at it leaves the wave's records, and what it counts on the way.
            const bool D = bh && t < tm;          // would descend, under the current closest distance
            const bool A = triHit && dist < tm;   // a leaf whose triangle the walk would accept
            int to = D ? (A ? (kWgAccept | tid) : tid + 1) : link;
            int cn = 1, ct = (D && isLeaf) ? 1 : 0;
            if (!valid) to = kWgTraceThreads, cn = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const bool inWave = to < nvalid && (to >> 6) == w;
                const int src = inWave ? (to & 63) : lane;
                const int to2 = __shfl(to, src, 64), cn2 = __shfl(cn, src, 64), ct2 = __shfl(ct, src, 64);
                if (inWave) {
                    to = to2;
                    cn += cn2;
                    ct += ct2;
                }
            }
            sh.jump[tid] = make_int4(to, cn, ct, 0);
            __syncthreads();
            if (tid < 64) {  // wave 0 hops from wave to wave (every lane computes the same; lane 0 stores)
                int cur = sh.cur;
                unsigned vn = 0u, vt = 0u;
                int state = 0;
                while (cur < nvalid) {
                    const int4 e = sh.jump[cur];
                    vn += (unsigned)e.y;
                    vt += (unsigned)e.z;
                    cur = e.x;
                    if (cur & kWgAccept) {
                        const int a = cur & 0xffff;
                        if (lane == 0) {
                            if (ANY) {
                                sh.found = 1;
                            } else {
                                sh.hitPrim = sh.prim[a];
                                sh.tmax = sh.dist[a];
                                sh.baryU = sh.bu[a];
                                sh.baryV = sh.bv[a];
                            }
                        }
                        state = 1;
                        cur = a + 1;
                        break;
                    }
                }
                if (lane == 0) {
                    sh.nodes += vn;
                    sh.tris += vt;
                    sh.cur = cur;
                    sh.state = state;
                    if (state == 0) sh.node = base + cur;  // ran off the window, or a miss link that leaves it
                }
            }
            __syncthreads();
            if (sh.state == 0 || ANY) break;  // uniform
            tm = sh.tmax;  // the closer hit prunes the rest of the window: new D / A, new jump table (wave 0 rewrites
                           // sh.state / sh.tmax only after the barrier that follows the table, so everyone has read them)
        }
    }
    CoopTrace o{sh.hitPrim, mk2(sh.baryU, sh.baryV), sh.tmax, sh.found != 0, sh.nodes, sh.tris};
    __syncthreads();  // sh may be reused by the caller's next trace
    return o;
}

}  // namespace rd
