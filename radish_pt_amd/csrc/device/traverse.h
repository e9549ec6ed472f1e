// radish_pt_amd/csrc/device/traverse.h — ray/box, ray/triangle and the two threaded-BVH walks.
//
// Result contract: for every ray the closest-hit record {primId, bary, dist} and the any-hit boolean are those of
// DevScene::intersect / DevScene::testOcclusion (/root/reference/src/scene.h:262-334): same visiting order (the
// six direction-ordered arrays), same strict `<` tie rules, same slab test with its axis-parallel special cases
// (/root/reference/src/bvh.h:91-155) and the same two-sided Möller–Trumbore (/root/reference/src/intersections.h:20-68).
// What differs is mechanical: one 32-byte record per step instead of node → box indirection, 1/dir and the
// ray's slab-test class computed once per ray instead of once per node.
#pragma once
#include "layouts.h"

namespace rd {

struct Ray {  // src/sceneStructs.h:13-19
    v3 o, d;
};

RD_DEV Ray makeOffsetedRay(v3 ori, v3 dir) { return {ori + dir * 1e-5f, dir}; }  // intersections.h:16-18

RD_DEV int getMTBVHId(v3 dir) {  // scene.h:114-129 (called with -ray.direction)
    float ax = fabs_(dir.x), ay = fabs_(dir.y), az = fabs_(dir.z);
    if (ax > ay) {
        if (ax > az) return dir.x > 0 ? 0 : 1;
        return dir.z > 0 ? 4 : 5;
    }
    if (ay > az) return dir.y > 0 ? 2 : 3;
    return dir.z > 0 ? 4 : 5;
}

// Per-ray constants of AABB::intersect.
struct RaySlab {
    v3 o, d, inv;
    int cls;  // 0: no special case applies; 1/2/3: |d.x|/|d.y|/|d.z| > 1-Eps; 4: some |d.k| < Eps;
              // 5: non-finite or astronomically far ray (literal path, so NaN/inf propagate as in the reference)
};
RD_DEV RaySlab makeRaySlab(const Ray &r) {
    const float Eps = 1e-6f;
    RaySlab s;
    s.o = r.o;
    s.d = r.d;
    s.inv = rdiv(1.f, r.d);
    float ax = fabs_(r.d.x), ay = fabs_(r.d.y), az = fabs_(r.d.z);
    if (ax > 1.f - Eps) s.cls = 1;
    else if (ay > 1.f - Eps) s.cls = 2;
    else if (az > 1.f - Eps) s.cls = 3;
    else if (ax < Eps || ay < Eps || az < Eps) s.cls = 4;
    else s.cls = 0;
    // `!(x < bound)` is also true for NaN
    const float Far = 1e30f;
    if (!(fabs_(r.o.x) < Far) || !(fabs_(r.o.y) < Far) || !(fabs_(r.o.z) < Far) || !(ax <= 1.f) || !(ay <= 1.f) ||
        !(az <= 1.f))
        s.cls = (s.cls == 0) ? 5 : s.cls;
    return s;
}

// Is a ray of this class worth setting aside for a workgroup-per-ray launch (k_trace_wg_list, k_gbuffer_literal)?  Classes 1-4
// wander through the tree for thousands of box steps.  Class 5 does not: a non-finite ray — ReSTIR's zero-length shadow segment of
// an empty reservoir gives a NaN direction, 5 % of a frame's segments — fails the root's box test at once and stays in the walker.
RD_DEV bool raySetAside(int cls) { return cls >= 1 && cls <= 4; }

RD_DEV bool between(float x, float mn, float mx) { return x >= mn && x <= mx; }  // mathUtil.h:34-36
RD_DEV bool distMinMax(float a1, float a2, float b1, float b2, float &tMin) {     // bvh.h:72-78
    tMin = c_fminf(a1, a2);
    float tMax = c_fmaxf(b1, b2);
    return (tMax >= 0.f && tMax >= tMin);
}
RD_DEV bool distMaxMin(float a1, float a2, float b1, float b2, float &tMin) {  // bvh.h:80-86
    tMin = c_fmaxf(a1, a2);
    float tMax = c_fminf(b1, b2);
    return (tMax >= 0.f && tMax >= tMin);
}

// AABB::intersect (bvh.h:91-155), literal.
RD_DEV bool aabbIntersect(v3 pMin, v3 pMax, const RaySlab &r, float &tMin) {
    const float Eps = 1e-6f;
    if (r.cls >= 1 && r.cls <= 3) {
        if (r.cls == 1) {
            if (between(r.o.y, pMin.y, pMax.y) && between(r.o.z, pMin.z, pMax.z)) {
                float t1 = (pMin.x - r.o.x) * r.inv.x, t2 = (pMax.x - r.o.x) * r.inv.x;
                return distMinMax(t1, t2, t1, t2, tMin);
            }
            return false;
        } else if (r.cls == 2) {
            if (between(r.o.z, pMin.z, pMax.z) && between(r.o.x, pMin.x, pMax.x)) {
                float t1 = (pMin.y - r.o.y) * r.inv.y, t2 = (pMax.y - r.o.y) * r.inv.y;
                return distMinMax(t1, t2, t1, t2, tMin);
            }
            return false;
        } else {
            if (between(r.o.x, pMin.x, pMax.x) && between(r.o.y, pMin.y, pMax.y)) {
                float t1 = (pMin.z - r.o.z) * r.inv.z, t2 = (pMax.z - r.o.z) * r.inv.z;
                return distMinMax(t1, t2, t1, t2, tMin);
            }
            return false;
        }
    }
    v3 t1 = (pMin - r.o) * r.inv;
    v3 t2 = (pMax - r.o) * r.inv;
    v3 tNear = gmin(t1, t2);
    v3 tFar = gmax(t1, t2);
    v3 tDist = tFar - tNear;
    float yz = tFar.z - tNear.y;
    float zx = tFar.x - tNear.z;
    float xy = tFar.y - tNear.x;
    if (r.cls == 4) {
        if (fabs_(r.d.x) < Eps && tDist.y + tDist.z > yz) return distMaxMin(tNear.y, tNear.z, tFar.y, tFar.z, tMin);
        if (fabs_(r.d.y) < Eps && tDist.z + tDist.x > zx) return distMaxMin(tNear.z, tNear.x, tFar.z, tFar.x, tMin);
        if (fabs_(r.d.z) < Eps && tDist.x + tDist.y > xy) return distMaxMin(tNear.x, tNear.y, tFar.x, tFar.y, tMin);
    }
    if (tDist.y + tDist.z > yz && tDist.z + tDist.x > zx && tDist.x + tDist.y > xy) {
        return distMaxMin(c_fmaxf(tNear.x, tNear.y), tNear.z, c_fminf(tFar.x, tFar.y), tFar.z, tMin);
    }
    return false;
}

// intersectTriangle (intersections.h:20-68), straight-line: every lane that tests computes the whole expression and
// the early-outs become one conjunction.  Same operations in the same order, so the accepted hits and their
// {bary, dist} bits are those of the early-out form (a rejected candidate's leftovers are never read).
RD_DEV bool intersectTriangle(const RaySlab &ray, v3 v0, v3 v1, v3 vc, v2 &bary, float &dist) {
    v3 e01 = v1 - v0;
    v3 e02 = vc - v0;
    v3 pvec = cross(ray.d, e02);
    float det = dot(e01, pvec);
    bool ok = !(fabs_(det) < 1.1920928955078125e-7f);  // FLT_EPSILON
    v3 v0ToOri = ray.o - v0;
    bool neg = det < 0.f;
    det = neg ? -det : det;
    v0ToOri = neg ? -v0ToOri : v0ToOri;
    float bx = dot(v0ToOri, pvec);
    ok = ok && !(bx < 0.f || bx > det);
    v3 qvec = cross(v0ToOri, e01);
    float by = dot(ray.d, qvec);
    ok = ok && !(by < 0.f || bx + by > det);
    float invDet = 1.f / det;
    bary.x = bx * invDet;
    bary.y = by * invDet;
    dist = dot(e02, qvec) * invDet;
    return ok && dist > 0.f;
}

struct TriVerts {
    v3 a, b, c;
    int matId;
};
RD_DEV TriVerts loadTri(const TriRec *tris, int prim) {
    const TriRec *t = tris + prim;
    float4 A = t->a, B = t->b, C = t->c;
    TriVerts r;
    r.a = mk3(A.x, A.y, A.z);
    r.b = mk3(A.w, B.x, B.y);
    r.c = mk3(B.z, B.w, C.x);
    r.matId = __float_as_int(C.y);
    return r;
}

struct WalkStats {
    unsigned nodes, tris;
};

struct HitRec {
    int prim;
    v2 bary;
    float dist;
};

// Slab test for the common ray class (cls 0: finite ray, no |d.k| > 1-Eps, no |d.k| < Eps): the last branch of
// AABB::intersect (bvh.h:125-154) with nothing else reachable.  All t values are finite here (|inv| <= 1e6 and
// rdh_scene_upload bounds the scene), so v_min/v_max equal glm::min/max and C fminf/fmaxf up to the sign of a zero,
// which no comparison below can see.
RD_DEV bool aabbFast(float4 lo, float4 hi, const RaySlab &r, float &tMin) {
    float t1x = (lo.x - r.o.x) * r.inv.x, t1y = (lo.y - r.o.y) * r.inv.y, t1z = (lo.z - r.o.z) * r.inv.z;
    float t2x = (hi.x - r.o.x) * r.inv.x, t2y = (hi.y - r.o.y) * r.inv.y, t2z = (hi.z - r.o.z) * r.inv.z;
    float nx = __builtin_fminf(t1x, t2x), ny = __builtin_fminf(t1y, t2y), nz = __builtin_fminf(t1z, t2z);
    float fx = __builtin_fmaxf(t1x, t2x), fy = __builtin_fmaxf(t1y, t2y), fz = __builtin_fmaxf(t1z, t2z);
    float dx = fx - nx, dy = fy - ny, dz = fz - nz;
    float yz = fz - ny, zx = fx - nz, xy = fy - nx;
    bool overlap = (dy + dz > yz) & (dz + dx > zx) & (dx + dy > xy);
    tMin = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
    float tMax = __builtin_fminf(__builtin_fminf(fx, fy), fz);
    return overlap & (tMax >= 0.f) & (tMax >= tMin);
}

// aabbFast with the twelve subtract / multiply of the slab test as four packed pairs (v_pk_add_f32 / v_pk_mul_f32 process
// two binary32 lanes per instruction, each IEEE-exact, so the bits are those of aabbFast); the ray's origin and reciprocal
// direction are kept as register pairs.  The .w halves of the records (int bits) ride along and are never read.
typedef float f2v __attribute__((ext_vector_type(2)));
struct RaySlabPk {
    f2v oxy, ozw, ixy, izw;  // origin.xy, {origin.z, 0}, inv.xy, {inv.z, 0}
};
RD_DEV RaySlabPk packSlab(const RaySlab &r) {
    RaySlabPk p;
    p.oxy = f2v{r.o.x, r.o.y};
    p.ozw = f2v{r.o.z, 0.f};
    p.ixy = f2v{r.inv.x, r.inv.y};
    p.izw = f2v{r.inv.z, 0.f};
    return p;
}
RD_DEV bool aabbFastPk(float4 lo, float4 hi, const RaySlabPk &r, float &tMin) {
    const f2v t1xy = (f2v{lo.x, lo.y} - r.oxy) * r.ixy, t1zw = (f2v{lo.z, lo.w} - r.ozw) * r.izw;
    const f2v t2xy = (f2v{hi.x, hi.y} - r.oxy) * r.ixy, t2zw = (f2v{hi.z, hi.w} - r.ozw) * r.izw;
    float nx = __builtin_fminf(t1xy.x, t2xy.x), ny = __builtin_fminf(t1xy.y, t2xy.y), nz = __builtin_fminf(t1zw.x, t2zw.x);
    float fx = __builtin_fmaxf(t1xy.x, t2xy.x), fy = __builtin_fmaxf(t1xy.y, t2xy.y), fz = __builtin_fmaxf(t1zw.x, t2zw.x);
    float dx = fx - nx, dy = fy - ny, dz = fz - nz;
    float yz = fz - ny, zx = fx - nz, xy = fy - nx;
    bool overlap = (dy + dz > yz) & (dz + dx > zx) & (dx + dy > xy);
    tMin = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
    float tMax = __builtin_fminf(__builtin_fminf(fx, fy), fz);
    return overlap & (tMax >= 0.f) & (tMax >= tMin);
}
// ---- pair-shared node fetch -----------------------------------------------------------------------------------------------
// The vector L1 prices a divergent gather per LANE REQUEST: a 64-lane dwordx4 load of 64 different lines costs ~27 ns of the
// CU's L1 whatever its width, and a box step needs two of them per lane (the two halves of one 32-byte record) — 54 ns per
// wave-step, the ceiling of every walker here (304 G box steps/s; scripts/micro/gather_modes.hip, mode 0).  Two lanes reading
// the two halves of ONE record in the same instruction cost one request (mode 2: 16-19 ns for 32 records).  So lanes work in
// pairs (2k, 2k+1): instruction 1 fetches the EVEN lane's record (even lane its low half, odd lane its high half), instruction 2
// the ODD lane's record, and each lane takes the half it is missing from its partner with a DPP quad-permute (VALU, no LDS):
// two requests per PAIR and step instead of four.  Must be called by all 64 lanes (uniform control flow).
RD_DEV int dppSwapPairI(int v) { return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false); }  // quad_perm [1,0,3,2]
RD_DEV float dppSwapPairF(float v) { return __int_as_float(dppSwapPairI(__float_as_int(v))); }
// Returns the record's two halves as (mine, other): `mine` is the half this lane loaded itself (even lane: pMin | primitiveId,
// odd lane: pMax | nextNodeIfMiss), `other` the half its partner loaded for it.  The slab test is symmetric in the two corners
// (it only takes componentwise min / max of the two t vectors), so callers pass them in this order; `prim` and `next` are the
// .w fields, sorted out here.  (DPP bank masks select groups of four consecutive lanes, not lane parity, hence the selects.)
RD_DEV void fetchNodePaired(const char *base, unsigned ofs, bool walking, bool odd, float4 &mine, float4 &other, int &prim, int &next) {
    const unsigned pofs = (unsigned)dppSwapPairI((int)ofs);
    const bool pw = dppSwapPairI(walking ? 1 : 0) != 0;
    const unsigned half = odd ? 16u : 0u;
    const unsigned a1 = (odd ? pofs : ofs) + half;  // the even lane's record
    const unsigned a2 = (odd ? ofs : pofs) + half;  // the odd lane's record
    const bool need1 = odd ? pw : walking, need2 = odd ? walking : pw;
    float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f), r2 = r1;
    if (need1) r1 = *reinterpret_cast<const float4 *>(base + a1);
    if (need2) r2 = *reinterpret_cast<const float4 *>(base + a2);
    mine = odd ? r2 : r1;
    const float4 send = odd ? r1 : r2;
    other = make_float4(dppSwapPairF(send.x), dppSwapPairF(send.y), dppSwapPairF(send.z), dppSwapPairF(send.w));
    prim = __float_as_int(odd ? other.w : mine.w);
    next = __float_as_int(odd ? mine.w : other.w);
}

RD_DEV unsigned long long ballotb(bool p) { return __builtin_amdgcn_ballot_w64(p); }  // no bool -> int -> compare round trip

// Out-of-line wrapper for the literal test: only rays with an axis-parallel, tiny or non-finite direction component
// come here, and inlining it makes the compiler fuse it with the fast path of the hot loop.  Everything by value (an
// address-taken RaySlab would be forced into scratch memory).  Returns {hit ? 1 : 0, tMin}.
__device__ __attribute__((noinline)) float2 aabbSlow(float4 lo, float4 hi, float ox, float oy, float oz, float dx, float dy,
                                                     float dz, float ix, float iy, float iz, int cls) {
    RaySlab r;
    r.o = mk3(ox, oy, oz);
    r.d = mk3(dx, dy, dz);
    r.inv = mk3(ix, iy, iz);
    r.cls = cls;
    float t = 0.f;
    bool h = aabbIntersect(mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), r, t);
    return make_float2(h ? 1.f : 0.f, t);
}

RD_DEV bool boxTest(float4 lo, float4 hi, const RaySlab &rs, float &t) {
    if (rs.cls == 0) return aabbFast(lo, hi, rs, t);
    float2 r = aabbSlow(lo, hi, rs.o.x, rs.o.y, rs.o.z, rs.d.x, rs.d.y, rs.d.z, rs.inv.x, rs.inv.y, rs.inv.z, rs.cls);
    t = r.y;
    return r.x != 0.f;
}

// How many of the still-running lanes must be parked on a leaf before the wave leaves the box loop to run the
// triangle tests together: parked * kLeafDen >= running * kLeafNum.
#ifndef RD_LEAF_NUM
#define RD_LEAF_NUM 1
#endif
#ifndef RD_LEAF_DEN
#define RD_LEAF_DEN 4
#endif

// ---- wave-cooperative walk of ONE ray ------------------------------------------------------------------------------
// A lone ray pays ~370 ns per box step (dependent gather + ~65 instructions, measured), and node visits per ray are
// heavy-tailed, so the last few rays of a launch decide when it ends.  When only a few lanes of a wave are still
// walking, the whole wave works for one of them: lane j tests node `base + j` of the 64 consecutive records that
// follow the ray's position (threaded order = memory order, so this is one coalesced 2 KB read), then the walk through
// that window is resolved with scalar bit operations — runs of inner-node descents are skipped with a find-first-set,
// a miss follows the record's miss link if it lands inside the window.  Every decision is the per-lane walk's
// (same box test on the same record against the same tmax), only evaluated ahead of time; nodes that the sequential
// walk would not visit are tested but not counted.  Returns at the first leaf hit (the triangle test belongs to the
// owner lane and may change tmax), at the end of the walk, or after `maxWindows` windows.
// ALL 64 lanes must call this with identical arguments (exec = full wave).
struct CoopResult {
    int node, pending;
    unsigned visited;
};
RD_DEV int readlaneI(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
RD_DEV float readlaneF(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
RD_DEV CoopResult coopWalk(const NodeRec *nodes, int node, int end, const RaySlab &u, float tmax, int maxWindows) {
    const int lane = int(threadIdx.x & 63u);
    CoopResult r{node, -1, 0u};
    for (int w = 0; w < maxWindows && r.node != end && r.pending < 0; ++w) {
        const int base = r.node;
        const int nvalid = (end - base) < 64 ? (end - base) : 64;
        const bool valid = lane < nvalid;
        float4 lo = make_float4(0.f, 0.f, 0.f, __int_as_float(-1)), hi = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            lo = nodes[base + lane].lo_prim;
            hi = nodes[base + lane].hi_next;
        }
        float t = 0.f;
        bool bh;
        if (u.cls == 0) bh = aabbFast(lo, hi, u, t);  // the ray is wave-uniform here, so this is a scalar branch:
        else bh = aabbIntersect(mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), u, t);  // the literal test costs nothing extra
        const unsigned long long D = __ballot(valid && bh && t < tmax);             // would descend
        const unsigned long long LEAF = __ballot(valid && __float_as_int(lo.w) >= 0);
        const unsigned long long run = D & ~LEAF;                                  // inner-node descents
        int cur = 0;
        for (;;) {
            unsigned long long rest = ~(run >> cur);
            int k = rest ? (__ffsll((long long)rest) - 1) : 64;
            r.visited += (unsigned)k;
            cur += k;
            if (cur >= nvalid) {  // walked off the window (or to the end of the array)
                r.node = base + cur;
                break;
            }
            r.visited += 1u;
            if ((D >> cur) & 1ull) {  // leaf whose box is hit: park for the triangle test
                r.pending = readlaneI(__float_as_int(lo.w), cur);
                r.node = base + cur + 1;
                break;
            }
            int nxt = readlaneI(__float_as_int(hi.w), cur);  // miss link (always > base + cur)
            if (nxt - base < nvalid) {
                cur = nxt - base;
            } else {
                r.node = nxt;
                break;
            }
        }
    }
    return r;
}

RD_DEV unsigned long long laneMaskLt() { return (1ull << (threadIdx.x & 63u)) - 1ull; }  // lanes below this one

// Broadcast lane L's ray to the whole wave (SGPRs).
RD_DEV RaySlab readlaneRay(const RaySlab &rs, int L) {
    RaySlab u;
    u.o = mk3(readlaneF(rs.o.x, L), readlaneF(rs.o.y, L), readlaneF(rs.o.z, L));
    u.d = mk3(readlaneF(rs.d.x, L), readlaneF(rs.d.y, L), readlaneF(rs.d.z, L));
    u.inv = mk3(readlaneF(rs.inv.x, L), readlaneF(rs.inv.y, L), readlaneF(rs.inv.z, L));
    u.cls = readlaneI(rs.cls, L);
    return u;
}
RD_DEV const NodeRec *readlanePtr(const NodeRec *p, int L) {
    unsigned long long v = (unsigned long long)p;
    unsigned lo = (unsigned)readlaneI((int)(unsigned)v, L), hi = (unsigned)readlaneI((int)(unsigned)(v >> 32), L);
    return (const NodeRec *)(((unsigned long long)hi << 32) | lo);
}

// A whole trace of ONE ray by the whole wave: coopWalk to the next leaf hit, triangle test, repeat.  This is how rays
// of the literal classes (a direction component that is axis-parallel, tiny or non-finite) are traced: the
// reference's box test ignores one slab for them (bvh.h:138-148), so they descend into nearly every box — thousands of
// steps, at ~0.8 us each through the per-lane literal path, enough for ONE such ray to decide when a launch ends
// (measured: an 8 ms G-buffer pass on the teapots scene, of which ~5 ms was one ray).  Cooperatively the literal test is
// a scalar branch, 64 boxes are tested per read, and runs of descents are skipped in one find-first-set.
// Every lane computes the (uniform) triangle test; all lanes return the same record.
struct CoopTrace {
    int hitPrim;
    v2 bary;
    float tmax;
    bool found;
    unsigned nodes, tris;
};
template <bool ANY>
RD_DEV CoopTrace coopTraceWhole(const DScene &s, const NodeRec *nodes, const RaySlab &u, float tLimit) {
    // One window = the 64 records that follow the walk's position.  Every lane tests its box, and the lanes whose record
    // is a leaf with a hit box also test their triangle right away (speculatively: under the current tmax, which can only
    // shrink, so the set covers every leaf the walk can still accept).  The walk through the window is then resolved with
    // scalar bit operations: runs of inner-node descents are skipped by a find-first-set, a visited leaf takes its
    // triangle result from the lane that computed it, an accepted hit re-evaluates the `boundDist < closestDist` mask of
    // the rest of the window, a miss follows its link.  Literal-class rays visit nearly every node (one slab of their box
    // test is ignored, bvh.h:138-148) and every second node of the tree is a leaf, so leaving the window for each leaf —
    // a dependent triangle fetch per leaf, as the per-lane walk does — cost ~0.15 us x 20 000 visits = 3 ms for ONE such
    // ray on the teapots scene; a whole window per fetch brings that down several-fold.  Decisions and counters are those of
    // the sequential walk: speculative tests of boxes or triangles the walk does not reach are not counted.
    CoopTrace o{-1, mk2(0.f, 0.f), tLimit, false, 0u, 0u};
    const int lane = int(threadIdx.x & 63u);
    const int end = s.bvhSize;
    int node = 0;
    // the next window's records are requested before the current one is resolved (the walk usually runs off the end of
    // its window into the next 64 records).  (Fetching and testing two windows per round trip was measured and rejected:
    // 1.79 ms against 1.43 ms for the G-buffer pass below — the second window is often not reached.)
    int preBase = -1;
    float4 preLo = make_float4(0.f, 0.f, 0.f, 0.f), preHi = preLo;
    while (node != end) {
        const int base = node;
        const int nvalid = (end - base) < 64 ? (end - base) : 64;
        const bool valid = lane < nvalid;
        float4 lo = make_float4(0.f, 0.f, 0.f, __int_as_float(-1)), hi = make_float4(0.f, 0.f, 0.f, 0.f);
        if (base == preBase) {  // wave-uniform
            lo = preLo;
            hi = preHi;
        } else if (valid) {
            lo = nodes[base + lane].lo_prim;
            hi = nodes[base + lane].hi_next;
        }
        preBase = base + 64;
        if (preBase + lane < end) {
            preLo = nodes[preBase + lane].lo_prim;
            preHi = nodes[preBase + lane].hi_next;
        } else {
            preLo = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            preHi = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float t = 0.f;
        bool bh;
        if (u.cls == 0) bh = aabbFast(lo, hi, u, t);  // wave-uniform ray: a scalar branch
        else bh = aabbIntersect(mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), u, t);
        bh = bh && valid;
        const int prim = __float_as_int(lo.w);
        const bool isLeaf = valid && prim >= 0;
        bool triHit = false;
        float dist = 0.f;
        v2 bary = mk2(0.f, 0.f);
        if (isLeaf && bh && t < o.tmax) {
            TriVerts tv = loadTri(s.tris, prim);
            triHit = intersectTriangle(u, tv.a, tv.b, tv.c, bary, dist);
        }
        const unsigned long long LEAF = __ballot(isLeaf);
        unsigned long long D = __ballot(bh && t < o.tmax);         // would descend, under the current closest distance
        unsigned long long ACC = __ballot(triHit && dist < o.tmax);  // leaves whose triangle the walk would accept
        int cur = 0;
        for (;;) {
            if (cur >= nvalid) {  // walked off the window (or to the end of the array)
                node = base + cur;
                break;
            }
            // skip a run of descents — inner nodes, and leaves whose triangle changes nothing — in one find-first-set
            const unsigned long long rest = ~((D & ~ACC) >> cur);
            const int k = rest ? (__ffsll((long long)rest) - 1) : 64;
            if (k > 0) {
                const unsigned long long range = (k >= 64 ? ~0ull : ((1ull << k) - 1ull)) << cur;
                o.nodes += (unsigned)k;
                o.tris += (unsigned)__popcll(LEAF & D & range);
                cur += k;
                if (cur >= nvalid) {
                    node = base + cur;
                    break;
                }
            }
            o.nodes += 1u;
            if ((D >> cur) & 1ull) {  // a leaf whose triangle (tested by lane `cur`) is accepted
                o.tris += 1u;
                if (ANY) {
                    o.found = true;
                    return o;
                }
                o.hitPrim = readlaneI(prim, cur);
                o.tmax = readlaneF(dist, cur);
                o.bary = mk2(readlaneF(bary.x, cur), readlaneF(bary.y, cur));
                D = __ballot(bh && t < o.tmax);  // the closer hit prunes the rest of the window
                ACC = __ballot(triHit && dist < o.tmax);
                cur += 1;
            } else {
                const int nxt = readlaneI(__float_as_int(hi.w), cur);  // miss link (always > base + cur)
                if (nxt - base < nvalid) {
                    cur = nxt - base;
                } else {
                    node = nxt;
                    break;
                }
            }
        }
    }
    return o;
}

#ifndef RD_COOP_MAX
#define RD_COOP_MAX 1      // cooperate once at most this many lanes of the wave are still walking (2-3: measured slower)
#endif
#ifndef RD_COOP_WINDOWS
#define RD_COOP_WINDOWS 8  // windows per lane per turn
#endif

// One threaded-BVH walk per lane (DevScene::intersect, scene.h:262-301, and DevScene::testOcclusion's loop,
// :316-333).  Each lane performs exactly the reference's sequence — box test, on a leaf hit the triangle test, strict
// `<` updates, node++ / nextNodeIfMiss — but the wave runs it "while-while": lanes step through boxes until enough of
// them are parked on a leaf, then the parked lanes test their triangles together.  Only ~3 % of visits reach a
// triangle, so testing as soon as ONE lane needs it (the if-if form) makes every step pay for a triangle test.
// ANY = true: any-hit with a fixed distance bound; returns true on the first accepted triangle.
// `active` = false lets a lane take part in the wave-level scheduling (and in coopWalk) without tracing anything;
// call sites that can, call this from uniform control flow with a flag instead of from inside a branch.
template <bool COUNT, bool ANY>
RD_DEV bool walkRay(const DScene &s, const Ray &ray, float tLimit, HitRec &h, WalkStats &ws, bool active = true) {
    RaySlab rs = makeRaySlab(ray);
    const NodeRec *nodes = s.nodes[getMTBVHId(-ray.d)];
    h.prim = -1;
    h.bary = mk2(0.f, 0.f);
    h.dist = tLimit;
    int node = 0;
    const int end = s.bvhSize;
    int pending = -1;
    bool alive = active && node != end;
    bool found = false;
    const bool fullWave = __ballot(true) == ~0ull;  // the cooperative routines need every lane of the wave
    if (fullWave) {
        // literal-class rays: traced whole, one after the other, by the whole wave (see coopTraceWhole)
        unsigned long long lit = __ballot(alive && rs.cls != 0);
        while (lit) {
            const int L = __ffsll((long long)lit) - 1;
            lit &= lit - 1ull;
            CoopTrace ct = coopTraceWhole<ANY>(s, readlanePtr(nodes, L), readlaneRay(rs, L), readlaneF(tLimit, L));
            if (int(threadIdx.x & 63u) == L) {
                h.prim = ct.hitPrim;
                h.bary = ct.bary;
                h.dist = ct.tmax;
                found = ct.found;
                if (COUNT) {
                    ws.nodes += ct.nodes;
                    ws.tris += ct.tris;
                }
                alive = false;
            }
        }
    }
    for (;;) {
        for (;;) {
            bool walking = alive && pending < 0;
            unsigned long long wm = __ballot(walking);
            if (wm == 0ull) break;
            unsigned long long pm = __ballot(alive && pending >= 0);
            if (__popcll(pm) * RD_LEAF_DEN >= (__popcll(wm) + __popcll(pm)) * RD_LEAF_NUM && pm != 0ull) break;
            if (fullWave && __popcll(wm) <= RD_COOP_MAX) {
                // few walkers left: the wave serves them one at a time
                unsigned long long todo = wm;
                while (todo) {
                    const int L = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    {
                        CoopResult cr = coopWalk(readlanePtr(nodes, L), readlaneI(node, L), end, readlaneRay(rs, L),
                                                 readlaneF(h.dist, L), RD_COOP_WINDOWS);
                        if (int(threadIdx.x & 63u) == L) {
                            node = cr.node;
                            pending = cr.pending;
                            if (COUNT) ws.nodes += cr.visited;
                            alive = (node != end) || pending >= 0;
                        }
                    }
                }
                continue;
            }
            if (walking) {
                float4 lo = nodes[node].lo_prim;
                float4 hi = nodes[node].hi_next;
                float boundDist;
                if (COUNT) ws.nodes++;
                bool boundHit = boxTest(lo, hi, rs, boundDist);
                if (boundHit && boundDist < h.dist) {
                    pending = __float_as_int(lo.w);  // -1 for an inner node
                    node++;
                } else {
                    node = __float_as_int(hi.w);
                }
                alive = (node != end) || pending >= 0;
            }
        }
        if (__ballot(alive) == 0ull) break;
        if (alive && pending >= 0) {
            TriVerts t = loadTri(s.tris, pending);
            float dist;
            v2 bary;
            if (COUNT) ws.tris++;
            bool hit = intersectTriangle(rs, t.a, t.b, t.c, bary, dist);
            if (hit && dist < h.dist) {
                if (ANY) {
                    found = true;
                    node = end;
                } else {
                    h.prim = pending;
                    h.dist = dist;
                    h.bary = bary;
                }
            }
            pending = -1;
            alive = node != end;
        }
    }
    return ANY ? found : (h.prim != -1);
}

// DevScene::intersect (scene.h:262-301), geometry part.  `prim` = NullPrimitive (-1) on a miss.
template <bool COUNT>
RD_DEV HitRec traceClosest(const DScene &s, const Ray &ray, WalkStats &ws, bool active = true) {
    HitRec h;
    walkRay<COUNT, false>(s, ray, 3.402823466e+38f /* FLT_MAX */, h, ws, active);
    return h;
}

// DevScene::testOcclusion (scene.h:303-334)
template <bool COUNT>
RD_DEV bool traceOccluded(const DScene &s, v3 x, v3 y, WalkStats &ws, bool active = true) {
    const float eps = 1e-4f;
    v3 dir = y - x;
    float dist = length(dir);
    dir = dir / dist;
    dist -= eps;
    Ray ray = makeOffsetedRay(x, dir);
    HitRec h;
    return walkRay<COUNT, true>(s, ray, dist, h, ws, active);
}

// ---- walks over SIBLING PAIRS (DScene::pairs, layouts.h) -----------------------------------------------------------------------
// DevScene::intersect visits the two children of an inner node one after the other — the near one (by the ray's ordering) when it
// enters the node, the far one when the near subtree is done — and tests each box against the closest distance of THAT moment.
// Here a lane that enters a node fetches both children in one round trip (one 64-byte record) and tests both boxes at once:
//   * the near child is "visited" now: a hit leaf is parked for its triangle test, a hit inner node is entered next;
//   * the far child is visited LATER, so its test is only PROVISIONAL: tmax can but shrink until then, hence a box that fails now
//     (missed, or boundDist >= tmax) fails then as well and nothing need be kept for it, while a box that passes is pushed as
//     {w, boundDist} and re-checked against the tmax of the moment it is popped — exactly the reference's `boundDist < closestDist`
//     at the reference's time.  Triangles are therefore tested in the reference's order against the reference's distances: same
//     hits (same strict-< ties), and with COUNT the failed far children are pushed too (distance +inf) and every far child is
//     counted when it is popped, i.e. when the sequential walk reaches it — so an any-hit walk that ends early counts what the
//     reference counts.
// What it buys: half the dependent round trips per visit, and stacks a few entries deep (only far children that are hit) where a
// one-node-per-step walk over a shared tree needs one entry per level of these 46-102-level trees (measured first: profiles/r03_g_*).
// Per-lane state: `cur` = the pair to enter next (-1: none) and the stack {sp, lo}: entries lo .. sp - 1 — the most recent, at
// most kPairLds of them — live in LDS (`stk`: this wave's [slot][lane] block of int2, a ring indexed by entry number mod kPairLds),
// entries 0 .. lo - 1 have been moved to this wave's strip of global memory (`ovf`: [entry][lane]).  A push that finds the ring
// full moves the OLDEST resident entry out (one LDS read + one global store, nothing waits for it); a pop that finds the ring
// empty takes the entry back from global memory.  So a walk that stays deep for a long time keeps working at the top of its
// stack in LDS, and global traffic is paid for net crossings of the boundary only (the first version put everything above row
// kPairLds in global memory: every step of a deep walk went there, and 6 / 8 / 12 rows measured 9.39 / 8.95 / 8.77 ms per frame).
#ifndef RD_PAIR_LDS
#define RD_PAIR_LDS 8  // a power of two
#endif
constexpr int kPairLds = RD_PAIR_LDS;
static_assert((kPairLds & (kPairLds - 1)) == 0, "RD_PAIR_LDS");
constexpr int kPairNone = -1, kPairFresh = -2;  // `cur`: nothing to enter / a literal-class ray that has not started (traced whole)
struct PairStack {
    int sp, lo;  // entries on the stack; how many of them (the oldest) are in global memory
};
// (Explicit address spaces: left generic, the compiler folds "ring or global" into ONE flat load through a selected pointer, which
// waits on both memory counters; these are a ds_read_b64 / ds_write_b64 and, in the rare branch, a global access.)
typedef int stackWord2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) stackWord2 LdsWord2;
typedef __attribute__((address_space(1))) stackWord2 GlobalWord2;
RD_DEV void pairPush(int2 *stk, int lane, int2 *__restrict__ ovf, PairStack &st, int w, float d) {
    LdsWord2 *ring = (LdsWord2 *)stk;
    if (st.sp - st.lo == kPairLds) {  // rare: the oldest resident entry moves out
        ((GlobalWord2 *)ovf)[(size_t)st.lo * 64 + lane] = ring[(st.lo & (kPairLds - 1)) * 64 + lane];
        st.lo++;
    }
    ring[(st.sp & (kPairLds - 1)) * 64 + lane] = stackWord2{w, __float_as_int(d)};
    st.sp++;
}
RD_DEV int2 pairPop(const int2 *stk, int lane, const int2 *__restrict__ ovf, PairStack &st) {
    st.sp--;
    stackWord2 e = ((const LdsWord2 *)stk)[(st.sp & (kPairLds - 1)) * 64 + lane];
    if (__builtin_expect(st.sp < st.lo, 0)) {  // rare: the entry has been moved out
        st.lo = st.sp;
        e = ((const GlobalWord2 *)ovf)[(size_t)st.sp * 64 + lane];
    }
    return make_int2(e.x, e.y);
}
// The root: a single box, from the kernel arguments.  Class-0 rays only.  Returns true when the walk goes on.
template <bool COUNT>
RD_DEV void pairStart(const DScene &s, const RaySlab &rs, float tmax, int &cur, int &pending, WalkStats &ws) {
    cur = kPairNone;
    pending = -1;
    if (s.bvhSize == 0) return;
    if (COUNT) ws.nodes++;
    float d;
    if (aabbFast(s.rootLo, s.rootHi, rs, d) && d < tmax) {
        const int w = __float_as_int(s.rootLo.w);
        if (w >= 0) pending = w;
        else cur = ~w;
    }
}
// One iteration of a busy lane (pending < 0, and cur >= 0 or entries stacked): a lane with nothing to enter pops ONE entry — the far
// child the sequential walk reaches next — and drops it unless its box still beats the closest distance; a lane with a pair to
// enter (also one that has just popped an inner node) does pairStep.  `busy` goes false when the lane parks on a leaf or its walk
// is over.  (A first version popped in a wave-level loop until every lane had something to enter: the loop-carried lane
// predicates cost more instructions than the idle slots of this form.)
template <bool COUNT>
RD_DEV void pairPopOne(const int2 *stk, int lane, const int2 *__restrict__ ovf, float tmax, bool &busy, int &cur, PairStack &sp, int &pending,
                       WalkStats &ws) {
    if (busy && cur < 0) {
        if (sp.sp == 0) {
            busy = false;
        } else {
            const int2 e = pairPop(stk, lane, ovf, sp);
            if (COUNT) ws.nodes++;
            if (__int_as_float(e.y) < tmax) {
                if (e.x >= 0) {
                    pending = e.x;
                    busy = false;
                } else {
                    cur = ~e.x;
                }
            }
        }
    }
}
// One pair step of a lane with cur >= 0.
template <bool COUNT>
RD_DEV void pairStep(const PairRec *__restrict__ pairs, int2 *stk, int lane, int2 *__restrict__ ovf, const RaySlab &rs, float tmax, int ord,
                     int &cur, PairStack &sp, int &pending, WalkStats &ws) {
    const char *pb = reinterpret_cast<const char *>(pairs);
    const unsigned ofs = (unsigned)cur << 6;
    const float4 a0 = *reinterpret_cast<const float4 *>(pb + ofs);
    const float4 a1 = *reinterpret_cast<const float4 *>(pb + ofs + 16u);
    const float4 b0 = *reinterpret_cast<const float4 *>(pb + ofs + 32u);
    const float4 b1 = *reinterpret_cast<const float4 *>(pb + ofs + 48u);
    float d0, d1;
    const bool h0 = aabbFast(a0, a1, rs, d0) && d0 < tmax;
    const bool h1 = aabbFast(b0, b1, rs, d1) && d1 < tmax;
    const bool second = ((__float_as_int(a1.w) >> ord) & 1) != 0;  // this ordering visits child 1 first
    const int wN = __float_as_int(second ? b0.w : a0.w), wF = __float_as_int(second ? a0.w : b0.w);
    // near / far selection on LANE MASKS (scalar and / or; written with bools the compiler goes through 0 / 1 VGPR values: seven VALU)
    const unsigned long long S = ballotb(second), H0 = ballotb(h0), H1 = ballotb(h1);
    const bool hN = __builtin_amdgcn_inverse_ballot_w64((H1 & S) | (H0 & ~S)), hF = __builtin_amdgcn_inverse_ballot_w64((H0 & S) | (H1 & ~S));
    const float dF = second ? d0 : d1;
    if (COUNT) ws.nodes++;  // the near child, now; the far one when it is popped
    if (hF || COUNT) pairPush(stk, lane, ovf, sp, wF, hF ? dF : __builtin_inff());
    cur = kPairNone;
    if (hN) {
        if (wN >= 0) pending = wN;
        else cur = ~wN;
    }
}

// ---- PACKET walk: the rays of a wave through the threaded order TOGETHER ------------------------------------------------------
// For rays that start side by side — the primary rays of an 8x8 pixel block — the walks of the 64 lanes overlap almost entirely:
// measured on the teapots camera (scripts/packet_walk_model.py) a wave's lanes visit 89 nodes each and 107 DIFFERENT nodes between
// them.  So the wave walks ONE node at a time, n = the smallest node any of its lanes wants next, and only the lanes that want n
// (p == n) test its box and move on (p = n + 1 on a hit, nextNodeIfMiss on a miss); the others wait at the node they skipped
// to.  In the threaded pre-order a lane that waits skipped from an ancestor-or-self m of n, and nextNodeIfMiss[m] >= nextNodeIfMiss[n]
// because subtrees nest, so the next n needs no reduction over lanes: n + 1 if any lane hit, else nextNodeIfMiss[n].  Every lane
// makes exactly the visits of DevScene::intersect (scene.h:262-301) for its ray, in the same order, against its own closest distance:
// same hit records, same counters.  What changes is the cost of a visit: the node record is ONE uniform load for the wave instead
// of 64 lane requests to the vector L1 — the walkers' ceiling does not apply.  Rays of class 0 only (`mine`); the array
// `nd` is the ordering the participating lanes share (the caller loops over the orderings present in the wave).
#ifndef RD_PACKET_BUDGET
#define RD_PACKET_BUDGET 256  // 96 … 384 measured: k_gbuffer_packet 377 / 365 / 367 / 351 / 355 us, k_walk_packet 338 / 334 / 337 / 326 / 325 (384 without a budget)
#endif
constexpr int kPacketBudget = RD_PACKET_BUDGET;  // visits a wave makes as a packet before its lanes part (see the end of packetWalk)
template <bool COUNT>
RD_DEV void packetWalk(const DScene &s, const NodeRec *__restrict__ nd, bool mine, const RaySlab &rs, float &tmax, int &hitPrim, v2 &hitBary,
                       WalkStats &ws, int budget = kPacketBudget) {
    const int end = s.bvhSize;
    const RaySlabPk rp = packSlab(rs);  // the slab test on register pairs (aabbFastPk: the bits of aabbFast)
    int p = mine ? 0 : end;
    int n = __ballot(mine) != 0ull ? 0 : end;
    // (Requesting the two candidates for the next step — n + 1, and the skip target once record n is here — while the lanes test box
    // n was measured: k_gbuffer_packet 389 -> 469 us, k_walk_packet 351 -> 420 us.  Three scalar loads per visit instead of one cost
    // more than the latency they hide: eight waves per SIMD hide it already.)
    // (constant address space: the records are read-only for the launch, and a uniform load from it is a scalar load whatever the
    // kernel has stored before the walk — without it the compiler falls back to four VECTOR loads per visit as soon as a store
    // precedes the loop.  A resident grid striding over the blocks, which this makes possible, was measured: k_walk_packet 338 ->
    // 428-492 us at 4 096 / 2 048 / 1 024 workgroups; the one-shot grid balances itself.)
    typedef float rawFloat4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(4))) const rawFloat4 ConstFloat4;
    ConstFloat4 *const ndc = (ConstFloat4 *)nd;  // record n = vectors 2 n (lo_prim) and 2 n + 1 (hi_next)
    while (n != end) {
        const rawFloat4 lo4 = ndc[2 * (size_t)n], hi4 = ndc[2 * (size_t)n + 1];  // a uniform address: one scalar load for the wave
        const float4 lo = make_float4(lo4.x, lo4.y, lo4.z, lo4.w), hi = make_float4(hi4.x, hi4.y, hi4.z, hi4.w);
        const int prim = __builtin_amdgcn_readfirstlane(__float_as_int(lo.w)), nxt = __builtin_amdgcn_readfirstlane(__float_as_int(hi.w));
        const bool act = p == n;
        bool hit = false;
        if (act) {
            float boundDist;
            if (COUNT) ws.nodes++;
            hit = aabbFastPk(lo, hi, rp, boundDist) && boundDist < tmax;
            p = hit ? n + 1 : nxt;
        }
        const bool any = __ballot(hit) != 0ull;
        if (prim >= 0 && any) {  // a leaf some lanes enter: their triangle test, now (scene.h:281-292)
            if (hit) {
                const TriVerts tv = loadTri(s.tris, prim);
                float dist;
                v2 bary;
                if (COUNT) ws.tris++;
                if (intersectTriangle(rs, tv.a, tv.b, tv.c, bary, dist) && dist < tmax) {
                    hitPrim = prim;
                    tmax = dist;
                    hitBary = bary;
                }
            }
        }
        n = any ? n + 1 : nxt;
        if (--budget == 0) break;
    }
    // A block on a silhouette holds groups of rays that walk different subtrees, and as a packet the wave walks them one after the
    // other: the union it visits is 107 nodes on average but up to 1 300 (scripts/packet_walk_model.py), and that ONE wave then lasts as
    // long as the whole launch.  After kPacketBudget visits the lanes that are not done go on EACH ON ITS OWN — node p, own closest
    // distance, own hit: the threaded walk needs no other state — with one record fetch per lane and step, which costs a ray at most
    // its own ~90 visits however far the packet's rays have parted.
    while (__ballot(p != end) != 0ull) {
        if (p != end) {
            const float4 lo = nd[p].lo_prim, hi = nd[p].hi_next;
            float boundDist;
            if (COUNT) ws.nodes++;
            if (aabbFastPk(lo, hi, rp, boundDist) && boundDist < tmax) {
                const int prim = __float_as_int(lo.w);
                if (prim >= 0) {
                    const TriVerts tv = loadTri(s.tris, prim);
                    float dist;
                    v2 bary;
                    if (COUNT) ws.tris++;
                    if (intersectTriangle(rs, tv.a, tv.b, tv.c, bary, dist) && dist < tmax) {
                        hitPrim = prim;
                        tmax = dist;
                        hitBary = bary;
                    }
                }
                p++;
            } else {
                p = __float_as_int(hi.w);
            }
        }
    }
}
// The packet walks of a wave whose lanes hold rays of possibly different orderings (a block of primary rays: one, rarely two or three).
template <bool COUNT>
RD_DEV void packetWalkAll(const DScene &s, bool mine, int ord, const RaySlab &rs, float &tmax, int &hitPrim, v2 &hitBary, WalkStats &ws,
                          int budget = kPacketBudget) {
    unsigned long long todo = __ballot(mine);
    while (todo) {
        const int L = __ffsll((long long)todo) - 1;
        const int q = __builtin_amdgcn_readlane(ord, L);
        const bool now = mine && ord == q;
        todo &= ~__ballot(now);
        packetWalk<COUNT>(s, s.nodes[0] + (size_t)q * (size_t)(s.bvhSize + 1), now, rs, tmax, hitPrim, hitBary, ws, budget);
    }
}

// Wave-level reduction of the per-lane walk statistics, then one atomic per counter per wave.
RD_DEV unsigned long long waveSum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
RD_DEV void flushCounters(Counters *c, unsigned closest, unsigned any, unsigned hits, const WalkStats &ws) {
    unsigned long long a = waveSum(closest), b = waveSum(any), n = waveSum(ws.nodes), t = waveSum(ws.tris),
                       h = waveSum(hits);
    if ((threadIdx.x & 63) == 0) {
        if (a) atomicAdd(&c->closestRays, a);
        if (b) atomicAdd(&c->anyRays, b);
        if (n) atomicAdd(&c->nodeVisits, n);
        if (t) atomicAdd(&c->triTests, t);
        if (h) atomicAdd(&c->closestHits, h);
    }
}

}  // namespace rd
