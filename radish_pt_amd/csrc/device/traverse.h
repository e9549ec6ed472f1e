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

// One threaded-BVH walk per lane (DevScene::intersect, scene.h:262-301, and DevScene::testOcclusion's loop,
// :316-333).  Each lane performs exactly the reference's sequence — box test, on a leaf hit the triangle test, strict
// `<` updates, node++ / nextNodeIfMiss — but the wave runs it "while-while": lanes step through boxes until enough of
// them are parked on a leaf, then the parked lanes test their triangles together.  Only ~3 % of visits reach a
// triangle, so testing as soon as ONE lane needs it (the if-if form) makes every step pay for a triangle test.
// ANY = true: any-hit with a fixed distance bound; returns true on the first accepted triangle.
template <bool COUNT, bool ANY>
RD_DEV bool walkRay(const DScene &s, const Ray &ray, float tLimit, HitRec &h, WalkStats &ws) {
    RaySlab rs = makeRaySlab(ray);
    const NodeRec *nodes = s.nodes[getMTBVHId(-ray.d)];
    h.prim = -1;
    h.bary = mk2(0.f, 0.f);
    h.dist = tLimit;
    int node = 0;
    const int end = s.bvhSize;
    int pending = -1;
    bool alive = node != end;
    bool found = false;
    for (;;) {
        for (;;) {
            bool walking = alive && pending < 0;
            unsigned long long wm = __ballot(walking);
            if (wm == 0ull) break;
            unsigned long long pm = __ballot(alive && pending >= 0);
            if (__popcll(pm) * RD_LEAF_DEN >= (__popcll(wm) + __popcll(pm)) * RD_LEAF_NUM && pm != 0ull) break;
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
RD_DEV HitRec traceClosest(const DScene &s, const Ray &ray, WalkStats &ws) {
    HitRec h;
    walkRay<COUNT, false>(s, ray, 3.402823466e+38f /* FLT_MAX */, h, ws);
    return h;
}

// DevScene::testOcclusion (scene.h:303-334)
template <bool COUNT>
RD_DEV bool traceOccluded(const DScene &s, v3 x, v3 y, WalkStats &ws) {
    const float eps = 1e-4f;
    v3 dir = y - x;
    float dist = length(dir);
    dir = dir / dist;
    dist -= eps;
    Ray ray = makeOffsetedRay(x, dir);
    HitRec h;
    return walkRay<COUNT, true>(s, ray, dist, h, ws);
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
