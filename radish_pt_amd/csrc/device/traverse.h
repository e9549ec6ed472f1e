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
    int cls;  // 0: no special case applies; 1/2/3: |d.x|/|d.y|/|d.z| > 1-Eps; 4: some |d.k| < Eps
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

// AABB::intersect (bvh.h:91-155)
RD_DEV bool aabbIntersect(v3 pMin, v3 pMax, const RaySlab &r, float &tMin) {
    const float Eps = 1e-6f;
    if (r.cls != 0 && r.cls != 4) {
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

// intersectTriangle (intersections.h:20-68)
RD_DEV bool intersectTriangle(const RaySlab &ray, v3 v0, v3 v1, v3 vc, v2 &bary, float &dist) {
    v3 e01 = v1 - v0;
    v3 e02 = vc - v0;
    v3 pvec = cross(ray.d, e02);
    float det = dot(e01, pvec);
    if (fabs_(det) < 1.1920928955078125e-7f) return false;  // FLT_EPSILON
    v3 v0ToOri = ray.o - v0;
    if (det < 0.f) {
        det = -det;
        v0ToOri = -v0ToOri;
    }
    bary.x = dot(v0ToOri, pvec);
    if (bary.x < 0.f || bary.x > det) return false;
    v3 qvec = cross(v0ToOri, e01);
    bary.y = dot(ray.d, qvec);
    if (bary.y < 0.f || bary.x + bary.y > det) return false;
    float invDet = 1.f / det;
    bary.x = bary.x * invDet;
    bary.y = bary.y * invDet;
    dist = dot(e02, qvec) * invDet;
    return dist > 0.f;
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

// DevScene::intersect (scene.h:262-301), geometry part.  `prim` = NullPrimitive (-1) on a miss.
template <bool COUNT>
RD_DEV HitRec traceClosest(const DScene &s, const Ray &ray, WalkStats &ws) {
    RaySlab rs = makeRaySlab(ray);
    const NodeRec *nodes = s.nodes[getMTBVHId(-ray.d)];
    HitRec h;
    h.prim = -1;
    h.bary = mk2(0.f, 0.f);
    h.dist = 3.402823466e+38f;  // FLT_MAX
    int node = 0;
    const int end = s.bvhSize;
    while (node != end) {
        float4 lo = nodes[node].lo_prim;
        float4 hi = nodes[node].hi_next;
        float boundDist;
        if (COUNT) ws.nodes++;
        bool boundHit = aabbIntersect(mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), rs, boundDist);
        if (boundHit && boundDist < h.dist) {
            int primId = __float_as_int(lo.w);
            if (primId != -1) {
                TriVerts t = loadTri(s.tris, primId);
                float dist;
                v2 bary;
                if (COUNT) ws.tris++;
                bool hit = intersectTriangle(rs, t.a, t.b, t.c, bary, dist);
                if (hit && dist < h.dist) {
                    h.prim = primId;
                    h.dist = dist;
                    h.bary = bary;
                }
            }
            node++;
        } else {
            node = __float_as_int(hi.w);
        }
    }
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
    RaySlab rs = makeRaySlab(ray);
    const NodeRec *nodes = s.nodes[getMTBVHId(-ray.d)];
    int node = 0;
    const int end = s.bvhSize;
    while (node != end) {
        float4 lo = nodes[node].lo_prim;
        float4 hi = nodes[node].hi_next;
        float boundDist;
        if (COUNT) ws.nodes++;
        bool boundHit = aabbIntersect(mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z), rs, boundDist);
        if (boundHit && boundDist < dist) {
            int primId = __float_as_int(lo.w);
            if (primId != -1) {
                TriVerts t = loadTri(s.tris, primId);
                float d;
                v2 bary;
                if (COUNT) ws.tris++;
                bool hit = intersectTriangle(rs, t.a, t.b, t.c, bary, d);
                if (hit && d < dist) return true;
            }
            node++;
        } else {
            node = __float_as_int(hi.w);
        }
    }
    return false;
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
