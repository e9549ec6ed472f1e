// oracle/oracle.cpp — TEST INFRASTRUCTURE ONLY (see oracle.h for the usage rule and the
// "parity unpinned" statement).
//
// Single-threaded CPU restatement of the reference's per-pixel semantics.  Every function cites the
// reference lines it follows (paths relative to /root/reference/).  Structure deliberately mirrors
// the reference megakernels (one pixel at a time, whole path in one loop) — the HIP product is a
// wavefront design with its own data layout, so agreement between the two is a real check.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
#include "oracle.h"

#include <cstdlib>
#include <cstring>
#include <vector>

#include "omath.h"

using namespace om;

namespace {

// ------------------------------------------------------------------------------------------------
// Layouts.  All are plain 4-byte-aligned aggregates, sizes asserted below.
// ------------------------------------------------------------------------------------------------
struct Ray {  // src/sceneStructs.h:13-19
    vec3 origin, direction;
};
struct Camera {  // src/sceneStructs.h:118-130
    int resx, resy;
    vec3 position, rotation, view, up, right;
    vec2 fov, pixelLength;
    mat3 rotationMatInv;
    float viewProjection[16];
    float lensRadius, focalDist, tanFovY;
};
static_assert(sizeof(Camera) == 196, "Camera layout");
struct AABB {  // src/bvh.h:157-158
    vec3 pMin, pMax;
};
static_assert(sizeof(AABB) == 24, "AABB layout");
struct MTBVHNode {  // src/bvh.h:167-169
    int primitiveId, boundingBoxId, nextNodeIfMiss;
};
static_assert(sizeof(MTBVHNode) == 12, "MTBVHNode layout");
enum MaterialType { Lambertian = 0, MetallicWorkflow = 1, Dielectric = 2, Disney = 3, Light = 4 };  // material.h:129
struct Material {  // src/material.h:276-286
    int type;
    vec3 baseColor;
    float metallic, roughness, ior;
    int baseColorMapId, normalMapId, metallicMapId, roughnessMapId;
};
static_assert(sizeof(Material) == 44, "Material layout");
struct BinomialDistrib {  // src/sampler.h:66-69
    float prob;
    int failId;
};
struct LightLiSample {  // src/restir.h:95-99; value-initialised by Reservoir (`SampleT sample = SampleT()`, :88)
    vec3 Li, wi;
    float dist = 0.f;
};
struct DirectReservoir {  // src/restir.h:10-92 (field order :88-92)
    LightLiSample sample;
    int numSamples = 0;
    float weight = 0.f;
};
static_assert(sizeof(DirectReservoir) == 36, "Reservoir layout");

enum BSDFSampleType : uint32_t {  // src/material.h:18-26
    Diffuse = 1 << 0, Glossy = 1 << 1, Specular = 1 << 2, Reflection = 1 << 4, Transmission = 1 << 5, Invalid = 1 << 15
};
struct BSDFSample {  // src/material.h:28-33
    vec3 dir, bsdf;
    float pdf;
    uint32_t type;
};
struct Intersection {  // src/sceneStructs.h:163-190 (only the fields the path uses)
    int primId, matId;
    vec3 pos, norm;
    vec2 uv;
    vec3 wo;
};

constexpr int NullPrimitive = -1;     // src/bvh.h:13
constexpr int NullTextureId = -1;     // src/material.h:13
constexpr float INVALID_PDF = -1.f;   // src/material.h:16
constexpr float PI_F = 3.1415926535897932384626422832795028841971f;      // mathUtil.h:15
constexpr float TWO_PI_F = 6.2831853071795864769252867665590057683943f;  // mathUtil.h:16
// INV_PI is the unparenthesised macro `1.f / PI` (mathUtil.h:17): `x * INV_PI` parses as
// (x * 1.f) / PI.  x * 1.f is exact, so every use below is written `x / PI_F`.
constexpr int SobolSampleNum = 10000, SobolSampleDim = 200;  // sampler.h:12-13

}  // namespace

struct orc_scene {
    orc_scene_desc d;
    orc_stats st;
    // analysis hooks (orc_debug_visit_hist): per (ordering, node) visit counts and per-ray visit-count log2 histogram
    uint32_t *visitHist = nullptr;
    uint64_t *rayLenHist = nullptr;  // [2][32]: closest / any, bucket = floor(log2(visits))+1, 0 for 0 visits; then [2][32] sums
};

namespace {

// ------------------------------------------------------------------------------------------------
// Math helpers — src/mathUtil.h
// ------------------------------------------------------------------------------------------------
inline uint32_t utilhash(uint32_t a) {  // mathUtil.h:199-207
    a = (a + 0x7ed55d16) + (a << 12);
    a = (a ^ 0xc761c23c) ^ (a >> 19);
    a = (a + 0x165667b1) + (a << 5);
    a = (a + 0xd3a2646c) ^ (a << 9);
    a = (a + 0xfd7046c5) + (a << 3);
    a = (a ^ 0xb55a4f09) ^ (a >> 16);
    return a;
}
inline vec3 HDRToLDR(vec3 c) { return c / (c + 1.f) * 1.f; }  // mathUtil.h:49-51
inline bool isNanOrInf(float x) { return isnan_f(x) || isinf_f(x); }  // mathUtil.h:58-60
inline bool hasNanOrInf(vec3 v) { return isNanOrInf(v.x) || isNanOrInf(v.y) || isNanOrInf(v.z); }  // :62-65
inline float satDot(vec3 a, vec3 b) { return gmax(dot(a, b), 0.f); }  // :67-69
inline float absDot(vec3 a, vec3 b) { return fabsf(dot(a, b)); }      // :71-73
inline float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }  // :74-77
inline float powerHeuristic(float f, float g) { float f2 = f * f; return f2 / (f2 + g * g); }  // :81-84
inline float triangleArea(vec3 v0, vec3 v1, vec3 v2) { return length(cross(v1 - v0, v2 - v0)) * 0.5f; }  // :90-93
inline vec3 triangleNormal(vec3 v0, vec3 v1, vec3 v2) { return normalize(cross(v1 - v0, v2 - v0)); }     // :95-98
inline vec3 sampleTriangleUniform(vec3 v0, vec3 v1, vec3 v2, float ru, float rv) {  // :100-108
    float r = sqrtf(rv);
    float u = 1.f - r;
    float v = ru * r;
    return v1 * u + v2 * v + v0 * (1.f - u - v);
}
inline float luminance(vec3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }  // :128-130
inline vec2 concentricSampleDisk(float x, float y) {  // :132-136
    float r = sqrtf(x);
    float theta = TWO_PI_F * y;
    float s, c;
    sincos_det(theta, &s, &c);
    return vec2(r * c, r * s);
}
inline vec3 toSphere(vec2 v) {  // :138-142
    v = v * vec2(TWO_PI_F, PI_F);
    float sx, cx, sy, cy;
    sincos_det(v.x, &sx, &cx);
    sincos_det(v.y, &sy, &cy);
    return vec3(cx * sy, cy, sx * sy);
}
inline vec2 toPlane(vec3 v) {  // :143-147 (Blinn–Newell); `x * INV_PI * 0.5f` parses as ((x * 1.f) / PI) * 0.5f
    return vec2(fract(atan2_det(v.z, v.x) / PI_F * 0.5f + 1.f), atan2_det(length(vec2(v.x, v.z)), v.y) / PI_F);
}
inline mat3 localRefMatrix(vec3 n) {  // :149-155
    vec3 t = (fabsf(n.y) > 0.9999f) ? vec3(0.f, 0.f, 1.f) : vec3(0.f, 1.f, 0.f);
    vec3 b = normalize(cross(n, t));
    t = cross(b, n);
    return mat3(t, b, n);
}
inline vec3 localToWorld(vec3 n, vec3 v) { return normalize(localRefMatrix(n) * v); }  // :157-159
inline vec3 cosineSampleHemisphere(vec3 n, float rx, float ry) {  // :161-166
    vec2 d = concentricSampleDisk(rx, ry);
    float z = sqrtf(1.f - dot(d, d));
    return localToWorld(n, vec3(d, z));
}
inline bool refract(vec3 n, vec3 wi, float ior, vec3 &wt) {  // :168-186
    float cosIn = dot(n, wi);
    if (cosIn < 0) ior = 1.f / ior;
    float sin2In = gmax(0.f, 1.f - cosIn * cosIn);
    float sin2Tr = sin2In / (ior * ior);
    if (sin2Tr >= 1.f) return false;
    float cosTr = sqrtf(1.f - sin2Tr);
    if (cosIn < 0) cosTr = -cosTr;
    wt = normalize(-wi / ior + n * (cosIn / ior - cosTr));
    return true;
}
inline float pdfAreaToSolidAngle(float pdf, vec3 x, vec3 y, vec3 ny) {  // :188-192
    vec3 yx = x - y;
    return pdf * dot(yx, yx) / absDot(ny, normalize(yx));
}

// ------------------------------------------------------------------------------------------------
// Sampler — src/sampler.h:15-37,54-64.  Argument evaluation order of `glm::vec2(sample1D(s),
// sample1D(s))` is unspecified in C++; this restatement draws left to right (x first).
// ------------------------------------------------------------------------------------------------
struct Sampler {
    const uint32_t *data;
    uint32_t scramble;
    int ptr;
    float sample() {  // sampler.h:21-25
        uint32_t r = data[ptr++] ^ scramble;
        scramble = utilhash(scramble);
        return float(r) * 0x1p-32f;
    }
};
inline Sampler makeSeededRandomEngine(int iter, int index, int dim, const uint32_t *data) {  // :32-35
    return Sampler{data, utilhash((uint32_t)index), iter * SobolSampleDim + dim};
}
inline float sample1D(Sampler &s) { return s.sample(); }
inline vec2 sample2D(Sampler &s) { float a = s.sample(); float b = s.sample(); return vec2(a, b); }
inline vec3 sample3D(Sampler &s) { vec2 a = sample2D(s); float b = s.sample(); return vec3(a, b); }
inline vec4 sample4D(Sampler &s) { vec3 a = sample3D(s); float b = s.sample(); return vec4{a.x, a.y, a.z, b}; }

// ------------------------------------------------------------------------------------------------
// Ray / triangle — src/intersections.h
// ------------------------------------------------------------------------------------------------
inline Ray makeOffsetedRay(vec3 ori, vec3 dir) { return {ori + dir * 1e-5f, dir}; }  // intersections.h:16-18

inline bool intersectTriangle(Ray ray, vec3 v0, vec3 v1, vec3 v2, vec2 &bary, float &dist) {  // :20-68
    vec3 e01 = v1 - v0;
    vec3 e02 = v2 - v0;
    vec3 ori = ray.origin;
    vec3 dir = ray.direction;
    vec3 pvec = cross(dir, e02);
    float det = dot(e01, pvec);
    if (fabsf(det) < FLT_EPSILON) return false;
    vec3 v0ToOri = ori - v0;
    if (det < 0.f) {
        det = -det;
        v0ToOri = -v0ToOri;
    }
    bary.x = dot(v0ToOri, pvec);
    if (bary.x < 0.f || bary.x > det) return false;
    vec3 qvec = cross(v0ToOri, e01);
    bary.y = dot(dir, qvec);
    if (bary.y < 0.f || bary.x + bary.y > det) return false;
    float invDet = 1.f / det;
    bary = bary * invDet;
    dist = dot(e02, qvec) * invDet;
    return dist > 0.f;
}

// ------------------------------------------------------------------------------------------------
// AABB slab test — src/bvh.h:72-155
// ------------------------------------------------------------------------------------------------
inline bool between(float x, float mn, float mx) { return x >= mn && x <= mx; }  // mathUtil.h:34-36
inline bool getDistMinMax(float tMin1, float tMin2, float tMax1, float tMax2, float &tMin) {  // bvh.h:72-78
    tMin = c_fminf(tMin1, tMin2);
    float tMax = c_fmaxf(tMax1, tMax2);
    return (tMax >= 0.f && tMax >= tMin);
}
inline bool getDistMaxMin(float tMin1, float tMin2, float tMax1, float tMax2, float &tMin) {  // bvh.h:80-86
    tMin = c_fmaxf(tMin1, tMin2);
    float tMax = c_fminf(tMax1, tMax2);
    return (tMax >= 0.f && tMax >= tMin);
}
inline bool aabbIntersect(const AABB &b, Ray ray, float &tMin) {  // bvh.h:91-155
    const float Eps = 1e-6f;
    vec3 pMin = b.pMin, pMax = b.pMax;
    vec3 ori = ray.origin;
    vec3 dir = ray.direction;
    if (fabsf(dir.x) > 1.f - Eps) {
        if (between(ori.y, pMin.y, pMax.y) && between(ori.z, pMin.z, pMax.z)) {
            float dirInvX = 1.f / dir.x;
            float t1 = (pMin.x - ori.x) * dirInvX;
            float t2 = (pMax.x - ori.x) * dirInvX;
            return getDistMinMax(t1, t2, t1, t2, tMin);
        } else {
            return false;
        }
    } else if (fabsf(dir.y) > 1.f - Eps) {
        if (between(ori.z, pMin.z, pMax.z) && between(ori.x, pMin.x, pMax.x)) {
            float dirInvY = 1.f / dir.y;
            float t1 = (pMin.y - ori.y) * dirInvY;
            float t2 = (pMax.y - ori.y) * dirInvY;
            return getDistMinMax(t1, t2, t1, t2, tMin);
        } else {
            return false;
        }
    } else if (fabsf(dir.z) > 1.f - Eps) {
        if (between(ori.x, pMin.x, pMax.x) && between(ori.y, pMin.y, pMax.y)) {
            float dirInvZ = 1.f / dir.z;
            float t1 = (pMin.z - ori.z) * dirInvZ;
            float t2 = (pMax.z - ori.z) * dirInvZ;
            return getDistMinMax(t1, t2, t1, t2, tMin);
        } else {
            return false;
        }
    }
    vec3 dirInv = 1.f / dir;
    vec3 t1 = (pMin - ori) * dirInv;
    vec3 t2 = (pMax - ori) * dirInv;
    vec3 tNear = gmin(t1, t2);
    vec3 tFar = gmax(t1, t2);
    vec3 tDist = tFar - tNear;
    float yz = tFar.z - tNear.y;
    float zx = tFar.x - tNear.z;
    float xy = tFar.y - tNear.x;
    if (fabsf(dir.x) < Eps && tDist.y + tDist.z > yz) return getDistMaxMin(tNear.y, tNear.z, tFar.y, tFar.z, tMin);
    if (fabsf(dir.y) < Eps && tDist.z + tDist.x > zx) return getDistMaxMin(tNear.z, tNear.x, tFar.z, tFar.x, tMin);
    if (fabsf(dir.z) < Eps && tDist.x + tDist.y > xy) return getDistMaxMin(tNear.x, tNear.y, tFar.x, tFar.y, tMin);
    if (tDist.y + tDist.z > yz && tDist.z + tDist.x > zx && tDist.x + tDist.y > xy) {
        return getDistMaxMin(c_fmaxf(tNear.x, tNear.y), tNear.z, c_fminf(tFar.x, tFar.y), tFar.z, tMin);
    }
    return false;
}

// ------------------------------------------------------------------------------------------------
// BSDFs — src/material.h
// ------------------------------------------------------------------------------------------------
inline vec3 fresnelSchlick(float lDotH, vec3 f0) { return mix(f0, vec3(1.f), pow5(1.f - lDotH)); }  // material.h:40-42
inline float fresnel(float cosIn, float ior) {  // material.h:44-64 (exact branch: SURVEY App. A on the misspelt #if)
    if (cosIn < 0.f) {
        ior = 1.f / ior;
        cosIn = -cosIn;
    }
    float sinIn = sqrtf(1.f - cosIn * cosIn);
    float sinTr = sinIn / ior;
    if (sinTr >= 1.f) return 1.f;
    float cosTr = sqrtf(1.f - sinTr * sinTr);
    float rPar = (cosIn - ior * cosTr) / (cosIn + ior * cosTr);
    float rPer = (ior * cosIn - cosTr) / (ior * cosIn + cosTr);
    return (rPar * rPar + rPer * rPer) * .5f;
}
inline float schlickG(float cosTheta, float alpha) {  // :68-71
    float a = alpha * .5f;
    return cosTheta / (cosTheta * (1.f - a) + a);
}
inline float smithG(float cosWo, float cosWi, float alpha) {  // :74-76
    return schlickG(fabsf(cosWo), alpha) * schlickG(fabsf(cosWi), alpha);
}
inline float ggxDistribution(float cosTheta, float alpha) {  // :79-88
    if (cosTheta < 1e-6f) return 0.f;
    float alpha2 = alpha * alpha;
    float nom = alpha2;
    float denom = (cosTheta * cosTheta) * (alpha2 - 1.f) + 1.f;
    denom = denom * denom * PI_F;
    return nom / denom;
}
inline float ggxPdf(vec3 n, vec3 m, vec3 wo, float alpha) {  // :92-97
    return ggxDistribution(dot(n, m), alpha) * schlickG(dot(n, wo), alpha) * absDot(m, wo) / absDot(n, wo);
}
inline vec3 ggxSample(vec3 n, vec3 wo, float alpha, vec2 r) {  // :106-126
    mat3 transMat = localRefMatrix(n);
    mat3 transInv = inverse(transMat);
    vec3 vh = normalize((transInv * wo) * vec3(alpha, alpha, 1.f));
    float lenSq = vh.x * vh.x + vh.y * vh.y;
    vec3 t = lenSq > 0.f ? vec3(-vh.y, vh.x, 0.f) / sqrtf(lenSq) : vec3(1.f, 0.f, 0.f);
    vec3 b = cross(vh, t);
    vec2 p = concentricSampleDisk(r.x, r.y);
    float s = 0.5f * (vh.z + 1.f);
    p.y = (1.f - s) * sqrtf(1.f - p.x * p.x) + s * p.y;
    vec3 h = t * p.x + b * p.y + vh * sqrtf(gmax(0.f, 1.f - dot(p, p)));
    h = vec3(h.x * alpha, h.y * alpha, gmax(0.f, h.z));
    return normalize(transMat * h);
}

inline vec3 lambertianBSDF(const Material &m) { return m.baseColor / PI_F; }  // :131-134
inline float lambertianPdf(vec3 n, vec3 wi) { return satDot(n, wi) / PI_F; }  // :136-139
inline void lambertianSample(const Material &m, vec3 n, vec3 r, BSDFSample &s) {  // :141-147
    s.dir = cosineSampleHemisphere(n, r.x, r.y);
    s.bsdf = m.baseColor / PI_F;
    s.pdf = satDot(n, s.dir) / PI_F;
    s.type = Diffuse | Reflection;
}
inline void dielectricSample(const Material &m, vec3 n, vec3 wo, vec3 r, BSDFSample &s) {  // :159-183
    float pdfRefl = fresnel(dot(n, wo), m.ior);
    s.bsdf = m.baseColor;
    if (r.z < pdfRefl) {
        s.dir = reflect(-wo, n);
        s.type = Specular | Reflection;
        s.pdf = 1.f;
    } else {
        bool result = refract(n, wo, m.ior, s.dir);
        if (!result) {
            s.type = Invalid;
            return;
        }
        float eta = m.ior;
        if (dot(n, wo) < 0) eta = 1.f / eta;
        s.type = Specular | Transmission;
        s.pdf = 1.f;
        s.bsdf /= eta * eta;
    }
}
inline vec3 metallicBSDF(const Material &m, vec3 n, vec3 wo, vec3 wi) {  // :187-205
    float alpha = m.roughness * m.roughness;
    vec3 h = normalize(wo + wi);
    float cosO = dot(n, wo);
    float cosI = dot(n, wi);
    if (cosI * cosO < 1e-7f) return vec3(0.f);
    vec3 f = fresnelSchlick(dot(h, wo), mix(vec3(.08f), m.baseColor, m.metallic));
    float d = ggxDistribution(dot(n, h), alpha);
    float g = smithG(cosO, cosI, alpha);
    return mix(m.baseColor / PI_F * (1.f - m.metallic), vec3(g * d / (4.f * cosI * cosO)), f);
}
inline float metallicPdf(const Material &m, vec3 n, vec3 wo, vec3 wi) {  // :207-213
    vec3 h = normalize(wo + wi);
    return mix(satDot(n, wi) / PI_F, ggxPdf(n, h, wo, m.roughness * m.roughness) / (4.f * absDot(h, wo)),
               1.f / (2.f - m.metallic));
}
inline void metallicSample(const Material &m, vec3 n, vec3 wo, vec3 r, BSDFSample &s) {  // :215-233
    float alpha = m.roughness * m.roughness;
    if (r.z > (1.f / (2.f - m.metallic))) {
        s.dir = cosineSampleHemisphere(n, r.x, r.y);
    } else {
        vec3 h = ggxSample(n, wo, alpha, vec2(r.x, r.y));
        s.dir = -reflect(wo, h);
    }
    if (dot(n, s.dir) < 0.f) {
        s.type = Invalid;
    } else {
        s.type = Glossy | Reflection;
        s.pdf = metallicPdf(m, n, wo, s.dir);
        s.bsdf = metallicBSDF(m, n, wo, s.dir);
    }
}
inline vec3 materialBSDF(const Material &m, vec3 n, vec3 wo, vec3 wi) {  // :235-246
    switch (m.type) {
    case Lambertian: return lambertianBSDF(m);
    case MetallicWorkflow: return metallicBSDF(m, n, wo, wi);
    case Dielectric: return vec3(0.f);
    }
    return vec3(0.f);
}
inline float materialPdf(const Material &m, vec3 n, vec3 wo, vec3 wi) {  // :248-258
    switch (m.type) {
    case Lambertian: return lambertianPdf(n, wi);
    case MetallicWorkflow: return metallicPdf(m, n, wo, wi);
    case Dielectric: return 0.f;
    }
    return 0.f;
}
inline void materialSample(const Material &m, vec3 n, vec3 wo, vec3 r, BSDFSample &s) {  // :260-275
    switch (m.type) {
    case Lambertian: lambertianSample(m, n, r, s); break;
    case MetallicWorkflow: metallicSample(m, n, wo, r, s); break;
    case Dielectric: dielectricSample(m, n, wo, r, s); break;
    default: s.type = Invalid;
    }
}

// ------------------------------------------------------------------------------------------------
// Camera — src/sceneStructs.h:21-131
// ------------------------------------------------------------------------------------------------
// glm::tan(glm::radians(fov.y)) is evaluated per call on the device in the reference (:75); here
// (and in the HIP host code) it is evaluated once per launch on the host with libm tanf.
inline float camTanFovY(const Camera &c) { return tanf(radians(c.fov.y)); }

inline Ray cameraSample(const Camera &c, float tanFovY, int x, int y, vec4 r) {  // :72-91
    Ray ray;
    float aspect = float(c.resx) / float(c.resy);
    vec2 pixelSize = 1.f / vec2(float(c.resx), float(c.resy));
    vec2 scr = vec2(float(x), float(y)) * pixelSize;
    vec2 ruv = scr + pixelSize * vec2(r.x, r.y);
    ruv = 1.f - ruv * 2.f;
    vec2 pAperture(0.f);
    vec3 pLens = vec3(pAperture * c.lensRadius, 0.f);
    vec3 pFocus = vec3(ruv * vec2(aspect, 1.f) * tanFovY, 1.f) * c.focalDist;
    vec3 dir = pFocus - pLens;
    ray.direction = normalize(mat3(c.right, c.up, c.view) * dir);
    ray.origin = c.position + c.right * pLens.x + c.up * pLens.y;
    return ray;
}
inline vec2 cameraRasterUV(const Camera &c, vec3 pos) {  // :22-43
    vec3 dir = normalize(pos - c.position);
    float d = 1.f / dot(dir, c.view);
    vec3 p = c.rotationMatInv * (dir * d);
    float aspect = float(c.resx) / float(c.resy);
    float tanFovY = camTanFovY(c);
    p = p / vec3(vec2(aspect, 1.f) * tanFovY, 1.f);
    vec2 ndc(p.x, p.y);
    ndc = -ndc;
    return ndc * .5f + .5f;
}

// ------------------------------------------------------------------------------------------------
// DevScene — src/scene.h:73-518
// ------------------------------------------------------------------------------------------------
// linearSample — src/image.h:42-87 (wrap-around bilinear)
inline vec3 linearSample(const vec3 *data, vec2 uv, int width, int height) {
    const float eps = FLT_MIN;
    uv = fract(uv);
    float fx = uv.x * (float(width) - eps) + 0.5f;
    float fy = uv.y * (float(height) - eps) + 0.5f;
    int ix = int(fract(fx) > 0.5f ? fx : fx - 1.f);
    int iy = int(fract(fy) > 0.5f ? fy : fy - 1.f);
    if (ix < 0) ix += width;
    if (iy < 0) iy += height;
    int ux = ix + 1;
    int uy = iy + 1;
    if (ux >= width) ux -= width;
    if (uy >= height) uy -= height;
    float lx = fract(fx + 0.5f);
    float ly = fract(fy + 0.5f);
    vec3 c1 = mix(data[iy * width + ix], data[iy * width + ux], lx);
    vec3 c2 = mix(data[uy * width + ix], data[uy * width + ux], lx);
    return mix(c1, c2, ly);
}
// proceduralTexture — src/scene.h:77-86.  thrust::default_random_engine is minstd_rand (x' = 48271 x mod 2^31-1,
// seed 0 → 1) and thrust::uniform_real_distribution<float> maps a draw u to float(u - 1) / (1.f + float(2^31 - 3))
// (thrust 1.15 / CUDA 11.7, thrust/random/detail/{linear_congruential_engine,uniform_real_distribution}.inl — not
// under /root/reference; restated from the published source, parity unpinned).
inline float minstdUniform(uint32_t &x) {
    x = uint32_t((uint64_t(x) * 48271ull) % 2147483647ull);
    return float(x - 1u) / (1.f + float(2147483646u - 1u));
}
inline vec3 proceduralTexture(vec2 uv) {
    uint32_t seed = uint32_t(int(uv.x * 1024) * 1024 + int(uv.y * 1024));
    uint32_t x = seed % 2147483647u;
    if (x == 0u) x = 1u;
    float rx = minstdUniform(x);
    float ry = minstdUniform(x);
    float s0, c0, s1, c1;
    sincos_det(uv.x * 10.f * TWO_PI_F + rx * TWO_PI_F, &s0, &c0);
    sincos_det(uv.y * 10.f * TWO_PI_F + ry * TWO_PI_F, &s1, &c1);
    float f = (s0 + +1.f) * .5f;
    float g = (s1 + +1.f) * .5f;
    return vec3(f * g);
}
constexpr int ProceduralTexId = -2;  // src/material.h:15

struct SceneView {
    orc_scene *h;
    const orc_scene_desc &d() const { return h->d; }
    const vec3 *vertices() const { return (const vec3 *)h->d.vertices; }
    const vec3 *normals() const { return (const vec3 *)h->d.normals; }
    const vec2 *texcoords() const { return (const vec2 *)h->d.texcoords; }
    const AABB *boxes() const { return (const AABB *)h->d.boundingBoxes; }
    const Material *materials() const { return (const Material *)h->d.materials; }

    vec3 texSample(int id, vec2 uv) const {  // DevTextureObj::linearSample (image.h:82-84)
        const orc_texture &t = d().textures[id];
        return linearSample((const vec3 *)t.data, uv, t.width, t.height);
    }
    bool hasEnvMap() const { return d().envMapTexId >= 0; }
    Material getTexturedMaterialAndSurface(Intersection &intersec) const {  // scene.h:88-112
        Material mat = materials()[intersec.matId];
        if (mat.baseColorMapId != NullTextureId) {
            mat.baseColor = mat.baseColorMapId == ProceduralTexId ? proceduralTexture(intersec.uv)
                                                                  : texSample(mat.baseColorMapId, intersec.uv);
        }
        if (mat.metallicMapId > NullTextureId) mat.metallic = texSample(mat.metallicMapId, intersec.uv).x;
        if (mat.roughnessMapId > NullTextureId) mat.roughness = texSample(mat.roughnessMapId, intersec.uv).x;
        if (mat.normalMapId != NullTextureId) {
            vec3 mapped = texSample(mat.normalMapId, intersec.uv);
            vec3 localNorm = normalize(vec3(mapped.x, mapped.y, mapped.z) * 1.f - 0.5f);  // sic
            intersec.norm = localToWorld(intersec.norm, localNorm);
        }
        return mat;
    }
    int envSample(float r1, float r2) const {  // envMapSampler.sample (sampler.h:204-208)
        const BinomialDistrib *t = (const BinomialDistrib *)d().envMapSampler;
        int length = d().envMapSamplerLength;
        int passId = gmin(int(float(length) * r1), length - 1);
        BinomialDistrib distrib = t[passId];
        return (r2 < distrib.prob) ? passId : distrib.failId;
    }
    vec3 envLookup(vec3 dir) const { return texSample(d().envMapTexId, toPlane(dir)); }
    float environmentMapPdf(vec3 wi) const {  // scene.h:374-378
        const orc_texture &e = d().textures[d().envMapTexId];
        vec3 radiance = envLookup(wi);
        return luminance(radiance) * d().sumLightPowerInv * e.width * e.height * 0.5f;
    }
    // sampleEnvironmentMap / sampleEnvMapNoVisbility (scene.h:380-414); visibility = false skips the occlusion test
    float sampleEnvironmentMap(vec3 pos, vec2 r, vec3 &radiance, vec3 &wi, bool visibility) const {
        const orc_texture &e = d().textures[d().envMapTexId];
        int pixId = envSample(r.x, r.y);
        int y = pixId / e.width;
        int x = pixId - y * e.width;
        radiance = ((const vec3 *)e.data)[pixId];
        wi = toSphere(vec2((x + 0.5f) / e.width, (y + 0.5f) / e.height));
        if (visibility) {
            bool occ = testOcclusion(pos, pos + wi * 1e6f);
            if (occ) return INVALID_PDF;
        }
        return luminance(radiance) * d().sumLightPowerInv * e.width * e.height / PI_F / PI_F * 0.5f;
    }
    static int getMTBVHId(vec3 dir) {  // scene.h:114-129
        vec3 absDir = gabs(dir);
        if (absDir.x > absDir.y) {
            if (absDir.x > absDir.z) return dir.x > 0 ? 0 : 1;
            else return dir.z > 0 ? 4 : 5;
        } else {
            if (absDir.y > absDir.z) return dir.y > 0 ? 2 : 3;
            else return dir.z > 0 ? 4 : 5;
        }
    }
    float getPrimitiveArea(int primId) const {  // scene.h:139-145
        return triangleArea(vertices()[primId * 3], vertices()[primId * 3 + 1], vertices()[primId * 3 + 2]);
    }
    void getIntersecGeomInfo(int primId, vec2 bary, Intersection &intersec) const {  // scene.h:147-165
        vec3 va = vertices()[primId * 3], vb = vertices()[primId * 3 + 1], vc = vertices()[primId * 3 + 2];
        vec3 na = normals()[primId * 3], nb = normals()[primId * 3 + 1], nc = normals()[primId * 3 + 2];
        vec2 ta = texcoords()[primId * 3], tb = texcoords()[primId * 3 + 1], tc = texcoords()[primId * 3 + 2];
        intersec.pos = vb * bary.x + vc * bary.y + va * (1.f - bary.x - bary.y);
        intersec.norm = normalize(nb * bary.x + nc * bary.y + na * (1.f - bary.x - bary.y));
        intersec.uv = tb * bary.x + tc * bary.y + ta * (1.f - bary.x - bary.y);
    }
    bool intersectPrim(int primId, Ray ray, float &dist, vec2 &bary) const {  // scene.h:167-179
        h->st.triTests++;
        return intersectTriangle(ray, vertices()[primId * 3], vertices()[primId * 3 + 1], vertices()[primId * 3 + 2],
                                 bary, dist);
    }
    bool intersectPrim(int primId, Ray ray, float distRange) const {  // scene.h:181-189
        vec2 bary;
        float dist;
        h->st.triTests++;
        bool hit = intersectTriangle(ray, vertices()[primId * 3], vertices()[primId * 3 + 1],
                                     vertices()[primId * 3 + 2], bary, dist);
        return (hit && dist < distRange);
    }

    // Closest hit, threaded-BVH walk — scene.h:262-301.  Returns the raw record too.
    void intersect(Ray ray, Intersection &intersec, orc_hit *rec = nullptr) const {
        h->st.closestRays++;
        int closestPrimId = NullPrimitive;
        vec2 closestBary;
        float closestDist = FLT_MAX;
        const int ordering = getMTBVHId(-ray.direction);
        const MTBVHNode *nodes = (const MTBVHNode *)d().bvhNodes[ordering];
        int node = 0;
        const int BVHSize = d().bvhSize;
        const uint64_t visitsBefore = h->st.nodeVisits;
        while (node != BVHSize) {
            const AABB &bound = boxes()[nodes[node].boundingBoxId];
            float boundDist;
            h->st.nodeVisits++;
            if (h->visitHist) h->visitHist[(size_t)ordering * BVHSize + node]++;
            bool boundHit = aabbIntersect(bound, ray, boundDist);
            if (boundHit && boundDist < closestDist) {
                int primId = nodes[node].primitiveId;
                if (primId != NullPrimitive) {
                    float dist;
                    vec2 bary;
                    bool hit = intersectPrim(primId, ray, dist, bary);
                    if (hit && dist < closestDist) {
                        closestPrimId = primId;
                        closestDist = dist;
                        closestBary = bary;
                    }
                }
                node++;
            } else {
                node = nodes[node].nextNodeIfMiss;
            }
        }
        logRayLen(0, h->st.nodeVisits - visitsBefore);
        if (closestPrimId != NullPrimitive) {
            h->st.closestHits++;
            getIntersecGeomInfo(closestPrimId, closestBary, intersec);
            intersec.matId = d().materialIds[closestPrimId];
        }
        intersec.primId = closestPrimId;
        if (rec) {
            rec->primId = closestPrimId;
            rec->u = closestPrimId != NullPrimitive ? closestBary.x : 0.f;
            rec->v = closestPrimId != NullPrimitive ? closestBary.y : 0.f;
            rec->t = closestPrimId != NullPrimitive ? closestDist : FLT_MAX;
        }
    }
    // scene.h:209-232
    void naiveIntersect(Ray ray, orc_hit *rec) const {
        float closestDist = FLT_MAX;
        vec2 closestBary;
        int closestPrimId = NullPrimitive;
        for (int i = 0; i < (d().bvhSize + 1) / 2; i++) {
            float dist;
            vec2 bary;
            bool hit = intersectPrim(i, ray, dist, bary);
            if (hit && dist < closestDist) {
                closestDist = dist;
                closestBary = bary;
                closestPrimId = i;
            }
        }
        rec->primId = closestPrimId;
        rec->u = closestPrimId != NullPrimitive ? closestBary.x : 0.f;
        rec->v = closestPrimId != NullPrimitive ? closestBary.y : 0.f;
        rec->t = closestPrimId != NullPrimitive ? closestDist : FLT_MAX;
    }
    // Any hit — scene.h:303-334
    bool testOcclusion(vec3 x, vec3 y) const {
        h->st.anyRays++;
        const float eps = 1e-4f;
        vec3 dir = y - x;
        float dist = length(dir);
        dir /= dist;
        dist -= eps;
        Ray ray = makeOffsetedRay(x, dir);
        const int ordering = getMTBVHId(-ray.direction);
        const MTBVHNode *nodes = (const MTBVHNode *)d().bvhNodes[ordering];
        int node = 0;
        const int BVHSize = d().bvhSize;
        const uint64_t visitsBefore = h->st.nodeVisits;
        while (node != BVHSize) {
            const AABB &bound = boxes()[nodes[node].boundingBoxId];
            float boundDist;
            h->st.nodeVisits++;
            if (h->visitHist) h->visitHist[(size_t)ordering * BVHSize + node]++;
            bool boundHit = aabbIntersect(bound, ray, boundDist);
            if (boundHit && boundDist < dist) {
                int primId = nodes[node].primitiveId;
                if (primId != NullPrimitive) {
                    if (intersectPrim(primId, ray, dist)) {
                        logRayLen(1, h->st.nodeVisits - visitsBefore);
                        return true;
                    }
                }
                node++;
            } else {
                node = nodes[node].nextNodeIfMiss;
            }
        }
        logRayLen(1, h->st.nodeVisits - visitsBefore);
        return false;
    }
    void logRayLen(int kind, uint64_t visits) const {
        if (!h->rayLenHist) return;
        int b = 0;
        for (uint64_t v = visits; v; v >>= 1) b++;
        h->rayLenHist[kind * 32 + b]++;
        h->rayLenHist[64 + kind * 32 + b] += visits;
    }
    int lightSample(float r1, float r2) const {  // sampler.h:204-208 (DevDiscreteSampler1D::sample)
        const BinomialDistrib *t = (const BinomialDistrib *)d().lightSampler;
        int length = d().lightSamplerLength;
        int passId = gmin(int(float(length) * r1), length - 1);
        BinomialDistrib distrib = t[passId];
        return (r2 < distrib.prob) ? passId : distrib.failId;
    }
    // scene.h:419-456
    float sampleDirectLight(vec3 pos, vec4 r, vec3 &radiance, vec3 &wi) const {
        if (d().lightSamplerLength == 0) return INVALID_PDF;
        int lightId = lightSample(r.x, r.y);
        if (lightId == d().lightSamplerLength - 1 && d().envMapSamplerLength != 0)
            return sampleEnvironmentMap(pos, vec2(r.z, r.w), radiance, wi, true);
        int primId = d().lightPrimIds[lightId];
        vec3 v0 = vertices()[primId * 3 + 0], v1 = vertices()[primId * 3 + 1], v2 = vertices()[primId * 3 + 2];
        vec3 sampled = sampleTriangleUniform(v0, v1, v2, r.z, r.w);
        bool occ = testOcclusion(pos, sampled);  // before the single-sided rejection
        if (occ) return INVALID_PDF;
        vec3 normal = triangleNormal(v0, v1, v2);
        vec3 posToSampled = sampled - pos;
        if (dot(normal, posToSampled) > -1e-6f) return INVALID_PDF;  // SCENE_LIGHT_SINGLE_SIDED
        float area = triangleArea(v0, v1, v2);
        radiance = ((const vec3 *)d().lightUnitRadiance)[lightId];
        wi = normalize(posToSampled);
        float power = luminance(radiance) / (area * 2.f * PI_F);
        return pdfAreaToSolidAngle(power * d().sumLightPowerInv, pos, sampled, normal);
    }
    // scene.h:458-492
    float sampleDirectLightNoVisibility(vec3 pos, vec4 r, vec3 &radiance, vec3 &wi, float &dist) const {
        if (d().lightSamplerLength == 0) return INVALID_PDF;
        int lightId = lightSample(r.x, r.y);
        if (lightId == d().lightSamplerLength - 1 && d().envMapSamplerLength != 0) {
            dist = 1e10f;
            return sampleEnvironmentMap(pos, vec2(r.z, r.w), radiance, wi, false);
        }
        int primId = d().lightPrimIds[lightId];
        vec3 v0 = vertices()[primId * 3 + 0], v1 = vertices()[primId * 3 + 1], v2 = vertices()[primId * 3 + 2];
        vec3 sampled = sampleTriangleUniform(v0, v1, v2, r.z, r.w);
        vec3 normal = triangleNormal(v0, v1, v2);
        vec3 posToSampled = sampled - pos;
        if (dot(normal, posToSampled) > -1e-6f) return INVALID_PDF;
        float area = triangleArea(v0, v1, v2);
        radiance = ((const vec3 *)d().lightUnitRadiance)[lightId];
        wi = normalize(posToSampled);
        dist = length(posToSampled);
        float power = luminance(radiance) / (area * 2.f * PI_F);
        return pdfAreaToSolidAngle(power * d().sumLightPowerInv, pos, sampled, normal);
    }
};

inline void storeRunningMean(float *img, int64_t index, vec3 v, int iter) {  // pathtrace.cu:287-290
    vec3 *p = (vec3 *)img + index;
    *p = (*p * float(iter) + v) / float(iter + 1);
}

// ------------------------------------------------------------------------------------------------
// singleKernelPT — src/pathtrace.cu:149-291, one pixel
// ------------------------------------------------------------------------------------------------
void pathTracePixel(const SceneView &scene, const Camera &cam, float tanFovY, int x, int y, int looper, int iter,
                    int maxDepth, float *directIllum, float *indirectIllum) {
    vec3 direct(0.f);
    vec3 indirect(0.f);
    int index = y * cam.resx + x;
    Sampler rng = makeSeededRandomEngine(looper, index, 0, scene.d().sobol);
    Ray ray = cameraSample(cam, tanFovY, x, y, sample4D(rng));
    Intersection intersec;
    scene.intersect(ray, intersec);
    do {  // `goto WriteRadiance` → break out of this block
        if (intersec.primId == NullPrimitive) {
            direct = vec3(1.f);
            break;
        }
        Material material = scene.getTexturedMaterialAndSurface(intersec);
        material.baseColor = vec3(1.f);  // DENOISER_DEMODULATE (:175-178)
        if (material.type == Light) {
            direct = vec3(1.f);
            break;
        }
        vec3 throughput(1.f);
        intersec.wo = -ray.direction;
        for (int depth = 1; depth <= maxDepth; depth++) {
            bool deltaBSDF = (material.type == Dielectric);
            if (material.type != Dielectric && dot(intersec.norm, intersec.wo) < 0.f) intersec.norm = -intersec.norm;
            if (!deltaBSDF) {
                vec3 radiance, wi;
                float lightPdf = scene.sampleDirectLight(intersec.pos, sample4D(rng), radiance, wi);
                if (lightPdf > 0.f) {
                    float BSDFPdf = materialPdf(material, intersec.norm, intersec.wo, wi);
                    (depth == 1 ? direct : indirect) +=
                        throughput * materialBSDF(material, intersec.norm, intersec.wo, wi) * radiance *
                        satDot(intersec.norm, wi) / lightPdf * powerHeuristic(lightPdf, BSDFPdf);
                }
            }
            BSDFSample sample;
            materialSample(material, intersec.norm, intersec.wo, sample3D(rng), sample);
            if (sample.type == Invalid) break;
            else if (sample.pdf < 1e-8f) break;
            bool deltaSample = (sample.type & Specular);
            throughput *= sample.bsdf / sample.pdf * (deltaSample ? 1.f : absDot(intersec.norm, sample.dir));
            ray = makeOffsetedRay(intersec.pos, sample.dir);
            vec3 curPos = intersec.pos;
            scene.intersect(ray, intersec);
            intersec.wo = -ray.direction;
            if (intersec.primId == NullPrimitive) {  // :232-247
                if (scene.hasEnvMap()) {
                    vec3 radiance = scene.envLookup(ray.direction) * throughput;
                    float weight = deltaSample ? 1.f : powerHeuristic(sample.pdf, scene.environmentMapPdf(ray.direction));
                    indirect += radiance * weight;
                }
                break;
            }
            material = scene.getTexturedMaterialAndSurface(intersec);
            if (material.type == Light) {
                if (dot(intersec.norm, ray.direction) < 0.f) break;  // SCENE_LIGHT_SINGLE_SIDED (:252-256)
                vec3 radiance = material.baseColor;
                float weight =
                    deltaSample ? 1.f
                                : powerHeuristic(sample.pdf, pdfAreaToSolidAngle(luminance(radiance) *
                                                                                     scene.d().sumLightPowerInv *
                                                                                     scene.getPrimitiveArea(intersec.primId),
                                                                                 curPos, intersec.pos, intersec.norm));
                indirect += radiance * throughput * weight;
                break;
            }
        }
    } while (false);
    if (hasNanOrInf(direct)) direct = vec3(0.f);
    if (hasNanOrInf(indirect)) indirect = vec3(0.f);
    direct = HDRToLDR(direct);
    indirect = HDRToLDR(indirect);
    storeRunningMean(directIllum, index, direct, iter);
    storeRunningMean(indirectIllum, index, indirect, iter);
}

// PTDirectKernel — src/pathtrace.cu:293-345
void pathTraceDirectPixel(const SceneView &scene, const Camera &cam, float tanFovY, int x, int y, int looper, int iter,
                          float *directIllum) {
    vec3 direct(0.f);
    int idx = x + y * cam.resx;
    Sampler rng = makeSeededRandomEngine(looper, idx, 0, scene.d().sobol);
    Ray ray = cameraSample(cam, tanFovY, x, y, sample4D(rng));
    Intersection intersec;
    scene.intersect(ray, intersec);
    do {
        if (intersec.primId == NullPrimitive) {  // :309-314
            if (scene.hasEnvMap()) direct = scene.envLookup(ray.direction);
            break;
        }
        Material material = scene.getTexturedMaterialAndSurface(intersec);
        if (material.type == Light) {
            direct = material.baseColor;
            break;
        }
        intersec.wo = -ray.direction;
        bool deltaBSDF = (material.type == Dielectric);
        if (!deltaBSDF && dot(intersec.norm, intersec.wo) < 0.f) intersec.norm = -intersec.norm;
        if (!deltaBSDF) {
            vec3 Li, wi;
            float lightPdf = scene.sampleDirectLight(intersec.pos, sample4D(rng), Li, wi);
            if (lightPdf > 0.f) {
                direct = Li * materialBSDF(material, intersec.norm, intersec.wo, wi) * satDot(intersec.norm, wi) /
                         lightPdf;
            }
        }
    } while (false);
    storeRunningMean(directIllum, idx, direct, iter);
}

// renderGBuffer — src/gBuffer.cu:3-76
void gbufferPixel(const SceneView &scene, const Camera &cam, const Camera &lastCam, float tanFovY, int x, int y,
                  orc_gbuffer *gb) {
    int idx = x + y * cam.resx;
    float aspect = float(cam.resx) / float(cam.resy);
    vec2 pixelsize = 1.f / vec2(float(cam.resx), float(cam.resy));
    vec2 scr = vec2(float(x), float(y)) * pixelsize;
    vec2 ruv = scr + pixelsize * vec2(0.5f);
    ruv = 1.f - ruv * 2.f;
    vec3 pLens(0.f);
    vec3 pFocus = vec3(ruv * vec2(aspect, 1.f) * tanFovY, 1.f) * cam.focalDist;
    vec3 dir = pFocus - pLens;
    Ray ray;
    ray.origin = cam.position + cam.right * pLens.x + cam.up * pLens.y;
    ray.direction = normalize(mat3(cam.right, cam.up, cam.view) * dir);
    Intersection intersect;
    scene.intersect(ray, intersect);
    int cur = gb->frameIdx;
    vec3 *albedo = (vec3 *)gb->albedo;
    vec3 *normal = (vec3 *)gb->normal[cur];
    if (intersect.primId != NullPrimitive) {
        bool isLight = scene.materials()[intersect.matId].type == Light;
        int matId = intersect.matId;
        if (isLight) matId = NullPrimitive - 1;  // (:36-43; the primId rewrite there is dead)
        Material material = scene.getTexturedMaterialAndSurface(intersect);
        albedo[idx] = material.baseColor;
        normal[idx] = intersect.norm;
        gb->primId[cur][idx] = matId;
        gb->depth[cur][idx] = distance(intersect.pos, ray.origin);
        vec2 ndc = cameraRasterUV(lastCam, intersect.pos);  // Camera::getRasterCoord (:45-48)
        vec2 rc = vec2(float(lastCam.resx), float(lastCam.resy)) * ndc;
        int lx = (int)rc.x, ly = (int)rc.y;
        if (lx >= 0 && lx < gb->width && ly >= 0 && ly < gb->height) gb->motion[idx] = ly * cam.resx + lx;
        else gb->motion[idx] = -1;
    } else {
        albedo[idx] = scene.hasEnvMap() ? scene.envLookup(ray.direction) : vec3(0.f);  // gBuffer.cu:61-66
        normal[idx] = vec3(0.f);
        gb->primId[cur][idx] = NullPrimitive;
        gb->depth[cur][idx] = 1.f;
        gb->motion[idx] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// Reservoir — src/restir.h:10-101
// ------------------------------------------------------------------------------------------------
inline void resvUpdate(DirectReservoir &r, const LightLiSample &newSample, float newWeight, float rand, bool faithful) {
    r.weight += newWeight;
    r.numSamples++;
    // restir.h:21 tests the float for truthiness: anything but exactly 0 replaces the sample.
    bool take = faithful ? ((rand * r.weight / newWeight) != 0.f) : (rand * r.weight < newWeight);
    if (take) r.sample = newSample;
}
inline void resvClear(DirectReservoir &r) { r.weight = 0.f; r.numSamples = 0; }  // :26-29 (keeps stale sample, Q3)
inline bool resvInvalid(const DirectReservoir &r) { return isNanOrInf(r.weight) || r.weight < 0.f; }  // :42
inline void resvCheckValidity(DirectReservoir &r) { if (resvInvalid(r)) resvClear(r); }  // :44-49
inline void resvMerge(DirectReservoir &r, const DirectReservoir &rhs, float rand) {  // :51-58
    r.weight += rhs.weight;
    r.numSamples += rhs.numSamples;
    if (rand * r.weight < rhs.weight) r.sample = rhs.sample;
}
inline void resvPreClampedMerge(DirectReservoir &r, DirectReservoir rhs, float rnd, int M) {  // :69-77
    if (rhs.numSamples > 0 && rhs.numSamples > (M - 1) * r.numSamples && r.numSamples > 0) {
        rhs.weight *= static_cast<float>(M - 1) * r.numSamples / rhs.numSamples;
        rhs.numSamples = (M - 1) * r.numSamples;
    }
    resvMerge(r, rhs, rnd);
}
inline vec3 resvPHat(const DirectReservoir &r, const Intersection &i, const Material &m) {  // :31-35
    return r.sample.Li * materialBSDF(m, i.norm, i.wo, r.sample.wi) * satDot(i.norm, r.sample.wi);
}
inline float resvW(const DirectReservoir &r, const Intersection &i, const Material &m) {  // :37-40
    return r.weight / (length(resvPHat(r, i, m)) * static_cast<float>(r.numSamples));
}

struct GBView {
    const orc_gbuffer *g;
    const int *getPrimId() const { return g->primId[g->frameIdx]; }
    const int *lastPrimId() const { return g->primId[g->frameIdx ^ 1]; }
    const vec3 *getNormal() const { return (const vec3 *)g->normal[g->frameIdx]; }
    const vec3 *lastNormal() const { return (const vec3 *)g->normal[g->frameIdx ^ 1]; }
    const float *getDepth() const { return g->depth[g->frameIdx]; }
};

DirectReservoir findTemporalNeighbor(const DirectReservoir *reservoir, int idx, const GBView &gb) {  // restir.cu:19-40
    int primId = gb.getPrimId()[idx];
    int lastIdx = gb.g->motion[idx];
    bool diff = false;
    if (lastIdx < 0) diff = true;
    else if (primId <= NullPrimitive) diff = true;
    else if (gb.lastPrimId()[lastIdx] != primId) diff = true;
    else {
        vec3 norm = gb.getNormal()[idx];
        vec3 lastNorm = gb.lastNormal()[lastIdx];
        if (absDot(norm, lastNorm) < .1f) diff = true;
    }
    return diff ? DirectReservoir() : reservoir[lastIdx];
}
DirectReservoir findSpatialNeighborDisk(const DirectReservoir *reservoir, int x, int y, const GBView &gb, vec2 rand) {
    // restir.cu:42-80
    const float radius = 5.f;
    int W = gb.g->width, H = gb.g->height;
    int idx = y * W + x;
    vec2 p = concentricSampleDisk(rand.x, rand.y) * radius;
    int px = (int)(float(x) + .5f + p.x);
    int py = (int)(float(y) + .5f + p.y);
    int pIdx = py * W + px;
    bool diff = false;
    if (px < 0 || px >= W || py < 0 || py >= H || (px == x && py == y)) diff = true;
    else if (gb.getPrimId()[pIdx] != gb.getPrimId()[idx]) diff = true;
    else {
        vec3 norm = gb.getNormal()[idx];
        vec3 pNorm = gb.getNormal()[pIdx];
        if (dot(norm, pNorm) < .1f) diff = true;
        float depth = gb.getDepth()[idx];
        float pDepth = gb.getDepth()[pIdx];
        if (fabsf(depth - pDepth) > depth * .1f) diff = true;
    }
    return diff ? DirectReservoir() : reservoir[pIdx];
}

struct RestirPixelState {  // what the reference keeps in registers across its (racy) barrier
    int status;            // 0 = shade, 1 = early exit (miss / emitter) with `direct` final
    vec3 direct;
    Intersection intersec;
    Material material;
    DirectReservoir reservoir;
    Sampler rng;
};

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

orc_scene *orc_scene_create(const orc_scene_desc *desc) {
    const Material *m = (const Material *)desc->materials;
    for (int i = 0; i < desc->numMaterials; i++) {
        int ids[4] = {m[i].baseColorMapId, m[i].normalMapId, m[i].metallicMapId, m[i].roughnessMapId};
        for (int k = 0; k < 4; k++)
            if (ids[k] >= desc->numTextures || ids[k] < -2 || (ids[k] == -2 && k != 0)) return nullptr;
    }
    if (desc->envMapTexId >= desc->numTextures) return nullptr;
    orc_scene *s = new orc_scene;
    s->d = *desc;
    memset(&s->st, 0, sizeof(s->st));
    return s;
}
void orc_scene_destroy(orc_scene *s) { delete s; }
void orc_stats_reset(orc_scene *s) { memset(&s->st, 0, sizeof(s->st)); }
void orc_debug_visit_hist(orc_scene *s, uint32_t *visitHist, uint64_t *rayLenHist) {
    s->visitHist = visitHist;
    s->rayLenHist = rayLenHist;
}
void orc_stats_get(const orc_scene *s, orc_stats *out) { *out = s->st; }

void orc_trace_closest(orc_scene *s, const float *rays, int64_t n, orc_hit *hits) {
    SceneView sv{s};
    for (int64_t i = 0; i < n; i++) {
        Ray r{vec3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), vec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5])};
        Intersection isec;
        sv.intersect(r, isec, &hits[i]);
    }
}
void orc_trace_closest_naive(orc_scene *s, const float *rays, int64_t n, orc_hit *hits) {
    SceneView sv{s};
    for (int64_t i = 0; i < n; i++) {
        Ray r{vec3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), vec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5])};
        sv.naiveIntersect(r, &hits[i]);
    }
}
void orc_trace_occluded(orc_scene *s, const float *seg, int64_t n, int32_t *occluded) {
    SceneView sv{s};
    for (int64_t i = 0; i < n; i++) {
        occluded[i] = sv.testOcclusion(vec3(seg[i * 6], seg[i * 6 + 1], seg[i * 6 + 2]),
                                       vec3(seg[i * 6 + 3], seg[i * 6 + 4], seg[i * 6 + 5]))
                          ? 1 : 0;
    }
}

void orc_path_trace(orc_scene *s, const void *camera196, float *directIllum, float *indirectIllum, int iter, int looper,
                    int maxDepth, int64_t pixBegin, int64_t pixEnd, int64_t pixStride) {
    Camera cam;
    memcpy(&cam, camera196, sizeof(cam));
    SceneView sv{s};
    float tanFovY = camTanFovY(cam);
    for (int64_t p = pixBegin; p < pixEnd; p += pixStride) {
        int x = int(p % cam.resx), y = int(p / cam.resx);
        pathTracePixel(sv, cam, tanFovY, x, y, looper, iter, maxDepth, directIllum, indirectIllum);
    }
}
void orc_path_trace_direct(orc_scene *s, const void *camera196, float *directIllum, int iter, int looper,
                           int64_t pixBegin, int64_t pixEnd, int64_t pixStride) {
    Camera cam;
    memcpy(&cam, camera196, sizeof(cam));
    SceneView sv{s};
    float tanFovY = camTanFovY(cam);
    for (int64_t p = pixBegin; p < pixEnd; p += pixStride) {
        int x = int(p % cam.resx), y = int(p / cam.resx);
        pathTraceDirectPixel(sv, cam, tanFovY, x, y, looper, iter, directIllum);
    }
}
void orc_gbuffer_render(orc_scene *s, const void *camera196, const void *lastCamera196, orc_gbuffer *gb) {
    Camera cam, lastCam;
    memcpy(&cam, camera196, sizeof(cam));
    memcpy(&lastCam, lastCamera196, sizeof(lastCam));
    SceneView sv{s};
    float tanFovY = camTanFovY(cam);
    for (int y = 0; y < cam.resy; y++)
        for (int x = 0; x < cam.resx; x++) gbufferPixel(sv, cam, lastCam, tanFovY, x, y, gb);
}

void orc_restir_direct(orc_scene *s, const void *camera196, float *directIllum, int iter, int looper,
                       void *reservoirOut_, const void *reservoirIn_, void *reservoirTemp_, const orc_gbuffer *gbuf,
                       int firstFrame, int reuseMask, int faithfulRIS, int numSpatial, int risCount) {
    Camera cam;
    memcpy(&cam, camera196, sizeof(cam));
    SceneView scene{s};
    GBView gb{gbuf};
    float tanFovY = camTanFovY(cam);
    DirectReservoir *reservoirOut = (DirectReservoir *)reservoirOut_;
    const DirectReservoir *reservoirIn = (const DirectReservoir *)reservoirIn_;
    DirectReservoir *reservoirTemp = (DirectReservoir *)reservoirTemp_;
    const int W = cam.resx, H = cam.resy;
    std::vector<RestirPixelState> state((size_t)W * H);

    // ---- pass 1: restir.cu:104-178 (up to the store that the reference follows with __syncthreads) ----
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int idx = x + y * W;
            RestirPixelState &ps = state[idx];
            ps.status = 1;
            ps.direct = vec3(0.f);
            ps.rng = makeSeededRandomEngine(looper, idx, 0, scene.d().sobol);
            Sampler &rng = ps.rng;
            Ray ray = cameraSample(cam, tanFovY, x, y, sample4D(rng));
            Intersection &intersec = ps.intersec;
            scene.intersect(ray, intersec);
            if (intersec.primId == NullPrimitive) {  // :117-122
                if (scene.hasEnvMap()) ps.direct = scene.envLookup(ray.direction);
                continue;
            }
            Material &material = ps.material;
            material = scene.getTexturedMaterialAndSurface(intersec);
            material.baseColor = vec3(1.f);  // :125
            if (material.type == Light) {
                ps.direct = material.baseColor;  // :127-130
                continue;
            }
            intersec.wo = -ray.direction;
            bool deltaBSDF = (material.type == Dielectric);
            if (!deltaBSDF && dot(intersec.norm, intersec.wo) < 0.f) intersec.norm = -intersec.norm;

            DirectReservoir reservoir;
            for (int i = 0; i < risCount; ++i) {  // RESERVOIR_SIZE = 32 (restir.h:9)
                vec3 Li(0.f), wi(0.f);  // zero-initialised: SURVEY Q19 defines the reference's UB
                float dist = 0.f;
                float lightPdf = scene.sampleDirectLightNoVisibility(intersec.pos, sample4D(rng), Li, wi, dist);
                vec3 bsdf = Li * materialBSDF(material, intersec.norm, intersec.wo, wi) * satDot(intersec.norm, wi);
                float weight = length(bsdf / lightPdf);
                if (isNanOrInf(weight) || lightPdf <= 0.f) weight = 0.f;
                resvUpdate(reservoir, LightLiSample{Li, wi, dist}, weight, sample1D(rng), faithfulRIS != 0);
            }
            LightLiSample sample = reservoir.sample;
            if (scene.testOcclusion(intersec.pos, intersec.pos + sample.wi * sample.dist)) reservoir.weight = 0.f;
            if (!firstFrame && (reuseMask & 1)) {
                DirectReservoir temporal = findTemporalNeighbor(reservoirIn, idx, gb);
                if (!resvInvalid(temporal)) resvPreClampedMerge(reservoir, temporal, sample1D(rng), 20);
            }
            DirectReservoir tempReservoir = reservoir;
            if (reuseMask & 2) {
                resvCheckValidity(reservoir);
                reservoirTemp[idx] = reservoir;
            }
            resvCheckValidity(tempReservoir);
            reservoirOut[idx] = tempReservoir;  // :186-187 (nobody reads reservoirOut within the frame)
            ps.reservoir = reservoir;
            ps.status = 0;
        }

    // ---- pass 2: restir.cu:180-203 ----
    const vec3 *albedo = (const vec3 *)gbuf->albedo;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int idx = x + y * W;
            RestirPixelState &ps = state[idx];
            vec3 direct = ps.direct;
            if (ps.status == 0) {
                DirectReservoir &reservoir = ps.reservoir;
                Sampler &rng = ps.rng;
                if (reuseMask & 2) {
                    DirectReservoir resvr;  // mergeSpatialNeighborDirect (:82-95)
                    for (int i = 0; i < numSpatial; i++) {
                        DirectReservoir spatial = findSpatialNeighborDisk(reservoirTemp, x, y, gb, sample2D(rng));
                        if (!resvInvalid(spatial)) resvMerge(resvr, spatial, sample1D(rng));
                    }
                    if (!resvInvalid(resvr) && !resvInvalid(reservoir)) resvMerge(reservoir, resvr, sample1D(rng));
                }
                LightLiSample sample = reservoir.sample;
                direct = vec3(0.f);
                if (!resvInvalid(reservoir)) {
                    direct = sample.Li * materialBSDF(ps.material, ps.intersec.norm, ps.intersec.wo, sample.wi) *
                             satDot(ps.intersec.norm, sample.wi) * resvW(reservoir, ps.intersec, ps.material);
                }
                if (hasNanOrInf(direct)) direct = vec3(0.f);
            }
            direct *= albedo[idx];
            storeRunningMean(directIllum, idx, direct, iter);
        }
}

uint32_t orc_utilhash(uint32_t a) { return utilhash(a); }
int orc_aabb_intersect(const float *b, const float *r, float *tMin) {
    AABB box{vec3(b[0], b[1], b[2]), vec3(b[3], b[4], b[5])};
    Ray ray{vec3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5])};
    float t = 0.f;
    bool hit = aabbIntersect(box, ray, t);
    *tMin = t;
    return hit ? 1 : 0;
}
int orc_intersect_triangle(const float *r, const float *v, float *bary2, float *dist) {
    Ray ray{vec3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5])};
    vec2 bary;
    float d = 0.f;
    bool hit = intersectTriangle(ray, vec3(v[0], v[1], v[2]), vec3(v[3], v[4], v[5]), vec3(v[6], v[7], v[8]), bary, d);
    bary2[0] = bary.x;
    bary2[1] = bary.y;
    *dist = d;
    return hit ? 1 : 0;
}
void orc_sincos(float x, float *s, float *c) { sincos_det(x, s, c); }
float orc_atan2(float y, float x) { return atan2_det(y, x); }
void orc_texture_sample(orc_scene *s, int texId, const float *uv2, float *rgb3) {
    SceneView sv{s};
    vec3 c = texId == ProceduralTexId ? proceduralTexture(vec2(uv2[0], uv2[1])) : sv.texSample(texId, vec2(uv2[0], uv2[1]));
    rgb3[0] = c.x; rgb3[1] = c.y; rgb3[2] = c.z;
}
void orc_material_eval(const void *material44, int which, const float *n3, const float *wo3, const float *w3,
                       float *out8) {
    Material m;
    memcpy(&m, material44, sizeof(m));
    vec3 n(n3[0], n3[1], n3[2]), wo(wo3[0], wo3[1], wo3[2]), w(w3[0], w3[1], w3[2]);
    memset(out8, 0, 8 * sizeof(float));
    if (which == 0) {
        vec3 f = materialBSDF(m, n, wo, w);
        out8[0] = f.x; out8[1] = f.y; out8[2] = f.z;
    } else if (which == 1) {
        out8[0] = materialPdf(m, n, wo, w);
    } else {
        BSDFSample smp;
        smp.dir = vec3(0.f); smp.bsdf = vec3(0.f); smp.pdf = 0.f; smp.type = 0;
        materialSample(m, n, wo, w, smp);
        out8[0] = smp.dir.x; out8[1] = smp.dir.y; out8[2] = smp.dir.z;
        out8[3] = smp.bsdf.x; out8[4] = smp.bsdf.y; out8[5] = smp.bsdf.z;
        out8[6] = smp.pdf;
        memcpy(&out8[7], &smp.type, 4);
    }
}
// sendImageToPBO's four overloads (pathtrace.cu:32-118) on host memory.  kind 0 vec3 (+ tone mapping, scale), 1 vec2,
// 2 float, 3 int pixel index.  Float → int as on the reference's GPU: NaN → 0, saturating.
static uint32_t displayByte(float c) {
    float v = c * 255.f;
    if (!(v > 0.f)) return 0u;
    if (v >= 255.f) return 255u;
    return (uint32_t)(int)v;
}
static float displayFilmic1(float c) {  // mathUtil.h:110-113
    return (c * (c * 0.22f + 0.03f) + 0.002f) / (c * (c * 0.22f + 0.3f) + 0.06f) - 1.f / 30.f;
}
void orc_copy_image_to_pbo(uint8_t *pbo, const void *image, int width, int height, int kind, int toneMapping, float scale) {
    const long long n = (long long)width * height;
    for (long long idx = 0; idx < n; idx++) {
        float c[3];
        if (kind == 0) {
            const float *img = (const float *)image;
            for (int k = 0; k < 3; k++) c[k] = img[3 * idx + k] * scale;
            if (toneMapping == 1) {  // Math::filmic (mathUtil.h:114-116)
                float d = displayFilmic1(11.2f);
                for (int k = 0; k < 3; k++) c[k] = displayFilmic1(c[k] * 1.6f) / d;
            } else if (toneMapping == 2) {  // Math::ACES (mathUtil.h:118-121)
                for (int k = 0; k < 3; k++) c[k] = (c[k] * (2.51f * c[k] + 0.03f)) / (c[k] * (2.43f * c[k] + 0.59f) + 0.14f);
            }
        } else if (kind == 1) {
            const float *img = (const float *)image;
            c[0] = img[2 * idx];
            c[1] = img[2 * idx + 1];
            c[2] = 0.f;
        } else if (kind == 2) {
            c[0] = c[1] = c[2] = ((const float *)image)[idx];
        } else {
            int v = ((const int32_t *)image)[idx];
            int px = v % width, py = v / width;
            c[0] = float(px) / float(width);
            c[1] = float(py) / float(height);
            c[2] = 0.f;
        }
        for (int k = 0; k < 3; k++) pbo[4 * idx + k] = (uint8_t)displayByte(om::pow_gamma_det(c[k]));
        pbo[4 * idx + 3] = 0;
    }
}
float orc_pow_gamma(float x) { return om::pow_gamma_det(x); }
float orc_exp(float x) { return om::exp_det(x); }
float orc_pow(float x, float y) { return om::pow_det(x, y); }

// ---- denoisers (denoiser.cu), on host memory ----------------------------------------------------------------------------
namespace {
const float kG3[3][3] = {{.075f, .124f, .075f}, {.124f, .204f, .124f}, {.075f, .124f, .075f}};
const float kG5[5][5] = {{.0030f, .0133f, .0219f, .0133f, .0030f},
                         {.0133f, .0596f, .0983f, .0596f, .0133f},
                         {.0219f, .0983f, .1621f, .0983f, .0219f},
                         {.0133f, .0596f, .0983f, .0596f, .0133f},
                         {.0030f, .0133f, .0219f, .0133f, .0030f}};
inline vec3 ld3(const float *p, int i) { return vec3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
inline void st3(float *p, int i, vec3 v) {
    p[3 * i] = v.x;
    p[3 * i + 1] = v.y;
    p[3 * i + 2] = v.z;
}
vec3 camGetPosition(const Camera &cam, int x, int y, float dist) {  // sceneStructs.h:50-70
    float aspect = float(cam.resx) / cam.resy;
    float tanFovY = camTanFovY(cam);
    vec2 pixelSize = vec2(1.f / float(cam.resx), 1.f / float(cam.resy));
    vec2 scr = vec2(float(x), float(y)) * pixelSize;
    vec2 ruv = scr + pixelSize * .5f;
    ruv = vec2(1.f - ruv.x * 2.f, 1.f - ruv.y * 2.f);
    vec3 pLens = vec3(0.f);
    vec2 f = ruv * vec2(aspect, 1.f) * tanFovY;
    vec3 pFocus = vec3(f.x, f.y, 1.f) * cam.focalDist;
    vec3 dir = pFocus - pLens;
    dir = normalize(mat3{cam.right, cam.up, cam.view} * dir);
    vec3 ori = cam.position + cam.right * pLens.x + cam.up * pLens.y;
    return ori + dir * dist;
}
}  // namespace

void orc_denoise_eaw(float *colorOut, const float *colorIn, const orc_gbuffer *gb, const void *camera196, float sigLumin,
                     float sigNormal, float sigDepth, int level) {  // denoiser.cu:17-84
    Camera cam;
    memcpy(&cam, camera196, sizeof(cam));
    const int f = gb->frameIdx, step = 1 << level;
    const int *primId = gb->primId[f];
    const float *normal = gb->normal[f], *depth = gb->depth[f];
    for (int y = 0; y < cam.resy; y++)
        for (int x = 0; x < cam.resx; x++) {
            int idxP = x + y * cam.resx;
            int primIdP = primId[idxP];
            if (primIdP <= NullPrimitive) {
                st3(colorOut, idxP, ld3(colorIn, idxP));
                continue;
            }
            vec3 colorP = ld3(colorIn, idxP), normalP = ld3(normal, idxP);
            vec3 posP = camGetPosition(cam, x, y, depth[idxP]);
            vec3 sum(0.f);
            float weightSum = 0.f;
            for (int i = -2; i <= 2; i++)
                for (int j = -2; j <= 2; j++) {
                    int qx = x + j * step, qy = y + i * step;
                    if (qx >= cam.resx || qy >= cam.resy || qx < 0 || qy < 0) continue;
                    int idxQ = qx + qy * cam.resx;
                    if (primId[idxQ] != primIdP) continue;
                    vec3 normalQ = ld3(normal, idxQ);
                    vec3 posQ = camGetPosition(cam, qx, qy, depth[idxQ]);
                    vec3 colorQ = ld3(colorIn, idxQ);
                    float wColor = gmin(1.f, exp_det(-dot(colorP - colorQ, colorP - colorQ) / sigLumin));
                    float wNormal = gmin(1.f, exp_det(-dot(normalP - normalQ, normalP - normalQ) / sigNormal));
                    float wPos = gmin(1.f, exp_det(-dot(posP - posQ, posP - posQ) / sigDepth));
                    float weight = wColor * wNormal * wPos * kG5[i + 2][j + 2];
                    sum = sum + colorQ * weight;
                    weightSum += weight;
                }
            st3(colorOut, idxP, (weightSum == 0.f) ? ld3(colorIn, idxP) : sum / weightSum);
        }
}

void orc_denoise_svgf(float *colorOut, const float *colorIn, float *varianceOut, const float *varianceIn, const float *varFiltered,
                      const orc_gbuffer *gb, const void *camera196, float sigLumin, float sigNormal, float sigDepth,
                      int level) {  // denoiser.cu:92-173
    Camera cam;
    memcpy(&cam, camera196, sizeof(cam));
    const int f = gb->frameIdx, step = 1 << level;
    const int *primId = gb->primId[f];
    const float *normal = gb->normal[f], *depth = gb->depth[f];
    for (int y = 0; y < cam.resy; y++)
        for (int x = 0; x < cam.resx; x++) {
            int idxP = x + y * cam.resx;
            if (primId[idxP] <= NullPrimitive) {
                st3(colorOut, idxP, ld3(colorIn, idxP));
                varianceOut[idxP] = varianceIn[idxP];
                continue;
            }
            vec3 colorP = ld3(colorIn, idxP), normalP = ld3(normal, idxP);
            vec3 posP = camGetPosition(cam, x, y, depth[idxP]);
            vec3 colorSum(0.f);
            float varianceSum = 0.f, weightSum = 0.f, weight2Sum = 0.f;
            for (int i = -2; i <= 2; i++)
                for (int j = -2; j <= 2; j++) {
                    int qx = x + j * step, qy = y + i * step;
                    if (qx >= cam.resx || qy >= cam.resy || qx < 0 || qy < 0) continue;
                    int idxQ = qx + qy * cam.resx;
                    vec3 normalQ = ld3(normal, idxQ);
                    vec3 posQ = camGetPosition(cam, qx, qy, depth[idxQ]);
                    float varQ = varianceIn[idxQ];
                    vec3 colorQ = ld3(colorIn, idxQ);
                    float wPos = exp_det(-dot(posP - posQ, posP - posQ) / (sigDepth + 1e-4f));
                    float wNormal = pow_det(satDot(normalP, normalQ), sigNormal) + 1e-4f;
                    float denom = sigLumin * sqrtf(gmax(varFiltered[idxP], 0.f)) + 1e-4f;
                    float wColor = exp_det(-fabsf(luminance(colorP) - luminance(colorQ)) / denom) + 1e-4f;
                    float weight = wColor * wNormal * wPos * kG5[i + 2][j + 2];
                    float weight2 = weight * weight;
                    colorSum = colorSum + colorQ * weight;
                    varianceSum += varQ * weight2;
                    weightSum += weight;
                    weight2Sum += weight2;
                }
            st3(colorOut, idxP, (weightSum < FLT_EPSILON) ? ld3(colorIn, idxP) : colorSum / weightSum);
            varianceOut[idxP] = (weight2Sum < FLT_EPSILON) ? varianceIn[idxP] : varianceSum / weight2Sum;
        }
}

void orc_denoise_modulate(float *image, const orc_gbuffer *gb) {  // denoiser.cu:175-185; LDRToHDR = identity (mathUtil.h:53-56)
    for (int i = 0; i < gb->width * gb->height; i++) {
        vec3 c = ld3(image, i) / 1.f, a = ld3(gb->albedo, i);
        st3(image, i, c * vec3(gmax(a.x, 0.f), gmax(a.y, 0.f), gmax(a.z, 0.f)));
    }
}
void orc_denoise_add(float *out, const float *in1, const float *in2, int width, int height) {  // denoiser.cu:187-206
    for (long long i = 0; i < 3ll * width * height; i++) out[i] = in1[i] + in2[i];
}

void orc_denoise_temporal_accumulate(float *colorAccumOut, const float *colorAccumIn, float *momentAccumOut,
                                     const float *momentAccumIn, const float *colorIn, const orc_gbuffer *gb, int first) {
    const float alpha = 0.2f;  // denoiser.cu:208-262
    const int f = gb->frameIdx;
    for (int idx = 0; idx < gb->width * gb->height; idx++) {
        int primId = gb->primId[f][idx], lastIdx = gb->motion[idx];
        bool diff = first != 0;
        if (lastIdx < 0) diff = true;
        else if (primId <= NullPrimitive) diff = true;
        else if (gb->primId[f ^ 1][lastIdx] != primId) diff = true;
        else if (fabsf(dot(ld3(gb->normal[f], idx), ld3(gb->normal[f ^ 1], lastIdx))) < .1f) diff = true;
        vec3 color = ld3(colorIn, idx);
        float lum = luminance(color);
        vec3 colorAccum, momentAccum;
        if (diff) {
            colorAccum = color;
            momentAccum = vec3(lum, lum * lum, 0.f);
        } else {
            vec3 lastColor = ld3(colorAccumIn, lastIdx), lastMoment = ld3(momentAccumIn, lastIdx);
            colorAccum = lastColor * (1.f - alpha) + color * alpha;
            momentAccum = vec3(lastMoment.x * (1.f - alpha) + lum * alpha, lastMoment.y * (1.f - alpha) + (lum * lum) * alpha,
                               lastMoment.z + 1.f);
        }
        st3(colorAccumOut, idx, colorAccum);
        st3(momentAccumOut, idx, momentAccum);
    }
}

void orc_denoise_estimate_variance(float *variance, const float *moment, int width, int height) {  // denoiser.cu:264-299
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            int idx = x + y * width;
            vec3 m = ld3(moment, idx);
            if (m.z > 3.5f) {
                variance[idx] = m.y - m.x * m.x;
            } else {
                float sx = 0.f, sy = 0.f;
                int pixelCount = 0;
                for (int i = -1; i <= 1; i++)
                    for (int j = -1; j <= 1; j++) {
                        int qx = x + j, qy = y + i;
                        if (qx < 0 || qx >= width || qy < 0 || qy >= height) continue;
                        vec3 q = ld3(moment, qx + qy * width);
                        sx += q.x;
                        sy += q.y;
                        pixelCount++;
                    }
                sx = sx / float(pixelCount);
                sy = sy / float(pixelCount);
                variance[idx] = sy - sx * sx;
            }
        }
}

void orc_denoise_filter_variance(float *varianceOut, const float *varianceIn, int width, int height) {  // denoiser.cu:301-328
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            float sum = 0.f, weightSum = 0.f;
            for (int i = -1; i <= 1; i++)
                for (int j = -1; j <= 1; j++) {
                    int qx = x + i, qy = y + j;
                    if (qx < 0 || qx >= width || qy < 0 || qy >= height) continue;
                    float weight = kG3[i + 1][j + 1];
                    sum += varianceIn[qx + qy * width] * weight;
                    weightSum += weight;
                }
            varianceOut[x + y * width] = sum / weightSum;
        }
}

void orc_camera_sample(const void *camera196, int x, int y, const float *r4, float *ray6) {
    Camera cam;
    memcpy(&cam, camera196, sizeof(cam));
    Ray r = cameraSample(cam, camTanFovY(cam), x, y, vec4{r4[0], r4[1], r4[2], r4[3]});
    ray6[0] = r.origin.x; ray6[1] = r.origin.y; ray6[2] = r.origin.z;
    ray6[3] = r.direction.x; ray6[4] = r.direction.y; ray6[5] = r.direction.z;
}

// ---- known-answer-test hooks for the light sampler and the reservoir arithmetic (tests/test_oracle_float64.py) ------------
void orc_sample_direct_light(orc_scene *s, const float *pos3, const float *r4, int visibility, float *radiance3, float *wi3,
                             float *dist, float *pdf) {
    SceneView scene{s};
    vec3 radiance(0.f), wi(0.f);
    float d = 0.f;
    vec3 pos(pos3[0], pos3[1], pos3[2]);
    vec4 r{r4[0], r4[1], r4[2], r4[3]};
    *pdf = visibility ? scene.sampleDirectLight(pos, r, radiance, wi) : scene.sampleDirectLightNoVisibility(pos, r, radiance, wi, d);
    radiance3[0] = radiance.x; radiance3[1] = radiance.y; radiance3[2] = radiance.z;
    wi3[0] = wi.x; wi3[1] = wi.y; wi3[2] = wi.z;
    *dist = d;
}
void orc_reservoir_op(int op, void *r36, const void *rhs36, float rnd, int M) {
    DirectReservoir r, rhs;
    memcpy(&r, r36, sizeof(r));
    memcpy(&rhs, rhs36, sizeof(rhs));
    switch (op) {
        case 0: resvMerge(r, rhs, rnd); break;
        case 1: resvPreClampedMerge(r, rhs, rnd, M); break;
        case 2: resvUpdate(r, rhs.sample, rhs.weight, rnd, true); break;   // rhs carries {newSample, newWeight}
        case 3: resvUpdate(r, rhs.sample, rhs.weight, rnd, false); break;
        case 4: resvCheckValidity(r); break;
        default: break;
    }
    memcpy(r36, &r, sizeof(r));
}
float orc_reservoir_W(const void *r36, const void *material44, const float *n3, const float *wo3) {
    DirectReservoir r;
    Material m;
    memcpy(&r, r36, sizeof(r));
    memcpy(&m, material44, sizeof(m));
    Intersection isec{};
    isec.norm = vec3(n3[0], n3[1], n3[2]);
    isec.wo = vec3(wo3[0], wo3[1], wo3[2]);
    return resvW(r, isec, m);
}
float orc_power_heuristic(float f, float g) { return powerHeuristic(f, g); }

}  // extern "C"
