// radish_pt_amd/csrc/device/bsdf.h — sampler, camera, sampling helpers and the three BSDFs.
//
// Follows /root/reference/src/sampler.h:15-64, sceneStructs.h:21-91, mathUtil.h:49-192 and material.h:18-287
// expression by expression (operation order matters: see the numerics contract in rmath.h).
#pragma once
#include "layouts.h"
#include "traverse.h"

namespace rd {

constexpr float PI_F = 3.1415926535897932384626422832795028841971f;      // mathUtil.h:15
constexpr float TWO_PI_F = 6.2831853071795864769252867665590057683943f;  // mathUtil.h:16
// INV_PI is the unparenthesised macro `1.f / PI` (mathUtil.h:17): `x * INV_PI` == (x * 1.f) / PI == x / PI.
constexpr float INVALID_PDF = -1.f;  // material.h:16

enum MaterialType { Lambertian = 0, MetallicWorkflow = 1, Dielectric = 2, Disney = 3, Light = 4 };  // material.h:129
enum BSDFSampleType : uint32_t {  // material.h:18-26
    Diffuse = 1 << 0, Glossy = 1 << 1, Specular = 1 << 2, Reflection = 1 << 4, Transmission = 1 << 5, Invalid = 1 << 15
};

// ---- Sampler (sampler.h:15-37): Sobol row XOR per-pixel hash chain.  8 bytes of state ride with each path. ----
struct Sampler {
    const uint32_t *data;
    uint32_t scramble;
    int ptr;
    RD_DEV float sample() {
        uint32_t r = data[ptr++] ^ scramble;
        scramble = utilhash(scramble);
        return float(r) * 0x1p-32f;
    }
};
RD_DEV Sampler makeSeededRandomEngine(int iter, int index, int dim, const uint32_t *data) {  // sampler.h:32-35
    return Sampler{data, utilhash((uint32_t)index), iter * 200 + dim};
}
// Draw order: left to right (x first) — see oracle.cpp's note on the unspecified order in the reference.
RD_DEV v2 sample2D(Sampler &s) { float a = s.sample(); float b = s.sample(); return {a, b}; }
RD_DEV v3 sample3D(Sampler &s) { float a = s.sample(); float b = s.sample(); float c = s.sample(); return {a, b, c}; }
RD_DEV v4 sample4D(Sampler &s) {
    float a = s.sample(); float b = s.sample(); float c = s.sample(); float d = s.sample();
    return {a, b, c, d};
}

// ---- Camera::sample (sceneStructs.h:72-91) ----
RD_DEV Ray cameraSample(const DCamera &c, int x, int y, v4 r) {
    float aspect = float(c.resx) / float(c.resy);
    v2 pixelSize = {1.f / float(c.resx), 1.f / float(c.resy)};
    v2 scr = mk2(float(x), float(y)) * pixelSize;
    v2 ruv = scr + pixelSize * mk2(r.x, r.y);
    ruv = {1.f - ruv.x * 2.f, 1.f - ruv.y * 2.f};
    v2 pAperture = {0.f, 0.f};
    v3 pLens = {pAperture.x * c.lensRadius, pAperture.y * c.lensRadius, 0.f};
    v2 f = (ruv * mk2(aspect, 1.f)) * c.tanFovY;
    v3 pFocus = mk3(f.x, f.y, 1.f) * c.focalDist;
    v3 dir = pFocus - pLens;
    Ray ray;
    ray.d = normalize(mul(m3{c.right, c.up, c.view}, dir));
    ray.o = c.position + c.right * pLens.x + c.up * pLens.y;
    return ray;
}
// Camera::getRasterUV (sceneStructs.h:22-43); tanFovY = tan(radians(fov.y)) of THAT camera.
RD_DEV v2 cameraRasterUV(const DCamera &c, v3 pos) {
    v3 dir = normalize(pos - c.position);
    float d = 1.f / dot(dir, c.view);
    v3 p = mul(c.rotationMatInv, dir * d);
    float aspect = float(c.resx) / float(c.resy);
    v2 sc = mk2(aspect, 1.f) * c.tanFovY;
    p = p / mk3(sc.x, sc.y, 1.f);
    v2 ndc = {-p.x, -p.y};
    return {ndc.x * .5f + .5f, ndc.y * .5f + .5f};
}

// ---- mathUtil.h helpers ----
RD_DEV v3 HDRToLDR(v3 c) { return c / (c + 1.f) * 1.f; }  // :49-51
RD_DEV bool hasNanOrInf(v3 v) { return isNanOrInf(v.x) || isNanOrInf(v.y) || isNanOrInf(v.z); }  // :62-65
RD_DEV float satDot(v3 a, v3 b) { return gmax(dot(a, b), 0.f); }  // :67-69
RD_DEV float absDot(v3 a, v3 b) { return fabs_(dot(a, b)); }      // :71-73
RD_DEV float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }  // :74-77
RD_DEV float powerHeuristic(float f, float g) { float f2 = f * f; return f2 / (f2 + g * g); }  // :81-84
RD_DEV float triangleArea(v3 v0, v3 v1, v3 vc) { return length(cross(v1 - v0, vc - v0)) * 0.5f; }  // :90-93
RD_DEV v3 triangleNormal(v3 v0, v3 v1, v3 vc) { return normalize(cross(v1 - v0, vc - v0)); }     // :95-98
RD_DEV v3 sampleTriangleUniform(v3 v0, v3 v1, v3 vc, float ru, float rv) {  // :100-108
    float r = __builtin_sqrtf(rv);
    float u = 1.f - r;
    float v = ru * r;
    return v1 * u + vc * v + v0 * (1.f - u - v);
}
RD_DEV float luminance(v3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }  // :128-130
RD_DEV v2 concentricSampleDisk(float x, float y) {  // :132-136
    float r = __builtin_sqrtf(x);
    float theta = TWO_PI_F * y;
    float s, c;
    sincos_det(theta, s, c);
    return {r * c, r * s};
}
RD_DEV m3 localRefMatrix(v3 n) {  // :149-155
    v3 t = (fabs_(n.y) > 0.9999f) ? mk3(0.f, 0.f, 1.f) : mk3(0.f, 1.f, 0.f);
    v3 b = normalize(cross(n, t));
    t = cross(b, n);
    return m3{t, b, n};
}
RD_DEV v3 localToWorld(v3 n, v3 v) { return normalize(mul(localRefMatrix(n), v)); }  // :157-159
RD_DEV v3 cosineSampleHemisphere(v3 n, float rx, float ry) {  // :161-166
    v2 d = concentricSampleDisk(rx, ry);
    float z = __builtin_sqrtf(1.f - dot(d, d));
    return localToWorld(n, mk3(d.x, d.y, z));
}
RD_DEV bool refract(v3 n, v3 wi, float ior, v3 &wt) {  // :168-186
    float cosIn = dot(n, wi);
    if (cosIn < 0) ior = 1.f / ior;
    float sin2In = gmax(0.f, 1.f - cosIn * cosIn);
    float sin2Tr = sin2In / (ior * ior);
    if (sin2Tr >= 1.f) return false;
    float cosTr = __builtin_sqrtf(1.f - sin2Tr);
    if (cosIn < 0) cosTr = -cosTr;
    wt = normalize((-wi) / ior + n * (cosIn / ior - cosTr));
    return true;
}
RD_DEV float pdfAreaToSolidAngle(float pdf, v3 x, v3 y, v3 ny) {  // :188-192
    v3 yx = x - y;
    return pdf * dot(yx, yx) / absDot(ny, normalize(yx));
}
// The same value where the caller already holds v = y - x (= -yx exactly), d2 = dot(v, v) and wi = normalize(v) = v * (1 / sqrt(d2)):
// dot(yx, yx) is d2 bit for bit (products of equal magnitudes), normalize(yx) is -wi (a product by the same reciprocal) and
// dot(ny, -wi) is -dot(ny, wi) (rounding to nearest is symmetric), so one dot product, one square root and one reciprocal less.
RD_DEV float pdfAreaToSolidAngleFrom(float pdf, float d2, v3 ny, v3 wi) { return pdf * d2 / fabs_(dot(ny, wi)); }

// ---- material.h ----
struct Material {
    int type;
    v3 baseColor;
    float metallic, roughness, ior;
};
// The untextured record (scene->materials[id], e.g. gBuffer.cu:33-34).
RD_DEV Material loadMaterial(const MatRec *mats, int id) {
    float4 a = mats[id].a, b = mats[id].b;
    return Material{__float_as_int(a.x), mk3(a.y, a.z, a.w), b.x, b.y, b.z};
}

// ---- textures: image.h:42-87, scene.h:77-112; env map mapping: mathUtil.h:138-147 ----
RD_DEV v3 toSphere(v2 v) {  // mathUtil.h:138-142
    v = v * mk2(TWO_PI_F, PI_F);
    float sx, cx, sy, cy;
    sincos_det(v.x, sx, cx);
    sincos_det(v.y, sy, cy);
    return mk3(cx * sy, cy, sx * sy);
}
RD_DEV v2 toPlane(v3 v) {  // mathUtil.h:143-147; `x * INV_PI * 0.5f` == ((x * 1.f) / PI) * 0.5f
    float len = __builtin_sqrtf(dot(mk2(v.x, v.z), mk2(v.x, v.z)));
    return mk2(fract_(atan2_det(v.z, v.x) / PI_F * 0.5f + 1.f), atan2_det(len, v.y) / PI_F);
}
RD_DEV v3 texel(const float *data, int i) { return mk3(data[3 * i], data[3 * i + 1], data[3 * i + 2]); }
// linearSample (image.h:42-87): wrap-around bilinear
RD_DEV v3 linearSample(const float *data, v2 uv, int width, int height) {
    const float eps = 1.17549435e-38f;  // FLT_MIN
    uv = mk2(fract_(uv.x), fract_(uv.y));
    float fx = uv.x * (float(width) - eps) + 0.5f;
    float fy = uv.y * (float(height) - eps) + 0.5f;
    int ix = int(fract_(fx) > 0.5f ? fx : fx - 1.f);
    int iy = int(fract_(fy) > 0.5f ? fy : fy - 1.f);
    if (ix < 0) ix += width;
    if (iy < 0) iy += height;
    int ux = ix + 1;
    int uy = iy + 1;
    if (ux >= width) ux -= width;
    if (uy >= height) uy -= height;
    float lx = fract_(fx + 0.5f);
    float ly = fract_(fy + 0.5f);
    v3 c1 = mix(texel(data, iy * width + ix), texel(data, iy * width + ux), lx);
    v3 c2 = mix(texel(data, uy * width + ix), texel(data, uy * width + ux), lx);
    return mix(c1, c2, ly);
}
RD_DEV v3 texSample(const DScene &s, int id, v2 uv) {  // DevTextureObj::linearSample
    int4 ti = s.texInfo[id];
    return linearSample(s.texData + 3 * (long long)ti.z, uv, ti.x, ti.y);
}
// proceduralTexture (scene.h:77-86): minstd_rand + thrust's uniform_real_distribution<float>, restated (see oracle.cpp)
RD_DEV float minstdUniform(uint32_t &x) {
    x = uint32_t(((unsigned long long)x * 48271ull) % 2147483647ull);
    return float(x - 1u) / (1.f + float(2147483646u - 1u));
}
RD_DEV v3 proceduralTexture(v2 uv) {
    uint32_t seed = uint32_t(int(uv.x * 1024) * 1024 + int(uv.y * 1024));
    uint32_t x = seed % 2147483647u;
    if (x == 0u) x = 1u;
    float rx = minstdUniform(x);
    float ry = minstdUniform(x);
    float s0, c0, s1, c1;
    sincos_det(uv.x * 10.f * TWO_PI_F + rx * TWO_PI_F, s0, c0);
    sincos_det(uv.y * 10.f * TWO_PI_F + ry * TWO_PI_F, s1, c1);
    float f = (s0 + +1.f) * .5f;
    float g = (s1 + +1.f) * .5f;
    return mk3(f * g);
}
struct BSDFSample {  // material.h:28-33
    v3 dir, bsdf;
    float pdf;
    uint32_t type;
};

RD_DEV v3 fresnelSchlick(float lDotH, v3 f0) { return mix(f0, mk3(1.f), pow5(1.f - lDotH)); }  // :40-42
RD_DEV float fresnel(float cosIn, float ior) {  // :44-64, exact branch
    if (cosIn < 0.f) {
        ior = 1.f / ior;
        cosIn = -cosIn;
    }
    float sinIn = __builtin_sqrtf(1.f - cosIn * cosIn);
    float sinTr = sinIn / ior;
    if (sinTr >= 1.f) return 1.f;
    float cosTr = __builtin_sqrtf(1.f - sinTr * sinTr);
    float rPar = (cosIn - ior * cosTr) / (cosIn + ior * cosTr);
    float rPer = (ior * cosIn - cosTr) / (ior * cosIn + cosTr);
    return (rPar * rPar + rPer * rPer) * .5f;
}
RD_DEV float schlickG(float cosTheta, float alpha) {  // :68-71
    float a = alpha * .5f;
    return cosTheta / (cosTheta * (1.f - a) + a);
}
RD_DEV float smithG(float cosWo, float cosWi, float alpha) {  // :74-76
    return schlickG(fabs_(cosWo), alpha) * schlickG(fabs_(cosWi), alpha);
}
RD_DEV float ggxDistribution(float cosTheta, float alpha) {  // :79-88
    if (cosTheta < 1e-6f) return 0.f;
    float alpha2 = alpha * alpha;
    float denom = (cosTheta * cosTheta) * (alpha2 - 1.f) + 1.f;
    denom = denom * denom * PI_F;
    return alpha2 / denom;
}
RD_DEV float ggxPdf(v3 n, v3 m, v3 wo, float alpha) {  // :92-97
    return ggxDistribution(dot(n, m), alpha) * schlickG(dot(n, wo), alpha) * absDot(m, wo) / absDot(n, wo);
}
RD_DEV v3 ggxSample(v3 n, v3 wo, float alpha, v2 r) {  // :106-126
    m3 transMat = localRefMatrix(n);
    m3 transInv = inverse(transMat);
    v3 vh = normalize(mul(transInv, wo) * mk3(alpha, alpha, 1.f));
    float lenSq = vh.x * vh.x + vh.y * vh.y;
    v3 t = lenSq > 0.f ? mk3(-vh.y, vh.x, 0.f) / __builtin_sqrtf(lenSq) : mk3(1.f, 0.f, 0.f);
    v3 b = cross(vh, t);
    v2 p = concentricSampleDisk(r.x, r.y);
    float s = 0.5f * (vh.z + 1.f);
    p.y = (1.f - s) * __builtin_sqrtf(1.f - p.x * p.x) + s * p.y;
    v3 h = t * p.x + b * p.y + vh * __builtin_sqrtf(gmax(0.f, 1.f - dot(p, p)));
    h = mk3(h.x * alpha, h.y * alpha, gmax(0.f, h.z));
    return normalize(mul(transMat, h));
}

RD_DEV v3 metallicBSDF(const Material &m, v3 n, v3 wo, v3 wi) {  // :187-205
    float alpha = m.roughness * m.roughness;
    v3 h = normalize(wo + wi);
    float cosO = dot(n, wo);
    float cosI = dot(n, wi);
    if (cosI * cosO < 1e-7f) return mk3(0.f);
    v3 f = fresnelSchlick(dot(h, wo), mix(mk3(.08f), m.baseColor, m.metallic));
    float d = ggxDistribution(dot(n, h), alpha);
    float g = smithG(cosO, cosI, alpha);
    return mix(m.baseColor / PI_F * (1.f - m.metallic), mk3(g * d / (4.f * cosI * cosO)), f);
}
RD_DEV float metallicPdf(const Material &m, v3 n, v3 wo, v3 wi) {  // :207-213
    v3 h = normalize(wo + wi);
    return mixf(satDot(n, wi) / PI_F, ggxPdf(n, h, wo, m.roughness * m.roughness) / (4.f * absDot(h, wo)),
                1.f / (2.f - m.metallic));
}

RD_DEV v3 materialBSDF(const Material &m, v3 n, v3 wo, v3 wi) {  // :235-246
    if (m.type == Lambertian) return m.baseColor / PI_F;        // :131-134
    if (m.type == MetallicWorkflow) return metallicBSDF(m, n, wo, wi);
    return mk3(0.f);                                            // Dielectric (:149-152), Light, Disney
}
RD_DEV float materialPdf(const Material &m, v3 n, v3 wo, v3 wi) {  // :248-258
    if (m.type == Lambertian) return satDot(n, wi) / PI_F;      // :136-139
    if (m.type == MetallicWorkflow) return metallicPdf(m, n, wo, wi);
    return 0.f;
}
RD_DEV void materialSample(const Material &m, v3 n, v3 wo, v3 r, BSDFSample &s) {  // :260-275
    if (m.type == Lambertian) {  // :141-147
        s.dir = cosineSampleHemisphere(n, r.x, r.y);
        s.bsdf = m.baseColor / PI_F;
        s.pdf = satDot(n, s.dir) / PI_F;
        s.type = Diffuse | Reflection;
    } else if (m.type == MetallicWorkflow) {  // :215-233
        float alpha = m.roughness * m.roughness;
        if (r.z > (1.f / (2.f - m.metallic))) {
            s.dir = cosineSampleHemisphere(n, r.x, r.y);
        } else {
            v3 h = ggxSample(n, wo, alpha, mk2(r.x, r.y));
            s.dir = -reflect(wo, h);
        }
        if (dot(n, s.dir) < 0.f) {
            s.type = Invalid;
        } else {
            s.type = Glossy | Reflection;
            s.pdf = metallicPdf(m, n, wo, s.dir);
            s.bsdf = metallicBSDF(m, n, wo, s.dir);
        }
    } else if (m.type == Dielectric) {  // :159-183
        float pdfRefl = fresnel(dot(n, wo), m.ior);
        s.bsdf = m.baseColor;
        if (r.z < pdfRefl) {
            s.dir = reflect(-wo, n);
            s.type = Specular | Reflection;
            s.pdf = 1.f;
        } else {
            bool ok = refract(n, wo, m.ior, s.dir);
            if (!ok) {
                s.type = Invalid;
                return;
            }
            float eta = m.ior;
            if (dot(n, wo) < 0) eta = 1.f / eta;
            s.type = Specular | Transmission;
            s.pdf = 1.f;
            s.bsdf = s.bsdf / (eta * eta);
        }
    } else {
        s.type = Invalid;
    }
}

}  // namespace rd
