// radish_pt_amd/csrc/device/lights.h — surface fetch at a hit and emissive-triangle sampling.
//
// Follows /root/reference/src/scene.h:139-165 (getPrimitiveArea, getIntersecGeomInfo), :419-492
// (sampleDirectLight[NoVisibility]) and /root/reference/src/sampler.h:204-208 (alias lookup).
#pragma once
#include "bsdf.h"

namespace rd {

struct Surface {  // the fields of Intersection the path uses (sceneStructs.h:163-190)
    int primId, matId;
    v3 pos, norm;
    v2 uv;
    v3 wo;
};

// getIntersecGeomInfo (scene.h:147-165) + materialIds[prim].  Reads the 48-B triangle record and the 64-B
// attribute record of the hit primitive (4 + 3 dwordx4 loads).
RD_DEV void fetchSurface(const DScene &s, int prim, v2 bary, Surface &o) {
    TriVerts t = loadTri(s.tris, prim);
    const AttrRec *ar = s.attrs + prim;
    float4 A = ar->a, B = ar->b, C = ar->c, D = ar->d;
    v3 na = mk3(A.x, A.y, A.z), nb = mk3(A.w, B.x, B.y), nc = mk3(B.z, B.w, C.x);
    v2 ta = mk2(C.y, C.z), tb = mk2(C.w, D.x), tc = mk2(D.y, D.z);
    float w = 1.f - bary.x - bary.y;
    o.primId = prim;
    o.matId = t.matId;
    o.pos = t.b * bary.x + t.c * bary.y + t.a * w;
    o.norm = normalize(nb * bary.x + nc * bary.y + na * w);
    o.uv = tb * bary.x + tc * bary.y + ta * w;
}

// getTexturedMaterialAndSurface (scene.h:88-112): may replace baseColor / metallic / roughness by texture fetches and
// perturb `isec.norm` by the normal map (`normalize(mapped * 1 - 0.5)`, sic — SURVEY Q16).
RD_DEV Material texturedMaterial(const DScene &s, Surface &isec) {
    Material mat = loadMaterial(s.mats, isec.matId);
    int4 maps = s.mats[isec.matId].maps;
    if ((maps.x & maps.y & maps.z & maps.w) == -1) return mat;  // all four ids are NullTextureId
    if (maps.x != -1) mat.baseColor = (maps.x == -2) ? proceduralTexture(isec.uv) : texSample(s, maps.x, isec.uv);
    if (maps.z > -1) mat.metallic = texSample(s, maps.z, isec.uv).x;
    if (maps.w > -1) mat.roughness = texSample(s, maps.w, isec.uv).x;
    if (maps.y != -1) {
        v3 mapped = texSample(s, maps.y, isec.uv);
        v3 localNorm = normalize(mapped * 1.f + (-0.5f));
        isec.norm = localToWorld(isec.norm, localNorm);
    }
    return mat;
}

RD_DEV bool hasEnvMap(const DScene &s) { return s.envTex >= 0; }
RD_DEV v3 envLookup(const DScene &s, v3 dir) { return texSample(s, s.envTex, toPlane(dir)); }
RD_DEV float environmentMapPdf(const DScene &s, v3 wi) {  // scene.h:374-378
    int4 ti = s.texInfo[s.envTex];
    v3 radiance = envLookup(s, wi);
    return luminance(radiance) * s.sumLightPowerInv * ti.x * ti.y * 0.5f;
}

RD_DEV float getPrimitiveArea(const DScene &s, int prim) {  // scene.h:139-145
    TriVerts t = loadTri(s.tris, prim);
    return triangleArea(t.a, t.b, t.c);
}

RD_DEV int lightAliasSample(const DScene &s, float r1, float r2) {  // DevDiscreteSampler1D::sample, sampler.h:204-208
    int length = s.lightSamplerLength;
    int passId = imin(int(float(length) * r1), length - 1);
    AliasRec d = s.lightAlias[passId];
    return (r2 < d.prob) ? passId : d.failId;
}

struct LightPick {
    v3 sampled, normal, radiance;
    float area;
    bool isEnv;   // the environment map was picked: `sampled` = pos + wi*1e6, `normal` holds wi, `area` the pdf
};
// Alias lookup + point on the chosen light (scene.h:423-433); the env map is the last entry (scene.h:426-428, :380-399).
RD_DEV LightPick pickLightPoint(const DScene &s, v3 pos, v4 r) {
    int lightId = lightAliasSample(s, r.x, r.y);
    if (lightId == s.lightSamplerLength - 1 && s.envSamplerLength != 0) {
        int4 ti = s.texInfo[s.envTex];
        int length = s.envSamplerLength;
        int passId = imin(int(float(length) * r.z), length - 1);
        AliasRec d = s.envAlias[passId];
        int pixId = (r.w < d.prob) ? passId : d.failId;
        int y = pixId / ti.x;
        int x = pixId - y * ti.x;
        LightPick p;
        p.radiance = texel(s.texData + 3 * (long long)ti.z, pixId);
        v3 wi = toSphere(mk2((x + 0.5f) / ti.x, (y + 0.5f) / ti.y));
        p.normal = wi;
        p.sampled = pos + wi * 1e6f;
        p.area = luminance(p.radiance) * s.sumLightPowerInv * ti.x * ti.y / PI_F / PI_F * 0.5f;
        p.isEnv = true;
        return p;
    }
    const LightRec *l = s.lights + lightId;
    float4 A = l->a, B = l->b, C = l->c;
    v3 v0 = mk3(A.x, A.y, A.z), v1 = mk3(A.w, B.x, B.y), v2_ = mk3(B.z, B.w, C.x);
    LightPick p;
    p.sampled = sampleTriangleUniform(v0, v1, v2_, r.z, r.w);
    p.normal = triangleNormal(v0, v1, v2_);
    p.area = triangleArea(v0, v1, v2_);
    p.radiance = mk3(C.y, C.z, C.w);
    p.isEnv = false;
    return p;
}

// Everything of sampleDirectLight (scene.h:419-456) except the shadow ray, which the caller traces — in the
// megakernel immediately (reference order: occlusion first, then the single-sided test), in the
// wavefront pipeline through the shadow queue.  Returns the solid-angle pdf the reference would return for an
// unoccluded sample (INVALID_PDF on the single-sided rejection) and the point to connect to.
RD_DEV float lightPdfUnoccluded(const DScene &s, v3 pos, const LightPick &p, v3 &radiance, v3 &wi) {
    if (p.isEnv) {  // sampleEnvironmentMap (scene.h:380-399) minus its occlusion test
        radiance = p.radiance;
        wi = p.normal;
        return p.area;
    }
    v3 posToSampled = p.sampled - pos;
    if (dot(p.normal, posToSampled) > -1e-6f) return INVALID_PDF;  // SCENE_LIGHT_SINGLE_SIDED
    radiance = p.radiance;
    const float d2 = dot(posToSampled, posToSampled);
    wi = posToSampled * rsqrt_exact(d2);  // normalize(posToSampled)
    float power = luminance(radiance) / (p.area * 2.f * PI_F);
    return pdfAreaToSolidAngleFrom(power * s.sumLightPowerInv, d2, p.normal, wi);  // pdfAreaToSolidAngle(.., pos, p.sampled, p.normal)
}

// sampleDirectLightNoVisibility (scene.h:458-492)
RD_DEV float sampleDirectLightNoVisibility(const DScene &s, v3 pos, v4 r, v3 &radiance, v3 &wi, float &dist) {
    if (s.lightSamplerLength == 0) return INVALID_PDF;
    LightPick p = pickLightPoint(s, pos, r);
    if (p.isEnv) {  // scene.h:466-469
        dist = 1e10f;
        radiance = p.radiance;
        wi = p.normal;
        return p.area;
    }
    v3 posToSampled = p.sampled - pos;
    if (dot(p.normal, posToSampled) > -1e-6f) return INVALID_PDF;
    radiance = p.radiance;
    const float d2 = dot(posToSampled, posToSampled);
    dist = __builtin_sqrtf(d2);            // length(posToSampled)
    wi = posToSampled * (1.f / dist);      // normalize(posToSampled) = v * (1 / sqrt(dot(v, v)))
    float power = luminance(radiance) / (p.area * 2.f * PI_F);
    return pdfAreaToSolidAngleFrom(power * s.sumLightPowerInv, d2, p.normal, wi);  // pdfAreaToSolidAngle(.., pos, p.sampled, p.normal)
}

}  // namespace rd
