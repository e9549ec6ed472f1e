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
};
RD_DEV LightPick pickLightPoint(const DScene &s, v4 r) {
    int lightId = lightAliasSample(s, r.x, r.y);
    const LightRec *l = s.lights + lightId;
    float4 A = l->a, B = l->b, C = l->c;
    v3 v0 = mk3(A.x, A.y, A.z), v1 = mk3(A.w, B.x, B.y), v2_ = mk3(B.z, B.w, C.x);
    LightPick p;
    p.sampled = sampleTriangleUniform(v0, v1, v2_, r.z, r.w);
    p.normal = triangleNormal(v0, v1, v2_);
    p.area = triangleArea(v0, v1, v2_);
    p.radiance = mk3(C.y, C.z, C.w);
    return p;
}

// Everything of sampleDirectLight (scene.h:419-456) except the shadow ray, which the caller traces — in the
// megakernel immediately (reference order: occlusion first, then the single-sided test), in the
// wavefront pipeline through the shadow queue.  Returns the solid-angle pdf the reference would return for an
// unoccluded sample (INVALID_PDF on the single-sided rejection) and the point to connect to.
RD_DEV float lightPdfUnoccluded(const DScene &s, v3 pos, const LightPick &p, v3 &radiance, v3 &wi) {
    v3 posToSampled = p.sampled - pos;
    if (dot(p.normal, posToSampled) > -1e-6f) return INVALID_PDF;  // SCENE_LIGHT_SINGLE_SIDED
    radiance = p.radiance;
    wi = normalize(posToSampled);
    float power = luminance(radiance) / (p.area * 2.f * PI_F);
    return pdfAreaToSolidAngle(power * s.sumLightPowerInv, pos, p.sampled, p.normal);
}

// sampleDirectLightNoVisibility (scene.h:458-492)
RD_DEV float sampleDirectLightNoVisibility(const DScene &s, v3 pos, v4 r, v3 &radiance, v3 &wi, float &dist) {
    if (s.lightSamplerLength == 0) return INVALID_PDF;
    LightPick p = pickLightPoint(s, r);
    v3 posToSampled = p.sampled - pos;
    if (dot(p.normal, posToSampled) > -1e-6f) return INVALID_PDF;
    radiance = p.radiance;
    wi = normalize(posToSampled);
    dist = length(posToSampled);
    float power = luminance(radiance) / (p.area * 2.f * PI_F);
    return pdfAreaToSolidAngle(power * s.sumLightPowerInv, pos, p.sampled, p.normal);
}

}  // namespace rd
