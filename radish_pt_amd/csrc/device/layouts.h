// radish_pt_amd/csrc/device/layouts.h — HBM layout of the scene as the gfx950 kernels read it.
//
// The reference keeps the scene as AoS glm::vec3 arrays plus a 12-byte node that points into a separate AABB
// array (src/scene.h:494-517, src/bvh.h:161-170): every traversal step is two dependent, unaligned gathers.
// rdh_scene_upload re-lays it out so that each step is ONE aligned 32-byte record (two dwordx4 loads) and each
// triangle test is ONE aligned 48-byte record; nothing here changes a value — records are byte copies of the
// reference's floats/ints, only their placement differs.
#pragma once
#include "rmath.h"

namespace rd {

// One threaded-BVH step: box + leaf primitive + miss link.  32 B, 32-B aligned.  Six arrays (one per ordering,
// src/bvh.cpp:136-183), each bvhSize records in traversal order.
struct __attribute__((aligned(16))) NodeRec {
    float4 lo_prim;  // pMin.xyz, primitiveId (int bits; -1 = inner node)
    float4 hi_next;  // pMax.xyz, nextNodeIfMiss (int bits)
};
static_assert(sizeof(NodeRec) == 32, "NodeRec");

// One triangle: the three vertices exactly as in `vertices[3*prim+{0,1,2}]` + the material id.  48 B.
struct __attribute__((aligned(16))) TriRec {
    float4 a;  // va.xyz, vb.x
    float4 b;  // vb.yz, vc.xy
    float4 c;  // vc.z, materialId (int bits), 0, 0
};
static_assert(sizeof(TriRec) == 48, "TriRec");

// Shading attributes of one triangle (read once per closest hit): normals + texcoords.  64 B.
struct __attribute__((aligned(16))) AttrRec {
    float4 a;  // na.xyz, nb.x
    float4 b;  // nb.yz, nc.xy
    float4 c;  // nc.z, ta.xy, tb.x
    float4 d;  // tb.y, tc.xy, 0
};
static_assert(sizeof(AttrRec) == 64, "AttrRec");

// One emissive triangle, indexed by light id (folds lightPrimIds → vertices and lightUnitRadiance).  48 B.
struct __attribute__((aligned(16))) LightRec {
    float4 a;  // v0.xyz, v1.x
    float4 b;  // v1.yz, v2.xy
    float4 c;  // v2.z, radiance.xyz
};
static_assert(sizeof(LightRec) == 48, "LightRec");

// The same light with what every RIS candidate would recompute from it — the plain normal (Math::triangleNormal) and the
// area pdf `luminance(radiance) / (area * 2 pi) * sumLightPowerInv` of scene.h:489-491 — evaluated ONCE per scene by
// k_light_precompute with the same device functions, hence the same bits.  64 B: what the ReSTIR RIS loop stages in LDS.
struct __attribute__((aligned(16))) LightPre {
    float4 a, b, c;  // as LightRec
    float4 d;        // triangleNormal.xyz, power * sumLightPowerInv
};
static_assert(sizeof(LightPre) == 64, "LightPre");

// Material (src/material.h:276-286) in three aligned quads.  48 B.
struct __attribute__((aligned(16))) MatRec {
    float4 a;  // type (int bits), baseColor.xyz
    float4 b;  // metallic, roughness, ior, 0
    int4 maps; // baseColorMapId, normalMapId, metallicMapId, roughnessMapId (-1 none, -2 procedural base colour)
};

struct AliasRec {  // BinomialDistrib<float> (src/sampler.h:66-69), unchanged
    float prob;
    int failId;
};

struct Counters {
    unsigned long long closestRays, anyRays, nodeVisits, triTests, closestHits;
};

// The tree ONCE, as SIBLING PAIRS, for the per-lane walks (round 3): record q = the two children of one inner node, 64 B on one
// 64-byte half line, shared by the six orderings (the threaded arrays hold every box six times) — both children are visited
// by every walk that enters the parent (the near one now, the far one when the near subtree is done), so one round trip serves
// two visits.  w = primitiveId of a leaf child, or ~q' when the child is an inner node whose children are pair q'.  `bits`: bit k set when ordering k
// (src/bvh.cpp:171-180) visits child 1 first.  Pairs are numbered in ordering 0's pre-order of their parents
// (the root's pair is 0).  The root itself — a box every walk tests first — travels in DScene (kernel arguments, no load).
struct __attribute__((aligned(64))) PairRec {
    float4 lo0_w0;    // child 0: pMin.xyz, w
    float4 hi0_bits;  // child 0: pMax.xyz, the parent's ordering bits
    float4 lo1_w1;    // child 1: pMin.xyz, w
    float4 hi1_pad;   // child 1: pMax.xyz, 0
};
static_assert(sizeof(PairRec) == 64, "PairRec");

struct DScene {
    const NodeRec *nodes[6];  // one allocation: nodes[k] = nodes[0] + k * (bvhSize + 1)
    const PairRec *pairs;     // (bvhSize - 1) / 2 records; null when the six arrays handed to rdh_scene_upload are not six pre-orders
                              // of ONE binary tree (then every kernel walks nodes[k])
    int treeDepth;            // most far children any walk can have pending at once (over the six orderings)
    float4 rootLo, rootHi;    // the root's box; rootLo.w = its w (a one-triangle scene has a leaf root)
    const TriRec *tris;
    const AttrRec *attrs;
    const MatRec *mats;
    const unsigned char *primClass;  // per triangle: shading class of its material (0 terminal / other, 1 Lambertian, 2 metallic
                                     // workflow, 3 dielectric) — what the wavefront pipeline's material sort bins by
    const LightRec *lights;
    const LightPre *lightPre;  // numLights records (no entry for the environment map)
    const AliasRec *lightAlias;
    const uint32_t *sobol;
    const float *texData;   // every texture's texels (vec3), concatenated (src/scene.cpp:464-480)
    const int4 *texInfo;    // per texture: width, height, first texel, 0   (DevTextureObj, src/image.h:89-91)
    const AliasRec *envAlias;  // envMapSampler (src/scene.h:512)
    int envTex;             // envMap's texture id or -1
    int envSamplerLength;
    Counters *counters;
    int bvhSize;
    int numPrims;
    int lightSamplerLength;
    float sumLightPowerInv;
};

// Camera fields the kernels use (src/sceneStructs.h:118-130), plus tan(radians(fov.y)) hoisted to the host
// (the reference re-evaluates it per thread, :75).
struct DCamera {
    int resx, resy;
    v3 position, view, up, right;
    m3 rotationMatInv;
    float lensRadius, focalDist;
    float tanFovY;  // tanf(radians(fov.y)) — NOT the struct's tanFovY field, which the kernels never read
};

// Which pixels this launch renders and where they are stored (single GPU: the whole frame in frame layout;
// multi-GPU: this rank's interleaved tiles in a packed tile-major buffer).
struct PixelMap {
    int W, H;
    int tile;          // tile edge in pixels (multiple of 8)
    int tilesX;        // tiles per frame row
    int numTiles;      // tiles in the frame
    int rank, world;
    int tilesPerRank;  // ceil(numTiles / world)
    int packed;        // 0: out index = y*W+x ; 1: out index = (localTile*tile*tile + ly*tile + lx)
    int numBlocks;     // 8x8-pixel blocks (= waves) of work in this launch
};

}  // namespace rd
