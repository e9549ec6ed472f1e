/* oracle/oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * C ABI of the single-threaded CPU restatement of Radish's per-pixel path-tracing inner loop
 * (/root/reference/src/pathtrace.cu, restir.cu and the device code they inline).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so; the product
 * (radish_pt_amd/, libradish_hip.so) never includes, links or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors, scenes or Sobol table, cannot be
 * compiled in this image (no nvcc/glm/stb/GLFW/xmake; MSVC-only headers) and depends on glm + CUDA
 * libdevice arithmetic that is not under /root/reference.  This oracle is therefore
 * pinned only by hand-computed known-answer tests (tests/test_oracle_kat.py) and by internal
 * cross-checks (BVH walk vs brute force, white-furnace, alias-table sums).
 */
#ifndef RADISH_ORACLE_H
#define RADISH_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_texture { int32_t width, height; const float *data; /* vec3[width*height] */ } orc_texture;

/* Host arrays in the reference's DevScene layout (src/scene.h:494-517). */
typedef struct orc_scene_desc {
    const float *vertices;      /* vec3[3*numPrims]  (triangle soup)            */
    const float *normals;       /* vec3[3*numPrims]                             */
    const float *texcoords;     /* vec2[3*numPrims]                             */
    const float *boundingBoxes; /* AABB[bvhSize] = {pMin.xyz, pMax.xyz}         */
    const int32_t *bvhNodes[6]; /* MTBVHNode[bvhSize] = {primId, boxId, next}   */
    int32_t bvhSize;            /* 2*numPrims-1                                 */
    int32_t numPrims;
    const int32_t *materialIds; /* int[numPrims]                                */
    const void *materials;      /* Material[numMaterials], 44 B each            */
    int32_t numMaterials;
    int32_t numLights;          /* entries in lightPrimIds / lightUnitRadiance  */
    const int32_t *lightPrimIds;
    const float *lightUnitRadiance; /* vec3[numLights]                          */
    float sumLightPowerInv;
    int32_t lightSamplerLength;     /* alias table length (== numLights, no env map) */
    const void *lightSampler;       /* BinomialDistrib<float>[len] = {float prob; int failId} */
    const uint32_t *sobol;          /* uint32[10000][200]                        */
    /* textures (src/image.h:89-91 DevTextureObj, host pointers) and the environment map (src/scene.h:511-512) */
    int32_t numTextures;
    const struct orc_texture *textures;
    int32_t envMapTexId;            /* -1 = none; else index into textures               */
    int32_t envMapSamplerLength;    /* width*height of the env map, 0 = none             */
    const void *envMapSampler;      /* BinomialDistrib<float>[envMapSamplerLength]       */
} orc_scene_desc;

/* G-buffer in the reference's field order (src/gBuffer.h:42-57), host pointers. */
typedef struct orc_gbuffer {
    float *albedo;     /* vec3[w*h]  */
    float *normal[2];  /* vec3[w*h]  */
    int32_t *motion;   /* int[w*h]   */
    float *depth[2];   /* float[w*h] */
    int32_t *primId[2];/* int[w*h] — holds the MATERIAL id, -2 light, -1 miss (gBuffer.cu:32-47) */
    int32_t frameIdx;
    int32_t width, height;
} orc_gbuffer;

typedef struct orc_stats {
    uint64_t closestRays;  /* calls to DevScene::intersect      */
    uint64_t anyRays;      /* calls to DevScene::testOcclusion  */
    uint64_t nodeVisits;   /* AABB tests, both walk kinds        */
    uint64_t triTests;     /* Möller–Trumbore tests              */
    uint64_t closestHits;  /* intersect calls that found a hit   */
} orc_stats;

typedef struct orc_hit { int32_t primId; float u, v, t; } orc_hit;

typedef struct orc_scene orc_scene;

/* The descriptor's arrays are borrowed, not copied: keep them alive while the handle is used. */
orc_scene *orc_scene_create(const orc_scene_desc *desc);
void orc_scene_destroy(orc_scene *s);
void orc_stats_reset(orc_scene *s);
void orc_stats_get(const orc_scene *s, orc_stats *out);
/* Analysis hook (tests/tools/visit_stats.py): while set, every box test increments visitHist[ordering*bvhSize + node]
 * (uint32[6*bvhSize]) and every ray adds to rayLenHist (uint64[128]: [kind][log2 bucket] counts, then visit sums;
 * kind 0 closest, 1 any).  Pass NULLs to turn it off. */
void orc_debug_visit_hist(orc_scene *s, uint32_t *visitHist, uint64_t *rayLenHist);

/* rays: {origin.xyz, direction.xyz} x n.  Restates DevScene::intersect (scene.h:262-301). */
void orc_trace_closest(orc_scene *s, const float *rays, int64_t n, orc_hit *hits);
/* Same result contract, brute force over all triangles (scene.h:209-232 naiveIntersect). */
void orc_trace_closest_naive(orc_scene *s, const float *rays, int64_t n, orc_hit *hits);
/* segments: {x.xyz, y.xyz} x n -> 1 if occluded.  Restates DevScene::testOcclusion (scene.h:303-334). */
void orc_trace_occluded(orc_scene *s, const float *segments, int64_t n, int32_t *occluded);

/* camera196: the reference's 196-byte Camera (sceneStructs.h:118-130).
 * Pixels processed: index = pixBegin, pixBegin+pixStride, ... < pixEnd (row-major y*W+x). */
void orc_path_trace(orc_scene *s, const void *camera196, float *directIllum, float *indirectIllum,
                    int iter, int looper, int maxDepth, int64_t pixBegin, int64_t pixEnd, int64_t pixStride);
void orc_path_trace_direct(orc_scene *s, const void *camera196, float *directIllum, int iter, int looper,
                           int64_t pixBegin, int64_t pixEnd, int64_t pixStride);
void orc_gbuffer_render(orc_scene *s, const void *camera196, const void *lastCamera196, orc_gbuffer *gb);

/* Two-pass ReSTIR DI (restir.cu:97-203 with the race of SURVEY F6 removed by a pass boundary).
 * reservoirs: DirectReservoir[w*h] of 36 B each.  reuseMask: bit0 temporal, bit1 spatial.
 * faithfulRIS != 0 reproduces Reservoir::update's truthiness test (restir.h:21). */
void orc_restir_direct(orc_scene *s, const void *camera196, float *directIllum, int iter, int looper,
                       void *reservoirOut, const void *reservoirIn, void *reservoirTemp,
                       const orc_gbuffer *gb, int firstFrame, int reuseMask, int faithfulRIS,
                       int numSpatial, int risCount);

/* Display path: sendImageToPBO's four overloads (pathtrace.cu:32-118) on host memory; pbo = uchar4[width*height].
 * kind 0 vec3 image (tone mapping 0 None / 1 Filmic / 2 ACES, scale), 1 vec2, 2 float, 3 int pixel index. */
void orc_copy_image_to_pbo(uint8_t *pbo, const void *image, int width, int height, int kind, int toneMapping, float scale);
float orc_pow_gamma(float x); /* the fixed x^(1/2.2f) recipe both sides use for Math::gammaCorrection */

/* Denoisers (denoiser.cu) on host memory: EAW filter (:17-84), SVGF filter with variance (:92-173), modulate (:175-185),
 * add (:187-206), temporalAccumulate (:208-262), estimateVariance (:264-299), filterVariance (:301-328).  gb supplies the
 * current (frameIdx) and last planes; width/height are gb's. */
void orc_denoise_eaw(float *colorOut, const float *colorIn, const orc_gbuffer *gb, const void *camera196, float sigLumin,
                     float sigNormal, float sigDepth, int level);
void orc_denoise_svgf(float *colorOut, const float *colorIn, float *varianceOut, const float *varianceIn, const float *varFiltered,
                      const orc_gbuffer *gb, const void *camera196, float sigLumin, float sigNormal, float sigDepth, int level);
void orc_denoise_modulate(float *image, const orc_gbuffer *gb);
void orc_denoise_add(float *out, const float *in1, const float *in2, int width, int height);
void orc_denoise_temporal_accumulate(float *colorAccumOut, const float *colorAccumIn, float *momentAccumOut,
                                     const float *momentAccumIn, const float *colorIn, const orc_gbuffer *gb, int first);
void orc_denoise_estimate_variance(float *variance, const float *moment, int width, int height);
void orc_denoise_filter_variance(float *varianceOut, const float *varianceIn, int width, int height);
float orc_exp(float x);          /* the fixed e^x recipe of the denoisers' weights */
float orc_pow(float x, float y); /* the fixed x^y recipe (x >= 0) */

/* Known-answer-test hooks for single device functions. */
uint32_t orc_utilhash(uint32_t a);
int orc_aabb_intersect(const float *box6, const float *ray6, float *tMin);
int orc_intersect_triangle(const float *ray6, const float *v9, float *bary2, float *dist);
void orc_sincos(float x, float *s, float *c);
float orc_atan2(float y, float x);
/* linearSample (src/image.h:42-87) of texture `texId` at uv; texId = -2 evaluates proceduralTexture (scene.h:77-86). */
void orc_texture_sample(orc_scene *s, int texId, const float *uv2, float *rgb3);
/* BSDF hooks: material44 = 44-byte Material; which: 0 BSDF (out[0..2]), 1 pdf (out[0]),
 * 2 sample (r3 -> out = dir.xyz, bsdf.xyz, pdf, type as float bits). */
void orc_material_eval(const void *material44, int which, const float *n3, const float *wo3, const float *wi_or_r3,
                       float *out8);
void orc_camera_sample(const void *camera196, int x, int y, const float *r4, float *ray6);
/* Light sampler: sampleDirectLight (scene.h:419-456, visibility != 0: traces the shadow ray) / sampleDirectLightNoVisibility
 * (:458-492); returns the pdf (INVALID_PDF = -1 when rejected), radiance, wi, dist (NoVisibility only). */
void orc_sample_direct_light(orc_scene *s, const float *pos3, const float *r4, int visibility, float *radiance3, float *wi3,
                             float *dist, float *pdf);
/* Reservoir arithmetic (restir.h:10-92) on 36-byte DirectReservoir records: op 0 merge(rhs, rnd), 1 preClampedMerge<M>(rhs, rnd),
 * 2 update(rhs.sample, rhs.weight, rnd) as written (truthiness test, :21), 3 update with the corrected `<` test, 4 checkValidity. */
void orc_reservoir_op(int op, void *r36, const void *rhs36, float rnd, int M);
/* Reservoir::W (restir.h:37-40) at a surface with normal n, outgoing direction wo and the 44-byte Material. */
float orc_reservoir_W(const void *r36, const void *material44, const float *n3, const float *wo3);
float orc_power_heuristic(float f, float g); /* Math::powerHeuristic (mathUtil.h:81-84) */

#ifdef __cplusplus
}
#endif
#endif
