/* include/radish_host.h — host-side scene preparation (CPU only, no HIP).
 *
 * These entry points produce the arrays of the reference's DevScene layout from a world-space triangle
 * soup.  They are the "step before the path" (SURVEY.md §8f N2): in a Radish build they replace the bodies
 * of BVHBuilder::build, DiscreteSampler1D's constructor and the light-list loop of Scene::buildDevData,
 * whose outputs DevScene::create uploads (src/scene.cpp:461-551).  libradish_host.so, built with g++.
 */
#ifndef RADISH_HOST_H
#define RADISH_HOST_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces BVHBuilder::build + buildMTBVH (src/bvh.cpp:12-134,136-183).
 * vertices: float[numPrims*9] triangle soup.  boxesOut: float[(2*numPrims-1)*6] AABBs in depth-first order.
 * nodesOut[i]: int32[(2*numPrims-1)*3] = {primitiveId, boundingBoxId, nextNodeIfMiss}, i = 0..5 (the six
 * direction-ordered threaded arrays).  Returns BVHSize = 2*numPrims-1, or a negative error code. */
int32_t rdh_build_bvh(const float *vertices, int32_t numPrims, float *boxesOut, int32_t *const nodesOut[6]);

/* Replaces DiscreteSampler1D<float>::DiscreteSampler1D (src/sampler.h:81-125).
 * tableOut: {float prob; int32 failId}[n].  *sumOut receives the float sum of `values`. Returns 0 / <0. */
int32_t rdh_build_alias_table(const float *values, int32_t n, void *tableOut, float *sumOut);

/* Replaces the emissive-triangle loop of Scene::buildDevData (src/scene.cpp:192-223).
 * materials: 44-byte Material[numMaterials].  Outputs sized for numPrims entries; returns the light count.
 * lightPowerOut[i] = luminance(baseColor) * 2 * pi * area. */
int32_t rdh_build_light_list(const float *vertices, const int32_t *materialIds, int32_t numPrims,
                             const void *materials, int32_t numMaterials, int32_t *lightPrimIdsOut,
                             float *lightUnitRadianceOut, float *lightPowerOut);

/* Replaces the environment-map half of Scene::createLightSampler (src/scene.cpp:146-157):
 * pdfOut[i*width+j] = luminance(texel) * sin((0.5 + i) / height * PI).  Feed pdfOut to rdh_build_alias_table to get
 * the env-map sampler; its sum is the power entry appended LAST to the light-power list (src/scene.cpp:163). */
int32_t rdh_build_envmap_pdf(const float *texels, int32_t width, int32_t height, float *pdfOut);

/* Replaces Camera::update (src/sceneStructs.h:93-107) plus the fov bookkeeping of Scene::loadCamera
 * (src/scene.cpp:378-383).  camera196 in/out: resolution, position, rotation, fov.y (degrees, "FovY"),
 * lensRadius, focalDist must be set; view/up/right/rotationMatInv/fov.x/tanFovY/viewProjection are written. */
void rdh_camera_update(void *camera196);

/* ---- Scene files -----------------------------------------------------------------------------------
 * Replaces Scene::Scene / loadMaterial / loadModel / loadCamera / addTexture (src/scene.cpp:108-141,256-459),
 * Resource::loadOBJMesh (src/scene.cpp:29-63) and the flattening loop of Scene::buildDevData (src/scene.cpp:190-223):
 * parses a Radish scene text file (grammar: SURVEY.md App. B), loads the OBJ meshes and textures it names and returns
 * the world-space triangle soup in the reference's layouts.  Feed the result to rdh_build_bvh / rdh_build_light_list /
 * rdh_build_alias_table / rdh_build_envmap_pdf and then to rdh_scene_upload (include/radish_hip.h). */
typedef struct rdh_host_texture {
    int32_t width, height;
    const float *data; /* vec3[width*height], row 0 first */
} rdh_host_texture;

/* Decoder for image formats the library does not read itself (.png .hdr .ppm .pgm .pfm are built in).  Must return 0 and a
 * malloc()ed vec3[w*h] (linear floats; LDR samples as v/255) — flipped vertically when flipY != 0 (stb's
 * stbi_set_flip_vertically_on_load, src/scene.cpp:110,133-135).  The library free()s the buffer. */
typedef int32_t (*rdh_texture_decode_fn)(const char *path, int32_t flipY, float **rgbOut, int32_t *width, int32_t *height,
                                         void *user);

typedef struct rdh_parsed_scene {
    int32_t numPrims;
    const float *vertices;      /* vec3[3*numPrims], world space            */
    const float *normals;       /* vec3[3*numPrims], normalised             */
    const float *texcoords;     /* vec2[3*numPrims]                         */
    const int32_t *materialIds; /* int[numPrims]                            */
    int32_t numMaterials;
    const void *materials;      /* Material[numMaterials], 44 B (src/material.h:276-286); map ids index `textures` */
    int32_t numTextures;
    const rdh_host_texture *textures;
    int32_t envMapTexId;        /* -1: none                                 */
    int32_t apertureMaskTexId;  /* -1: none (loaded, unused: the reference's lens sampling is disabled) */
    int32_t hasCamera;
    uint8_t camera[196];        /* Camera (src/sceneStructs.h:118-130), updated */
    int32_t traceDepth;         /* "Depth"  → Settings::traceDepth          */
    int32_t iterations;         /* "Sample" → state.iterations              */
    char imageName[256];        /* "File"                                   */
    void *opaque;
} rdh_parsed_scene;

/* Returns 0 and *out (release with rdh_scene_parse_free), or RDH_HOST_ERR_* with a message in err.  Where the reference
 * prints and throws / exits, this returns RDH_HOST_ERR_SCENE. */
int32_t rdh_scene_parse(const char *sceneFile, rdh_texture_decode_fn decodeOrNull, void *user, rdh_parsed_scene **out,
                        char *err, int32_t errLen);
void rdh_scene_parse_free(rdh_parsed_scene *scene);

#define RDH_HOST_ERR_ARGS (-1)
#define RDH_HOST_ERR_SCENE (-2)

#ifdef __cplusplus
}
#endif
#endif
