// tests/shim/shim_syntax_check.cpp — compile check of radish_pt_amd/csrc/radish_shim.hpp (tests/test_shim_compiles.py).
//
// The shim is the reference-signature side of the drop-in boundary; glm and the reference's headers are not in this image,
// so it cannot be compiled against them here.  This file declares — in its own minimal form, members named as the shim reads
// them — the few types of the reference the shim touches (src/sceneStructs.h:118-130 Camera, src/gBuffer.h:15-58 GBuffer,
// src/scene.h:33-70,520-577 MeshData / Scene, src/sampler.h:76-139 DiscreteSampler1D, src/image.h:9-40 Image,
// src/common.h:50-72 Settings / State, src/denoiser.h:16-82 the filter classes), includes the shim with every section
// switched on, and instantiates its templates.  It is compiled with -fsyntax-only -Wall -Werror: a typo, a wrong argument
// count against include/radish_hip.h or a layout static_assert that no longer holds fails the CPU test suite.  Nothing here is
// linked or run.
#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>  // hipMalloc / hipFree used by the denoiser section; also brings uchar4

namespace glm {
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct ivec2 { int x, y; };
struct mat3 { vec3 c[3]; };
struct mat4 { float m[16]; };
}  // namespace glm

struct Camera {  // 196 bytes, field order of src/sceneStructs.h:118-130
    glm::ivec2 resolution;
    glm::vec3 position, rotation, view, up, right;
    glm::vec2 fov, pixelLength;
    glm::mat3 rotationMatInv;
    glm::mat4 viewProjection;
    float lensRadius, focalDist, tanFovY;
};
static_assert(sizeof(Camera) == 196, "Camera layout");

struct DevScene;
struct GBuffer {  // src/gBuffer.h:15-58 with DENOISER_ENCODE_NORMAL false, DENOISER_ENCODE_POSITION true
    void create(int width, int height);
    void destroy();
    void render(DevScene *scene, const Camera &cam);
    void update(const Camera &cam);
    glm::vec3 *albedo = nullptr;
    glm::vec3 *normal[2] = {nullptr};
    int *motion = nullptr;
    float *depth[2] = {nullptr};
    int *primId[2] = {nullptr};
    int frameIdx = 0;
    Camera lastCam;
    int width;
    int height;
};

struct Material { int type; glm::vec3 baseColor; float metallic, roughness, ior; int maps[4]; };
static_assert(sizeof(Material) == 44, "Material layout");
struct AABB { glm::vec3 pMin, pMax; };
struct MTBVHNode { int primitiveId, boundingBoxId, nextNodeIfMiss; };
template <typename T> struct BinomialDistrib { T prob; int failId; };
template <typename T> struct DiscreteSampler1D {
    std::vector<BinomialDistrib<T>> binomDistribs;
    T sum = static_cast<T>(0);
};
struct Image {
    int width() const { return w; }
    int height() const { return h; }
    glm::vec3 *data() const { return pixels; }
    int w = 0, h = 0;
    glm::vec3 *pixels = nullptr;
};
struct MeshData { std::vector<glm::vec3> vertices, normals; std::vector<glm::vec2> texcoords; };
struct Scene {
    std::vector<Image *> textures;
    std::vector<Material> materials;
    std::vector<int> materialIds;
    int BVHSize = 0;
    std::vector<AABB> boundingBoxes;
    std::vector<std::vector<MTBVHNode>> BVHNodes;
    MeshData meshData;
    std::vector<int> lightPrimIds;
    std::vector<glm::vec3> lightUnitRadiance;
    DiscreteSampler1D<float> lightSampler, envMapSampler;
    int envMapTexId = -1;
    DevScene *devScene = nullptr;
    Camera camera;
};
struct Settings { static int traceDepth; static int reservoirReuse; };
struct State { static bool camChanged; static int looper; static Scene *scene; };

struct EAWaveletFilter {
    EAWaveletFilter() = default;
    EAWaveletFilter(int width, int height, float sigLumin, float sigNormal, float sigDepth)
        : sigLumin(sigLumin), sigNormal(sigNormal), sigDepth(sigDepth), width(width), height(height) {}
    void filter(glm::vec3 *colorOut, glm::vec3 *colorIn, const GBuffer &gBuffer, const Camera &cam, int level);
    void filter(glm::vec3 *colorOut, glm::vec3 *colorIn, float *varianceOut, float *varianceIn, float *filteredVar,
                const GBuffer &gBuffer, const Camera &cam, int level);
    float sigLumin, sigNormal, sigDepth;
    int width = 0, height = 0;
};
struct LeveledEAWFilter {
    void create(int width, int height, int level);
    void destroy();
    void filter(glm::vec3 *&colorOut, glm::vec3 *colorIn, const GBuffer &gBuffer, const Camera &cam);
    EAWaveletFilter waveletFilter;
    int level = 0;
    glm::vec3 *tmpImg = nullptr;
};
struct SpatioTemporalFilter {
    void create(int width, int height, int level);
    void destroy();
    void temporalAccumulate(glm::vec3 *colorIn, const GBuffer &gBuffer);
    void estimateVariance();
    void filterVariance();
    void filter(glm::vec3 *&colorOut, glm::vec3 *colorIn, const GBuffer &gBuffer, const Camera &cam);
    void nextFrame();
    EAWaveletFilter waveletFilter;
    int level = 0;
    glm::vec3 *accumColor[2] = {nullptr};
    glm::vec3 *accumMoment[2] = {nullptr};
    float *variance = nullptr;
    bool firstTime = true;
    glm::vec3 *tmpColor = nullptr;
    float *tmpVar = nullptr;
    float *filteredVar = nullptr;
    int frameIdx = 0;
};

#define RADISH_SHIM_NO_REFERENCE_HEADERS
#define RADISH_SHIM_WITH_DENOISER
#ifdef SHIM_CHECK_MULTI_GPU
#define RADISH_SHIM_MULTI_GPU
#endif
#ifdef SHIM_CHECK_ONE_PROCESS
#define RADISH_SHIM_ONE_PROCESS_GPUS
#endif
#include "radish_shim.hpp"

// use everything once, so that templates are instantiated and overloads resolved
void shim_syntax_check_uses(Scene &scene, GBuffer &gb, glm::vec3 *img, glm::vec3 *img2, uchar4 *pbo, const uint32_t *sobol) {
    radish_shim::devSceneCreate(scene, sobol);
#ifdef SHIM_CHECK_ONE_PROCESS
    radish_shim::commInitAll(scene, sobol, 2);
#else
    unsigned char id[128];
    radish_shim::commUniqueId(id);
    radish_shim::commInit(id, 0, 1);
#endif
    pathTraceInit();
    pathTrace(img, img2, 0);
    pathTraceDirect(img, 0);
    ReSTIRInit();
    gb.render(scene.devScene, scene.camera);
    ReSTIRDirect(img, 0, gb);
    copyImageToPBO(pbo, img, 4, 4, 2);
    copyImageToPBO(pbo, reinterpret_cast<glm::vec2 *>(img), 4, 4);
    copyImageToPBO(pbo, reinterpret_cast<float *>(img), 4, 4);
    copyImageToPBO(pbo, reinterpret_cast<int *>(img), 4, 4);
    LeveledEAWFilter eaw;
    eaw.create(4, 4, 5);
    glm::vec3 *out = img2;
    eaw.filter(out, img, gb, scene.camera);
    eaw.destroy();
    SpatioTemporalFilter svgf;
    svgf.create(4, 4, 5);
    svgf.filter(out, img, gb, scene.camera);
    svgf.nextFrame();
    svgf.destroy();
    modulateAlbedo(img, gb);
    addImage(img, img2, 4, 4);
    addImage(img, img2, img, 4, 4);
    ReSTIRFree();
    pathTraceFree();
    radish_shim::devSceneDestroy();
}
