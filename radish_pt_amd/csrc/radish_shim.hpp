// radish_pt_amd/csrc/radish_shim.hpp — source-compatible C++ shim: the reference's render API on top of the C ABI.
//
// Drop this header (and libradish_hip.so) into a Radish checkout in place of the BODIES of
//   pathtrace.cu  (pathTraceInit/Free, pathTrace, pathTraceDirect)        /root/reference/src/pathtrace.cu:28-30,351-407
//   restir.cu     (ReSTIRInit/Free, ReSTIRDirect)                         /root/reference/src/restir.cu:205-251
//   gBuffer.cu    (GBuffer::render)                                       /root/reference/src/gBuffer.cu:83-103
//   scene.cpp     (DevScene::create/destroy)                              /root/reference/src/scene.cpp:461-574
// keeping the declarations in pathtrace.h / restir.h / gBuffer.h / scene.h untouched, so main.cpp and scene.cpp call
// exactly what they call today.  It needs only what those translation units already include (glm, the reference's
// own Scene / Camera / GBuffer / Settings / State definitions).  glm and the reference's headers are not available in the
// build image, so it cannot be LINKED into Radish here; tests/test_shim_compiles.py compiles it
// (-fsyntax-only, -Wall -Werror) against a minimal local set of the declarations it touches (tests/shim/), so that a typo
// cannot ship, and the Python mirror radish_pt_amd/api.py is the host side the GPU tests exercise.
//
// Macros:  RADISH_SHIM_NO_REFERENCE_HEADERS  the including file has already declared glm / Camera / GBuffer / Scene /
//                                            Settings / State (skip the reference's #includes)
//          RADISH_SHIM_HELPERS_ONLY          only namespace radish_shim (no free functions)
//          RADISH_SHIM_WITH_DENOISER         also the bodies of denoiser.h's classes
//          RADISH_SHIM_MULTI_GPU             pathTrace / ReSTIRDirect / GBuffer::render run on a tile partition over RCCL:
//                                            call radish_shim::commInit(id, rank, world) once per process (one process per
//                                            GPU); every rank then gets the whole frame, exactly as the single-GPU calls do
//          RADISH_SHIM_ONE_PROCESS_GPUS      the same from ONE process driving n GPUs — the reference's own host model (one
//                                            process, one frame loop, main.cpp:163-202): after DevScene::create call
//                                            radish_shim::commInitAll(scene, sobol, n); the caller's images and G-buffer live on
//                                            device 0 as today, devices 1..n-1 get mirrors the shim owns (needs hip_runtime_api.h)
//
// Error behaviour reproduces checkCUDAError (src/cudaUtil.h:16-34): print `HIP error (file:line): msg: text` and exit.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "radish_hip.h"

#ifndef RADISH_SHIM_NO_REFERENCE_HEADERS
#include "common.h"   // Settings, State
#include "gBuffer.h"  // GBuffer
#include "scene.h"    // Scene, DevScene, Camera
#endif

namespace radish_shim {

inline rdh_ctx *&ctx() {
    static rdh_ctx *c = nullptr;
    return c;
}

inline void check(int rc, const char *msg, const char *file, int line) {
    if (rc == RDH_OK) return;
    std::fprintf(stderr, "HIP error (%s:%d): %s: %s\n", file, line, msg, rdh_last_error(ctx()));
    std::exit(EXIT_FAILURE);
}
#define RADISH_CHECK(expr, msg) ::radish_shim::check((expr), (msg), __FILE__, __LINE__)

// What DevScene::create does with a fully built `Scene` (src/scene.cpp:461-551): hand the host arrays to the device.
template <typename SceneT>
inline void devSceneCreate(const SceneT &scene, const uint32_t *sobol10kx200, rdh_ctx *target = nullptr) {
    if (!target && !ctx()) RADISH_CHECK(rdh_create(&ctx(), 0), "rdh_create");
    if (!target) target = ctx();
    rdh_scene_desc d{};
    d.vertices = reinterpret_cast<const float *>(scene.meshData.vertices.data());
    d.normals = reinterpret_cast<const float *>(scene.meshData.normals.data());
    d.texcoords = reinterpret_cast<const float *>(scene.meshData.texcoords.data());
    d.boundingBoxes = reinterpret_cast<const float *>(scene.boundingBoxes.data());
    for (int i = 0; i < 6; i++) d.bvhNodes[i] = reinterpret_cast<const int32_t *>(scene.BVHNodes[i].data());
    d.bvhSize = scene.BVHSize;
    d.numPrims = (scene.BVHSize + 1) / 2;
    d.materialIds = scene.materialIds.data();
    d.materials = scene.materials.data();
    d.numMaterials = static_cast<int32_t>(scene.materials.size());
    d.numLights = static_cast<int32_t>(scene.lightPrimIds.size());
    d.lightPrimIds = scene.lightPrimIds.data();
    d.lightUnitRadiance = reinterpret_cast<const float *>(scene.lightUnitRadiance.data());
    d.sumLightPowerInv = 1.f / scene.lightSampler.sum;
    d.lightSamplerLength = static_cast<int32_t>(scene.lightSampler.binomDistribs.size());
    d.lightSampler = scene.lightSampler.binomDistribs.data();
    d.sampleSequence = sobol10kx200;
    std::vector<rdh_texture> tex(scene.textures.size());
    for (size_t i = 0; i < tex.size(); i++)
        tex[i] = rdh_texture{scene.textures[i]->width(), scene.textures[i]->height(),
                             reinterpret_cast<const float *>(scene.textures[i]->data())};
    d.numTextures = static_cast<int32_t>(tex.size());
    d.textures = tex.data();
    d.envMapTexId = scene.envMapTexId;
    d.envMapSamplerLength = static_cast<int32_t>(scene.envMapSampler.binomDistribs.size());
    d.envMapSampler = scene.envMapSampler.binomDistribs.data();
    RADISH_CHECK(rdh_scene_upload(target, &d), "DevScene::create");
}
inline void devSceneDestroy() { RADISH_CHECK(rdh_scene_free(ctx()), "DevScene::destroy"); }

// Multi-GPU (no reference counterpart: the reference is single-device, src/preview.cpp:109): one process per GPU.
// Rank 0 fills id128 with commUniqueId and hands it to the other ranks (file, socket, MPI ...); every rank calls commInit
// after DevScene::create.  With RADISH_SHIM_MULTI_GPU the free functions below then render this rank's tiles and gather.
inline void commUniqueId(void *id128) { RADISH_CHECK(rdh_comm_unique_id(id128), "rdh_comm_unique_id"); }
inline void commInit(const void *id128, int rank, int world, int device = -1) {
    if (!ctx()) RADISH_CHECK(rdh_create(&ctx(), device < 0 ? rank : device), "rdh_create");
    RADISH_CHECK(rdh_comm_init(ctx(), id128, rank, world), "rdh_comm_init");
}

#ifdef RADISH_SHIM_ONE_PROCESS_GPUS
// ONE process, n GPUs.  Context 0 is ctx() on device 0 and works on the caller's buffers; every further device has
// a context with the same scene and, owned here, whole-frame mirrors of what the caller keeps on device 0: two images and the
// G-buffer planes.  Every device ends each call with the whole frame (tiles gathered over RCCL), so the running means and the
// temporal reuse see on every device what the single-GPU calls would.
struct Peer {
    rdh_ctx *c = nullptr;
    int device = 0;
    float *direct = nullptr, *indirect = nullptr;
    rdh_gbuffer gb{};
};
inline std::vector<Peer> &peers() {
    static std::vector<Peer> p;
    return p;
}
inline std::vector<rdh_ctx *> allContexts() {
    std::vector<rdh_ctx *> v{ctx()};
    for (Peer &p : peers()) v.push_back(p.c);
    return v;
}
template <typename T>
inline T *peerAlloc(int device, size_t n) {
    void *ptr = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&ptr, n * sizeof(T)) != hipSuccess || hipMemset(ptr, 0, n * sizeof(T)) != hipSuccess)
        check(RDH_ERR_NO_DEVICE, "hipMalloc (peer mirror)", __FILE__, __LINE__);
    return static_cast<T *>(ptr);
}
template <typename SceneT>
inline void commInitAll(const SceneT &scene, const uint32_t *sobol10kx200, int n) {
    const size_t px = (size_t)scene.camera.resolution.x * (size_t)scene.camera.resolution.y;
    for (int dev = 1; dev < n; dev++) {
        Peer p;
        p.device = dev;
        RADISH_CHECK(rdh_create(&p.c, dev), "rdh_create (peer)");
        devSceneCreate(scene, sobol10kx200, p.c);
        p.direct = peerAlloc<float>(dev, 3 * px);
        p.indirect = peerAlloc<float>(dev, 3 * px);
        p.gb.albedo = peerAlloc<float>(dev, 3 * px);
        p.gb.motion = peerAlloc<int32_t>(dev, px);
        for (int k = 0; k < 2; k++) {
            p.gb.normal[k] = peerAlloc<float>(dev, 3 * px);
            p.gb.depth[k] = peerAlloc<float>(dev, px);
            p.gb.primId[k] = peerAlloc<int32_t>(dev, px);
        }
        peers().push_back(p);
    }
    if (hipSetDevice(0) != hipSuccess) check(RDH_ERR_NO_DEVICE, "hipSetDevice", __FILE__, __LINE__);
    std::vector<rdh_ctx *> all = allContexts();
    RADISH_CHECK(rdh_comm_init_all(all.data(), (int)all.size()), "rdh_comm_init_all");
}
#endif

template <typename GBufferT>
inline rdh_gbuffer toC(const GBufferT &g) {
    static_assert(sizeof(GBufferT) == sizeof(rdh_gbuffer), "GBuffer layout (src/gBuffer.h:42-57)");
    rdh_gbuffer out;
    std::memcpy(&out, &g, sizeof(out));
    return out;
}

}  // namespace radish_shim

#ifndef RADISH_SHIM_HELPERS_ONLY
// ---- the reference's free functions, same signatures (src/pathtrace.h:19-23, src/restir.h:103-106) -----------------
inline void pathTraceInit() { RADISH_CHECK(rdh_synchronize(radish_shim::ctx()), "pathTraceInit"); }
inline void pathTraceFree() {}

inline void pathTrace(glm::vec3 *directIllum, glm::vec3 *indirectIllum, int iter) {
    rdh_ctx *c = radish_shim::ctx();
    RADISH_CHECK(rdh_set_camera(c, &State::scene->camera), "pathTrace");
#ifdef RADISH_SHIM_ONE_PROCESS_GPUS
    std::vector<rdh_ctx *> all = radish_shim::allContexts();
    std::vector<float *> dF{reinterpret_cast<float *>(directIllum)}, iF{reinterpret_cast<float *>(indirectIllum)};
    for (radish_shim::Peer &p : radish_shim::peers()) {
        RADISH_CHECK(rdh_set_camera(p.c, &State::scene->camera), "pathTrace");
        dF.push_back(p.direct);
        iF.push_back(p.indirect);
    }
    RADISH_CHECK(rdh_path_trace_gathered_all(all.data(), (int)all.size(), dF.data(), iF.data(), iter, State::looper, Settings::traceDepth,
                                             RDH_PT_AUTO),
                 "pathTrace");
    for (radish_shim::Peer &p : radish_shim::peers()) RADISH_CHECK(rdh_synchronize(p.c), "pathTrace");
#elif defined(RADISH_SHIM_MULTI_GPU)
    RADISH_CHECK(rdh_path_trace_gathered(c, reinterpret_cast<float *>(directIllum), reinterpret_cast<float *>(indirectIllum), iter,
                                         State::looper, Settings::traceDepth, RDH_PT_AUTO),
                 "pathTrace");
#else
    RADISH_CHECK(rdh_path_trace(c, reinterpret_cast<float *>(directIllum), reinterpret_cast<float *>(indirectIllum), iter,
                                State::looper, Settings::traceDepth, RDH_PT_AUTO),
                 "pathTrace");
#endif
    RADISH_CHECK(rdh_synchronize(c), "pathTrace");
    float ms = 0.f;
    if (rdh_last_kernel_ms(c, &ms) == RDH_OK) std::printf("PT runtime%.3f ms\n", ms);  // src/pathtrace.cu:374
    State::looper = (State::looper + 1) % 10000;                                          // :380-381
}

inline void pathTraceDirect(glm::vec3 *directIllum, int iter) {
    rdh_ctx *c = radish_shim::ctx();
    RADISH_CHECK(rdh_set_camera(c, &State::scene->camera), "pathTraceDirect");
    RADISH_CHECK(rdh_path_trace_direct(c, reinterpret_cast<float *>(directIllum), iter, State::looper, 0), "pathTraceDirect");
    RADISH_CHECK(rdh_synchronize(c), "pathTraceDirect");
    State::looper = (State::looper + 1) % 10000;
}

inline void ReSTIRInit() {
    rdh_ctx *c = radish_shim::ctx();
    RADISH_CHECK(rdh_set_camera(c, &State::scene->camera), "ReSTIRInit");
    RADISH_CHECK(rdh_restir_init(c), "ReSTIRInit");
#ifdef RADISH_SHIM_ONE_PROCESS_GPUS
    for (radish_shim::Peer &p : radish_shim::peers()) {
        RADISH_CHECK(rdh_set_camera(p.c, &State::scene->camera), "ReSTIRInit");
        RADISH_CHECK(rdh_restir_init(p.c), "ReSTIRInit");
    }
#endif
}
inline void ReSTIRFree() {
    RADISH_CHECK(rdh_restir_free(radish_shim::ctx()), "ReSTIRFree");
#ifdef RADISH_SHIM_ONE_PROCESS_GPUS
    for (radish_shim::Peer &p : radish_shim::peers()) RADISH_CHECK(rdh_restir_free(p.c), "ReSTIRFree");
#endif
}

inline void ReSTIRDirect(glm::vec3 *directIllum, int iter, const GBuffer &gBuffer) {
    rdh_ctx *c = radish_shim::ctx();
    rdh_gbuffer g = radish_shim::toC(gBuffer);
    rdh_restir_params p{Settings::reservoirReuse, 32, 5, 20, 1};  // RESERVOIR_SIZE, restir.cu:87, :168, restir.h:21
    RADISH_CHECK(rdh_set_camera(c, &State::scene->camera), "ReSTIR Direct");
#ifdef RADISH_SHIM_ONE_PROCESS_GPUS
    {
        std::vector<rdh_ctx *> all = radish_shim::allContexts();
        std::vector<float *> dF{reinterpret_cast<float *>(directIllum)};
        std::vector<rdh_gbuffer> gbs{g};
        for (radish_shim::Peer &p : radish_shim::peers()) {
            RADISH_CHECK(rdh_set_camera(p.c, &State::scene->camera), "ReSTIR Direct");
            p.gb.frameIdx = g.frameIdx;  // GBuffer::update stays the reference's: the mirrors follow the caller's G-buffer
            std::memcpy(p.gb.lastCam, g.lastCam, sizeof(g.lastCam));
            p.gb.width = g.width;
            p.gb.height = g.height;
            dF.push_back(p.direct);
            gbs.push_back(p.gb);
        }
        RADISH_CHECK(rdh_restir_direct_gathered_all(all.data(), (int)all.size(), dF.data(), iter, State::looper, gbs.data(), &p, 0), "ReSTIR Direct");
        for (radish_shim::Peer &q : radish_shim::peers()) RADISH_CHECK(rdh_synchronize(q.c), "ReSTIR Direct");
    }
#elif defined(RADISH_SHIM_MULTI_GPU)
    RADISH_CHECK(rdh_restir_direct_gathered(c, reinterpret_cast<float *>(directIllum), iter, State::looper, &g, &p, 0), "ReSTIR Direct");
#else
    RADISH_CHECK(rdh_restir_direct(c, reinterpret_cast<float *>(directIllum), iter, State::looper, &g, &p, 0), "ReSTIR Direct");
#endif
    RADISH_CHECK(rdh_synchronize(c), "ReSTIR Direct");
    State::looper = (State::looper + 1) % 10000;
}

// copyImageToPBO (src/pathtrace.h:25-29, src/pathtrace.cu:120-147): asynchronous, like the reference's bare launches.
inline void copyImageToPBO(uchar4 *devPBO, glm::vec3 *devImage, int width, int height, int toneMapping, float scale = 1.f) {
    RADISH_CHECK(rdh_copy_image_to_pbo(radish_shim::ctx(), devPBO, devImage, width, height, 0, toneMapping, scale), "copyImageToPBO");
}
inline void copyImageToPBO(uchar4 *devPBO, glm::vec2 *devImage, int width, int height) {
    RADISH_CHECK(rdh_copy_image_to_pbo(radish_shim::ctx(), devPBO, devImage, width, height, 1, 0, 1.f), "copyImageToPBO");
}
inline void copyImageToPBO(uchar4 *devPBO, float *devImage, int width, int height) {
    RADISH_CHECK(rdh_copy_image_to_pbo(radish_shim::ctx(), devPBO, devImage, width, height, 2, 0, 1.f), "copyImageToPBO");
}
inline void copyImageToPBO(uchar4 *devPBO, int *devImage, int width, int height) {
    RADISH_CHECK(rdh_copy_image_to_pbo(radish_shim::ctx(), devPBO, devImage, width, height, 3, 0, 1.f), "copyImageToPBO");
}

// GBuffer::render (src/gBuffer.cu:83-103); create/destroy/update stay the reference's (plain allocation / a swap).
inline void GBuffer::render(DevScene *, const Camera &cam) {
    rdh_ctx *c = radish_shim::ctx();
    rdh_gbuffer g = radish_shim::toC(*this);
    RADISH_CHECK(rdh_set_camera(c, &cam), "renderGBuffer");
#ifdef RADISH_SHIM_ONE_PROCESS_GPUS  // every device renders the records of ITS tiles, one grouped all-gather completes the planes
    {
        std::vector<rdh_ctx *> all = radish_shim::allContexts();
        std::vector<rdh_gbuffer> gbs{g};
        RADISH_CHECK(rdh_gbuffer_render(c, &g, RDH_PT_PARTITION_GBUFFER), "renderGBuffer");
        for (radish_shim::Peer &p : radish_shim::peers()) {
            RADISH_CHECK(rdh_set_camera(p.c, &cam), "renderGBuffer");
            p.gb.frameIdx = g.frameIdx;
            std::memcpy(p.gb.lastCam, g.lastCam, sizeof(g.lastCam));
            p.gb.width = g.width;
            p.gb.height = g.height;
            RADISH_CHECK(rdh_gbuffer_render(p.c, &p.gb, RDH_PT_PARTITION_GBUFFER), "renderGBuffer");
            gbs.push_back(p.gb);
        }
        RADISH_CHECK(rdh_gbuffer_exchange_all(all.data(), (int)all.size(), gbs.data()), "renderGBuffer");
        for (radish_shim::Peer &p : radish_shim::peers()) RADISH_CHECK(rdh_synchronize(p.c), "renderGBuffer");
    }
#elif defined(RADISH_SHIM_MULTI_GPU)  // this rank's tiles only, then one all-gather of 36 B per pixel
    RADISH_CHECK(rdh_gbuffer_render(c, &g, RDH_PT_PARTITION_GBUFFER), "renderGBuffer");
    RADISH_CHECK(rdh_gbuffer_exchange(c, &g), "renderGBuffer");
#else
    RADISH_CHECK(rdh_gbuffer_render(c, &g, 0), "renderGBuffer");
#endif
    RADISH_CHECK(rdh_synchronize(c), "renderGBuffer");
    float ms = 0.f;
    if (rdh_last_kernel_ms(c, &ms) == RDH_OK) std::printf("GBuffer runtime%.3f ms\n", ms);  // src/gBuffer.cu:98
}
// ---- denoiser.cu (src/denoiser.h:16-81): the filter classes keep the reference's declarations; these are their bodies ----
#ifdef RADISH_SHIM_WITH_DENOISER  // needs "denoiser.h" (EAWaveletFilter, LeveledEAWFilter, SpatioTemporalFilter) and hip_runtime_api.h
namespace radish_shim {
template <typename T>
inline T *devAlloc(size_t n) {  // cudaMalloc<T> (src/cudaUtil.h)
    void *p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess || hipMemset(p, 0, n * sizeof(T)) != hipSuccess)
        check(RDH_ERR_NO_DEVICE, "hipMalloc", __FILE__, __LINE__);
    return static_cast<T *>(p);
}
template <typename T>
inline void devFree(T *&p) {  // cudaSafeFree (src/cudaUtil.h)
    if (p && hipFree(p) != hipSuccess) check(RDH_ERR_NO_DEVICE, "hipFree", __FILE__, __LINE__);
    p = nullptr;
}
inline float *f(glm::vec3 *p) { return reinterpret_cast<float *>(p); }
}  // namespace radish_shim

inline void modulateAlbedo(glm::vec3 *devImage, const GBuffer &gBuffer) {  // denoiser.cu:363-371
    rdh_gbuffer g = radish_shim::toC(gBuffer);
    RADISH_CHECK(rdh_denoise_modulate(radish_shim::ctx(), radish_shim::f(devImage), &g), "modulate");
}
inline void addImage(glm::vec3 *devImage, glm::vec3 *in, int width, int height) {  // :373-378
    RADISH_CHECK(rdh_denoise_add(radish_shim::ctx(), radish_shim::f(devImage), radish_shim::f(devImage), radish_shim::f(in), width, height), "add");
}
inline void addImage(glm::vec3 *out, glm::vec3 *in1, glm::vec3 *in2, int width, int height) {  // :380-386
    RADISH_CHECK(rdh_denoise_add(radish_shim::ctx(), radish_shim::f(out), radish_shim::f(in1), radish_shim::f(in2), width, height), "add");
}
inline void EAWaveletFilter::filter(glm::vec3 *colorOut, glm::vec3 *colorIn, const GBuffer &gBuffer, const Camera &cam, int level) {
    rdh_gbuffer g = radish_shim::toC(gBuffer);  // :388-397
    RADISH_CHECK(rdh_denoise_eaw(radish_shim::ctx(), radish_shim::f(colorOut), radish_shim::f(colorIn), &g, &cam, sigLumin, sigNormal,
                                 sigDepth, level), "EAW Filter");
}
inline void EAWaveletFilter::filter(glm::vec3 *colorOut, glm::vec3 *colorIn, float *varianceOut, float *varianceIn, float *filteredVar,
                                    const GBuffer &gBuffer, const Camera &cam, int level) {
    rdh_gbuffer g = radish_shim::toC(gBuffer);  // :399-409
    RADISH_CHECK(rdh_denoise_svgf(radish_shim::ctx(), radish_shim::f(colorOut), radish_shim::f(colorIn), varianceOut, varianceIn,
                                  filteredVar, &g, &cam, sigLumin, sigNormal, sigDepth, level), "SVGF Filter");
}
inline void LeveledEAWFilter::create(int width, int height, int level) {  // :411-415
    this->level = level;
    waveletFilter = EAWaveletFilter(width, height, 64.f, .2f, 1.f);
    tmpImg = radish_shim::devAlloc<glm::vec3>((size_t)width * height);
}
inline void LeveledEAWFilter::destroy() { radish_shim::devFree(tmpImg); }
inline void LeveledEAWFilter::filter(glm::vec3 *&colorOut, glm::vec3 *colorIn, const GBuffer &gBuffer, const Camera &cam) {
    waveletFilter.filter(colorOut, colorIn, gBuffer, cam, 0);  // :419-434
    for (int lv = 1; lv <= 4; lv++) {
        waveletFilter.filter(tmpImg, colorOut, gBuffer, cam, lv);
        std::swap(colorOut, tmpImg);
    }
}
inline void SpatioTemporalFilter::create(int width, int height, int level) {  // :436-448
    this->level = level;
    for (int i = 0; i < 2; i++) {
        accumColor[i] = radish_shim::devAlloc<glm::vec3>((size_t)width * height);
        accumMoment[i] = radish_shim::devAlloc<glm::vec3>((size_t)width * height);
    }
    variance = radish_shim::devAlloc<float>((size_t)width * height);
    waveletFilter = EAWaveletFilter(width, height, 4.f, 128.f, 1.f);
    tmpColor = radish_shim::devAlloc<glm::vec3>((size_t)width * height);
    tmpVar = radish_shim::devAlloc<float>((size_t)width * height);
    filteredVar = radish_shim::devAlloc<float>((size_t)width * height);
}
inline void SpatioTemporalFilter::destroy() {  // :450-459
    for (int i = 0; i < 2; i++) { radish_shim::devFree(accumColor[i]); radish_shim::devFree(accumMoment[i]); }
    radish_shim::devFree(variance); radish_shim::devFree(tmpColor); radish_shim::devFree(tmpVar); radish_shim::devFree(filteredVar);
}
inline void SpatioTemporalFilter::temporalAccumulate(glm::vec3 *colorIn, const GBuffer &gBuffer) {  // :461-485
    rdh_gbuffer g = radish_shim::toC(gBuffer);
    RADISH_CHECK(rdh_denoise_temporal_accumulate(radish_shim::ctx(), radish_shim::f(accumColor[frameIdx]), radish_shim::f(accumColor[frameIdx ^ 1]),
                                                 radish_shim::f(accumMoment[frameIdx]), radish_shim::f(accumMoment[frameIdx ^ 1]),
                                                 radish_shim::f(colorIn), &g, firstTime ? 1 : 0), "SpatioTemporalFilter::temporalAccumulate");
    firstTime = false;
}
inline void SpatioTemporalFilter::estimateVariance() {  // :487-509
    RADISH_CHECK(rdh_denoise_estimate_variance(radish_shim::ctx(), variance, radish_shim::f(accumMoment[frameIdx]), waveletFilter.width,
                                               waveletFilter.height), "SpatioTemporalFilter::estimateVariance");
}
inline void SpatioTemporalFilter::filterVariance() {  // :511-523
    RADISH_CHECK(rdh_denoise_filter_variance(radish_shim::ctx(), filteredVar, variance, waveletFilter.width, waveletFilter.height),
                 "SpatioTemporalFilter::filterVariance");
}
inline void SpatioTemporalFilter::filter(glm::vec3 *&colorOut, glm::vec3 *colorIn, const GBuffer &gBuffer, const Camera &cam) {
    temporalAccumulate(colorIn, gBuffer);  // :525-558, swap for swap
    estimateVariance();
    filterVariance();
    waveletFilter.filter(colorOut, accumColor[frameIdx], tmpVar, variance, filteredVar, gBuffer, cam, 0);
    std::swap(colorOut, accumColor[frameIdx]);
    std::swap(tmpVar, variance);
    filterVariance();
    waveletFilter.filter(colorOut, accumColor[frameIdx], tmpVar, variance, filteredVar, gBuffer, cam, 1);
    std::swap(tmpVar, variance);
    for (int lv = 2; lv <= 4; lv++) {
        filterVariance();
        waveletFilter.filter(tmpColor, colorOut, tmpVar, variance, filteredVar, gBuffer, cam, lv);
        std::swap(tmpColor, colorOut);
        std::swap(tmpVar, variance);
    }
}
inline void SpatioTemporalFilter::nextFrame() { frameIdx ^= 1; }  // :560
#endif  // RADISH_SHIM_WITH_DENOISER
#endif  // RADISH_SHIM_HELPERS_ONLY
