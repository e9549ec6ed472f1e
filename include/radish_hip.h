/* include/radish_hip.h — C ABI of libradish_hip.so, the MI355X (gfx950) implementation of Radish's per-pixel
 * path-tracing inner loop.
 *
 * The reference has no FFI layer: its render API is a set of C++ free functions that read process globals
 * (SURVEY.md §8b).  Each entry point below names the reference function whose body it replaces; the
 * source-compatible C++ shim with the reference's exact signatures is radish_pt_amd/csrc/radish_shim.hpp, and
 * INTEGRATION.md shows the binding a Radish maintainer would add.
 *
 * Conventions: every function returns 0 on success or a negative RDH_ERR_* / positive hipError_t code, and
 * rdh_last_error() gives the message (the reference prints and exit()s: src/cudaUtil.h:16-34 — the shim
 * reproduces that).  All `d_*` pointers are DEVICE pointers owned by the caller; scene arrays passed to
 * rdh_scene_upload are HOST pointers in the reference's DevScene layout.  A context is bound to one device and
 * one stream and is not thread-safe; calls are asynchronous on that stream unless stated otherwise.
 */
#ifndef RADISH_HIP_H
#define RADISH_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDH_OK 0
#define RDH_ERR_ARGS (-1)        /* null / inconsistent arguments                         */
#define RDH_ERR_NO_SCENE (-2)    /* call needs rdh_scene_upload + rdh_set_camera first    */
#define RDH_ERR_UNSUPPORTED (-3) /* a combination that is not built (e.g. a diagnostic entry in a non-diagnostic build) */
#define RDH_ERR_NO_DEVICE (-4)   /* no usable HIP device — there is NO CPU fallback       */
#define RDH_ERR_STATE (-5)       /* e.g. rdh_restir_direct before rdh_restir_init         */
#define RDH_ERR_COMM (-6)        /* RCCL missing or an nccl* call failed (rdh_last_error has ncclGetErrorString's text) */

typedef struct rdh_ctx rdh_ctx;

/* One texture: DevTextureObj (src/image.h:89-91) with a HOST pointer; texels are glm::vec3, row-major (Image::data()). */
typedef struct rdh_texture { int32_t width, height; const float *data; } rdh_texture;

/* Host arrays in the reference's DevScene layout (src/scene.h:494-517).
 * Replaces the argument of DevScene::create(const Scene&) (src/scene.cpp:461-551). */
typedef struct rdh_scene_desc {
    const float *vertices;       /* glm::vec3[3*numPrims], world-space triangle soup (src/scene.cpp:198-222) */
    const float *normals;        /* glm::vec3[3*numPrims]                                                     */
    const float *texcoords;      /* glm::vec2[3*numPrims]                                                     */
    const float *boundingBoxes;  /* AABB[bvhSize] = {pMin, pMax} (src/bvh.h:157-158)                           */
    const int32_t *bvhNodes[6];  /* MTBVHNode[bvhSize] = {primitiveId, boundingBoxId, nextNodeIfMiss} (:167-169) */
    int32_t bvhSize;             /* 2*numPrims-1                                                               */
    int32_t numPrims;
    const int32_t *materialIds;  /* int[numPrims]                                                              */
    const void *materials;       /* Material[numMaterials], 44 B each (src/material.h:276-286)                 */
    int32_t numMaterials;
    int32_t numLights;
    const int32_t *lightPrimIds;      /* int[numLights]                                                        */
    const float *lightUnitRadiance;   /* glm::vec3[numLights]                                                  */
    float sumLightPowerInv;           /* 1 / lightSampler.sum (src/scene.cpp:527)                              */
    int32_t lightSamplerLength;       /* DevDiscreteSampler1D::length (== numLights without an env map)        */
    const void *lightSampler;         /* BinomialDistrib<float>[length] = {float prob; int failId}             */
    const uint32_t *sampleSequence;   /* uint32[10000][200] Sobol table (src/scene.cpp:543-548)                */
    /* Textures and environment map (src/scene.cpp:463-486, 529-532).  Material map ids index `textures`
     * (-1 = NullTextureId, -2 = ProceduralTexId for baseColorMapId only). */
    int32_t numTextures;
    const struct rdh_texture *textures;
    int32_t envMapTexId;              /* Scene::envMapTexId: -1 = no environment map                           */
    int32_t envMapSamplerLength;      /* envMapSampler.length = width*height of the env map, 0 = none;
                                         with an env map the LAST lightSampler entry is the env map (scene.cpp:163) */
    const void *envMapSampler;        /* BinomialDistrib<float>[envMapSamplerLength]                           */
} rdh_scene_desc;

/* Byte-for-byte the reference's GBuffer (src/gBuffer.h:42-57, 272 B) with DENOISER_ENCODE_NORMAL=false,
 * DENOISER_ENCODE_POSITION=true: device pointers owned by the caller (GBuffer::create, src/denoiser.cu:329-359). */
typedef struct rdh_gbuffer {
    float *albedo;     /* glm::vec3[w*h]                                         */
    float *normal[2];  /* glm::vec3[w*h] x2 (double-buffered by frameIdx)        */
    int32_t *motion;   /* int[w*h]                                               */
    float *depth[2];   /* float[w*h] x2                                          */
    int32_t *primId[2];/* int[w*h] x2 — holds the MATERIAL id, -2 light, -1 miss */
    int32_t frameIdx;
    uint8_t lastCam[196];
    int32_t width, height;
} rdh_gbuffer;

typedef struct rdh_hit { int32_t primId; float u, v, t; } rdh_hit;

/* Exact work counters of the launches since the last reset (device atomics, one add per wave). */
typedef struct rdh_counters {
    uint64_t closestRays; /* DevScene::intersect calls       */
    uint64_t anyRays;     /* DevScene::testOcclusion calls   */
    uint64_t nodeVisits;  /* AABB tests                      */
    uint64_t triTests;    /* triangle tests                  */
    uint64_t closestHits;
} rdh_counters;

/* rdh_path_trace flags */
#define RDH_PT_MEGAKERNEL 0u   /* one lane per pixel, whole path in one launch (the reference's structure)  */
#define RDH_PT_WAVEFRONT 1u    /* raygen / extend / shade / connect queues with wave64 ballot compaction    */
#define RDH_PT_SORT_MATERIAL 2u/* wavefront only: each wave bins the 1 024 hit records of its packet by BSDF type before shading them */
#define RDH_PT_COUNT 4u        /* maintain rdh_counters (adds atomics; leave off when timing)               */
#define RDH_PT_PERSISTENT 16u  /* one persistent launch: per-lane state machine with lane refill (kernels_persist.h) */
#define RDH_PT_ONE_LANE_PER_PIXEL 64u /* rdh_gbuffer_render: the one-lane-per-pixel kernel (k_gbuffer) instead of the persistent
                                         lane-refill one (k_gbuffer_persistent)                                              */
#define RDH_PT_MEGA_GBUFFER RDH_PT_ONE_LANE_PER_PIXEL
#define RDH_PT_WG_PER_RAY 256u /* rdh_trace_closest / rdh_trace_occluded: one 1 024-thread workgroup per ray (the routine the
                                  G-buffer uses for literal-class rays); for tests            */
#define RDH_PT_NO_DEFER 128u   /* rdh_gbuffer_render: trace literal-class rays where they are generated (one wave each)
                                  instead of setting them aside for the workgroup-per-ray launch (k_gbuffer_literal)   */
#define RDH_PT_PARTITION_GBUFFER 512u /* rdh_gbuffer_render on a tile partition: render the records of THIS rank's tiles only
                                  (complete the planes with rdh_gbuffer_exchange*); default: every rank renders the whole frame */
#define RDH_PT_WF_SUBFRAMES 2048u /* wavefront only: three sub-frames (8x8 blocks dealt round robin) as three pipelines on three streams, so
                                  that one pipeline's stage tails overlap the others' stage bodies (frames of >= 2 048 blocks) */
#define RDH_PT_RESTIR_FUSED 1024u /* rdh_restir_direct: round 1's pass 1, one lane per pixel with both walks inside the kernel
                                  (k_restir_pass1), instead of raygen / walk / RIS / walk / resolve (default; same results) */
#define RDH_PT_WF_SMALL_LISTS 4096u /* for tests.  Wavefront: the per-stage lists of literal-class rays hold 4 entries, so that the
                                  overflow path (such rays stay in the ordinary queues) runs.  rdh_gbuffer_render / rdh_restir_direct: a
                                  wave walks its block as a packet for 4 visits only (256 normally), so that every block goes through
                                  the hand-over to per-lane walks (traverse.h, packetWalk) */
#define RDH_PT_PAIRS 32768u    /* per-lane walks over SIBLING PAIRS (one 64-byte record per inner node, shared by the six orderings: a lane
                                  that enters a node fetches both children and tests both boxes in one round trip; the far child is
                                  re-checked when the walk reaches it) instead of the six threaded arrays; same records, same counters.
                                  Default: on, whenever the six uploaded arrays are orderings of ONE binary tree at most 2 048 levels
                                  deep (every tree Radish's builder makes is one tree; n coincident triangles make a chain of n - 1
                                  levels).  Otherwise the threaded arrays are walked: silently by the frame entries, while
                                  rdh_trace_closest / rdh_trace_occluded (with RDH_PT_PERSISTENT) return RDH_ERR_UNSUPPORTED when
                                  the flag asks for pairs that do not exist */
#define RDH_PT_NO_PAIRS 16384u /* never walk the sibling pairs */
#define RDH_PT_NO_PACKETS 8192u /* rdh_gbuffer_render, rdh_restir_direct: primary rays lane by lane (the lane-refill walkers) instead of as
                                  PACKETS — one 8x8 pixel block per wave, walked through the threaded order together with one uniform
                                  node load per visit (traverse.h, packetWalk); same records, same counters.  (rdh_path_trace keeps its
                                  primary rays in trace(0): measured) */
#define RDH_PT_AUTO 65536u     /* rdh_path_trace / rdh_path_trace_gathered*: the library picks the structure by what this launch holds —
                                  the wavefront pipeline with material sort and three sub-frames for a big tree (>= 100 000 nodes) and a
                                  big share of the frame (>= 12 000 8x8 blocks: a 1080p frame, or half of one), else the persistent kernel
                                  (measured: DESIGN 5d, 8).  Same pixels either way; other structure bits are ignored with it */
#define RDH_PT_NO_SCHEDULE 32u /* persistent only: ignore the longest-paths-first block order of the previous launch   */
#define RDH_PT_PROFILE 8u      /* bracket each launch of the traversal kernel (k_pt_persistent, k_wf_trace, or the megakernel)
                                  with hipEvents on the context's stream; read with rdh_profile_read.  With
                                  RDH_PT_WF_SUBFRAMES the pipelines' launches overlap by design: ONE pair brackets the whole
                                  frame (before the fork to after the join) and rdh_profile_read counts frames            */

/* ReSTIR reuse mask = ReservoirReuse (src/common.h:41-48) */
#define RDH_REUSE_TEMPORAL 1
#define RDH_REUSE_SPATIAL 2

typedef struct rdh_restir_params {
    int32_t reuseMask;    /* Settings::reservoirReuse                                              */
    int32_t risCount;     /* RESERVOIR_SIZE = 32 (src/restir.h:9)                                   */
    int32_t numSpatial;   /* 5 in the reference (src/restir.cu:87); BASELINE config 4 also asks for 4 */
    int32_t temporalClamp;/* 20 (src/restir.cu:168)                                                 */
    int32_t faithfulRIS;  /* 1 = Reservoir::update's truthiness test (src/restir.h:21), 0 = corrected */
} rdh_restir_params;

/* ---- lifetime ---------------------------------------------------------------------------------------- */
int rdh_create(rdh_ctx **out, int device);
void rdh_destroy(rdh_ctx *ctx);
const char *rdh_last_error(const rdh_ctx *ctx);
/* Launch on an existing hipStream_t (e.g. torch's current stream).  NULL selects HIP's default (null) stream; a
 * context on which this is never called uses a private non-blocking stream. */
int rdh_set_stream(rdh_ctx *ctx, void *hipStream);
int rdh_synchronize(rdh_ctx *ctx);

/* ---- scene / camera -------------------------------------------------------------------------------- */
/* Replaces DevScene::create (src/scene.cpp:461-551): uploads and re-lays-out the scene (32-byte threaded nodes,
 * 48-byte triangle records, packed light records).  Blocking. */
int rdh_scene_upload(rdh_ctx *ctx, const rdh_scene_desc *desc);
/* Replaces DevScene::destroy (src/scene.cpp:553-574). */
int rdh_scene_free(rdh_ctx *ctx);
/* The `Camera cam` kernel argument of every reference kernel (State::scene->camera, src/pathtrace.cu:356). */
int rdh_set_camera(rdh_ctx *ctx, const void *camera196);

/* ---- multi-GPU tile partition (no reference counterpart) --------------------------------- */
/* The frame is cut into tileSize x tileSize pixel tiles, tile t belongs to rank t % world.  With world > 1 the
 * image arguments of the render calls are PACKED per-rank buffers: float[tilesPerRank][tileSize*tileSize][3],
 * tilesPerRank = rdh_tiles_per_rank(); gather them over RCCL and call rdh_untile.  world == 1 renders straight
 * into frame layout (y*W+x), which is the reference's. */
int rdh_set_partition(rdh_ctx *ctx, int rank, int world, int tileSize);
int rdh_tiles_per_rank(const rdh_ctx *ctx);
/* d_gathered: float[world][tilesPerRank][tileSize^2][3] (all-gather output) -> d_frame: float[H*W][3]. */
int rdh_untile(rdh_ctx *ctx, const float *d_gathered, float *d_frame);

/* ---- the hot path ---------------------------------------------------------------------------------- */
/* Replaces pathTrace(glm::vec3*, glm::vec3*, int) (src/pathtrace.cu:351-385) minus the State::looper increment
 * (the caller / shim owns the globals).  looper = State::looper, maxDepth = Settings::traceDepth. */
int rdh_path_trace(rdh_ctx *ctx, float *d_directIllum, float *d_indirectIllum, int iter, int looper, int maxDepth,
                   uint32_t flags);
/* Replaces pathTraceDirect(glm::vec3*, int) (src/pathtrace.cu:387-407). */
int rdh_path_trace_direct(rdh_ctx *ctx, float *d_directIllum, int iter, int looper, uint32_t flags);
/* Replaces GBuffer::render(DevScene*, const Camera&) (src/gBuffer.cu:83-103). */
int rdh_gbuffer_render(rdh_ctx *ctx, const rdh_gbuffer *gb, uint32_t flags);
/* Replaces ReSTIRInit / ReSTIRFree (src/restir.cu:235-251): three DirectReservoir[w*h] buffers, zeroed. */
int rdh_restir_init(rdh_ctx *ctx);
int rdh_restir_free(rdh_ctx *ctx);
/* Replaces ReSTIRDirect(glm::vec3*, int, const GBuffer&) (src/restir.cu:205-233) incl. the reservoir swap and
 * the first-frame flag; two launches instead of one so that spatial reuse is race-free. */
int rdh_restir_direct(rdh_ctx *ctx, float *d_directIllum, int iter, int looper, const rdh_gbuffer *gb,
                      const rdh_restir_params *params, uint32_t flags);
/* Tile partition (world > 1): rdh_gbuffer_render always renders the WHOLE frame on every rank (temporal and spatial
 * reuse read the G-buffer at arbitrary / neighbouring pixels); rdh_restir_direct shades this rank's tiles into a
 * PACKED image buffer (as rdh_path_trace does), computing pass 1 on an 8-pixel apron as well so that pass 2 is local.
 * Temporal reuse needs last frame's reservoirs of the whole frame: after each rdh_restir_direct call
 * rdh_restir_exchange_pack(d_packed: float[tilesPerRank][tile^2][9]) → all-gather → rdh_restir_exchange_unpack(d_gathered:
 * float[world][tilesPerRank][tile^2][9]).  Results are bit-identical to the single-GPU frame. */
int rdh_restir_exchange_pack(rdh_ctx *ctx, float *d_packed);
int rdh_restir_exchange_unpack(rdh_ctx *ctx, const float *d_gathered);
/* G-buffer on a tile partition without replicated work: rdh_gbuffer_render(..., RDH_PT_PARTITION_GBUFFER) writes the records of
 * this rank's tiles (one primary ray per OWN pixel instead of per frame pixel); rdh_gbuffer_exchange_pack gathers them into
 * d_packed: float[tilesPerRank][tile^2][9] = {albedo.xyz, normal.xyz, motion, depth, primId} (ints as their bits) → all-gather →
 * rdh_gbuffer_exchange_unpack(d_gathered: float[world][tilesPerRank][tile^2][9]) scatters every rank's records into the
 * current planes (gb->frameIdx) of this rank's G-buffer: 36 B per pixel of the frame received per rank and frame. */
int rdh_gbuffer_exchange_pack(rdh_ctx *ctx, const rdh_gbuffer *gb, float *d_packed);
int rdh_gbuffer_exchange_unpack(rdh_ctx *ctx, const rdh_gbuffer *gb, const float *d_gathered);

/* ---- collectives from C++: RCCL over xGMI (no reference counterpart: the reference is single-device,
 *      /root/reference/src/preview.cpp:109 cudaGLSetGLDevice(0); frame loop /root/reference/src/main.cpp:163-202) ------------
 * One process (or host thread) per GPU.  Rank 0 calls rdh_comm_unique_id and hands the 128 bytes (== ncclUniqueId) to the other
 * ranks by any channel; every rank then calls rdh_comm_init on its context, which creates the communicator (ncclCommInitRank)
 * and sets the tile partition to (rank, world, current tile size).  RCCL is bound at run time (dlopen of the copy the process
 * already holds, else librccl.so.1; RADISH_RCCL_LIB overrides) so the library itself links only the HIP runtime.
 * Every collective below is enqueued on the context's stream and returns without synchronising.
 *   rdh_allgather_tiles          d_packed float[tilesPerRank][tile^2][3] → ONE ncclAllGather → k_untile → d_frame float[H*W][3]
 *   rdh_path_trace_gathered      pathTrace for N GPUs with the reference's signature semantics: whole-frame images in and out on
 *                                every rank (the running mean reads the caller's frame); renders this rank's tiles, then two
 *                                all-gathers (direct, indirect).  Works with world == 1 (RCCL accepts a single rank).
 *   rdh_restir_exchange          this rank's pre-spatial reservoirs of the frame just rendered → all-gather → every rank's
 *                                `last` reservoir buffer holds the whole frame (next frame's temporal reuse): 36 B/px
 *   rdh_restir_direct_gathered   ReSTIRDirect for N GPUs: whole-frame image in and out, includes rdh_restir_exchange
 *   rdh_gbuffer_exchange         pack → all-gather → unpack of the G-buffer records (see above)
 * ONE process driving n GPUs (the reference's host model: one process, one frame loop, /root/reference/src/main.cpp:71-116,
 * :163-202; SURVEY §8e "single process ... one host thread + stream per device"):
 *   rdh_comm_init_all            ctxs[i] (created on n DIFFERENT devices) becomes rank i of n: the n ncclCommInitRank calls are
 *                                made inside one ncclGroupStart/End, so one host thread can create them (rdh_comm_init blocks
 *                                until every rank has called it and therefore needs a thread or process per rank)
 *   rdh_path_trace_gathered_all  rdh_path_trace_gathered for all n contexts from one thread: n renders enqueued on n streams,
 *                                then per image ONE RCCL group with the n all-gathers, then n un-tile kernels.
 *                                d_directFrames[i] / d_indirectFrames[i]: whole-frame images on ctxs[i]'s device.
 *   rdh_gbuffer_exchange_all / rdh_restir_direct_gathered_all   the same for the ReSTIR frame: gbs[i] is context i's G-buffer
 *                                (planes on its device); one RCCL group per collective.
 * ReSTIR's exchanges OVERLAP rendering (default; rdh_comm_set_overlap(ctx, 0) puts everything back on the render stream): every
 * collective of rdh_gbuffer_exchange / rdh_restir_direct_gathered / rdh_restir_exchange is issued on a communication stream of
 * the context, in one order.  The G-buffer gather runs beside pass 1's ray generation, walks and RIS — rdh_restir_direct waits
 * for it where it first reads the planes (resolve step) — and the reservoir gather runs beside the NEXT frame's G-buffer pass and
 * pass 1.  The gathered IMAGE is complete in render-stream order when rdh_restir_direct_gathered returns.  Anything else that
 * reads G-buffer planes after rdh_gbuffer_exchange (the denoisers and rdh_copy_image_to_pbo do it themselves) calls
 * rdh_comm_join first: the render stream then waits for the exchanges in flight.  rdh_synchronize waits for both streams.        */
int rdh_comm_unique_id(void *id128);
int rdh_comm_init(rdh_ctx *ctx, const void *id128, int rank, int world);
int rdh_comm_init_all(rdh_ctx **ctxs, int n);
int rdh_comm_destroy(rdh_ctx *ctx);
int rdh_allgather_tiles(rdh_ctx *ctx, const float *d_packed, float *d_frame);
int rdh_path_trace_gathered(rdh_ctx *ctx, float *d_directFrame, float *d_indirectFrame, int iter, int looper, int maxDepth,
                            uint32_t flags);
int rdh_path_trace_gathered_all(rdh_ctx **ctxs, int n, float *const *d_directFrames, float *const *d_indirectFrames, int iter,
                                int looper, int maxDepth, uint32_t flags);
int rdh_restir_exchange(rdh_ctx *ctx);
int rdh_restir_direct_gathered(rdh_ctx *ctx, float *d_directFrame, int iter, int looper, const rdh_gbuffer *gb,
                               const rdh_restir_params *params, uint32_t flags);
int rdh_gbuffer_exchange(rdh_ctx *ctx, const rdh_gbuffer *gb);
int rdh_gbuffer_exchange_all(rdh_ctx **ctxs, int n, const rdh_gbuffer *gbs);
int rdh_restir_direct_gathered_all(rdh_ctx **ctxs, int n, float *const *d_directFrames, int iter, int looper, const rdh_gbuffer *gbs,
                                   const rdh_restir_params *params, uint32_t flags);
int rdh_comm_set_overlap(rdh_ctx *ctx, int enable);
int rdh_comm_join(rdh_ctx *ctx);

/* Display path (the step after the hot path): replaces copyImageToPBO's four overloads → sendImageToPBO
 * (/root/reference/src/pathtrace.cu:32-147; declared src/pathtrace.h:25-29).  d_pbo: uchar4[width*height] (alpha 0).
 * kind 0: d_image is vec3[w*h], colour = image * scale, tone mapping 0 None / 1 Filmic / 2 ACES (src/common.h:23-25),
 *         then gamma 1/2.2 (src/mathUtil.h:110-126);
 * kind 1: vec2[w*h] shown as (r, g, 0); kind 2: float[w*h] shown as grey; kind 3: int[w*h] pixel indices shown as
 *         normalised (x, y) (the motion-vector view).  toneMapping and scale are read for kind 0 only.
 * Asynchronous on the context's stream, like the reference's launch. */
int rdh_copy_image_to_pbo(rdh_ctx *ctx, void *d_pbo, const void *d_image, int width, int height, int kind, int toneMapping,
                          float scale);

/* Denoisers (/root/reference/src/denoiser.cu, declared src/denoiser.h).  One entry per reference kernel;
 * the filter classes (LeveledEAWFilter, SpatioTemporalFilter: buffers + the level schedule, denoiser.cu:411-558) are
 * composed from them by the C++ shim / the Python mirror.  All images are vec3[w*h] device buffers, variances float[w*h];
 * gb is the G-buffer rdh_gbuffer_render filled (current planes = gb->frameIdx, last = the other).  Asynchronous on the
 * context's stream.
 *   rdh_denoise_eaw                   waveletFilter, EAW (:17-84)  — EAWaveletFilter::filter (:388-397)
 *   rdh_denoise_svgf                  waveletFilter with variance (:92-173) — EAWaveletFilter::filter (:399-409)
 *   rdh_denoise_modulate              modulate (:175-185) — modulateAlbedo (:363-371)
 *   rdh_denoise_add                   add (:187-206) — addImage (:373-386); d_out may be d_in1
 *   rdh_denoise_temporal_accumulate   temporalAccumulate (:208-262)
 *   rdh_denoise_estimate_variance     estimateVariance (:264-299)
 *   rdh_denoise_filter_variance       filterVariance (:301-328)                                                      */
int rdh_denoise_eaw(rdh_ctx *ctx, float *d_colorOut, const float *d_colorIn, const rdh_gbuffer *gb, const void *camera196,
                    float sigLumin, float sigNormal, float sigDepth, int level);
int rdh_denoise_svgf(rdh_ctx *ctx, float *d_colorOut, const float *d_colorIn, float *d_varianceOut, const float *d_varianceIn,
                     const float *d_filteredVar, const rdh_gbuffer *gb, const void *camera196, float sigLumin, float sigNormal,
                     float sigDepth, int level);
int rdh_denoise_modulate(rdh_ctx *ctx, float *d_image, const rdh_gbuffer *gb);
int rdh_denoise_add(rdh_ctx *ctx, float *d_out, const float *d_in1, const float *d_in2, int width, int height);
int rdh_denoise_temporal_accumulate(rdh_ctx *ctx, float *d_colorAccumOut, const float *d_colorAccumIn, float *d_momentAccumOut,
                                    const float *d_momentAccumIn, const float *d_colorIn, const rdh_gbuffer *gb, int first);
int rdh_denoise_estimate_variance(rdh_ctx *ctx, float *d_variance, const float *d_moment, int width, int height);
int rdh_denoise_filter_variance(rdh_ctx *ctx, float *d_varianceOut, const float *d_varianceIn, int width, int height);

/* Test access to the reservoir buffers (36-byte DirectReservoir[w*h]): which = 0 current out, 1 last, 2 temp. */
int rdh_restir_read(rdh_ctx *ctx, int which, void *hostOut);
/* Test / analysis access to what the last split pass 1 (rdh_restir_direct without RDH_PT_RESTIR_FUSED) left between its launches,
 * per slot of its launch domain: which = 0 primary rays (6 floats), 1 shadow segments (6 floats, NaN first = none),
 * 2 the set-aside lists {int count[4]; int primarySlots[256]; int shadowSlots[256]}.  Returns the byte size (hostOut NULL: only
 * that), or a negative RDH_ERR_*. */
long long rdh_restir_read_scratch(rdh_ctx *ctx, int which, void *hostOut, long long maxBytes);

/* ---- traversal entry points (tests / roofline bench) ------------------------------------------------ */
/* d_rays: {origin.xyz, direction.xyz}[n].  DevScene::intersect (src/scene.h:262-301) per ray.
   flags: RDH_PT_COUNT; RDH_PT_PERSISTENT selects the walk-only lane-refill kernel (device/kernels_walk.h: a lane whose ray
   has ended takes the next ray of the batch; 1.6x the one-lane-per-ray kernel's rate, same records and counters) — for both
   entries. */
int rdh_trace_closest(rdh_ctx *ctx, const float *d_rays, int64_t n, rdh_hit *d_hits, uint32_t flags);
/* d_segments: {x.xyz, y.xyz}[n] -> d_occluded[n] in {0,1}.  DevScene::testOcclusion (src/scene.h:303-334). */
int rdh_trace_occluded(rdh_ctx *ctx, const float *d_segments, int64_t n, int32_t *d_occluded, uint32_t flags);

/* The ray lists of one frame, for the traversal-only measurements (bench.py roofline.traversal_only, cpu_baseline): runs the
 * frame `looper` of pathTrace (iter 0, one-lane-per-pixel kernel, scratch images) and appends every DevScene::intersect ray
 * {origin, direction} to d_closest (float[capClosest][6]) and every DevScene::testOcclusion segment {x, y} to d_any
 * (float[capAny][6]) — the input layouts of rdh_trace_closest / rdh_trace_occluded.  *nClosest / *nAny are the numbers of rays
 * the frame traces (rays beyond a capacity are counted, not stored; pass capacities 0 to size the lists).  Blocking. */
int rdh_dump_rays(rdh_ctx *ctx, int looper, int maxDepth, float *d_closest, int64_t capClosest, float *d_any, int64_t capAny,
                  int64_t *nClosest, int64_t *nAny);
/* Contexts that render CONCURRENTLY on one GPU (several frames in flight, each on its own stream): the persistent kernels size
 * their grid to fill the chip; `share` divides it so that `share` contexts together fill it once.  Default 1. */
int rdh_set_occupancy_share(rdh_ctx *ctx, int share);

int rdh_counters_reset(rdh_ctx *ctx);
int rdh_counters_read(rdh_ctx *ctx, rdh_counters *out); /* blocking */

/* Sum of the hipEvent-measured durations of the traversal-kernel launches made with RDH_PT_PROFILE since the last
 * rdh_profile_reset, and how many launches that was (at most 8192 are recorded).  Both calls block. */
int rdh_profile_reset(rdh_ctx *ctx);
int rdh_profile_read(rdh_ctx *ctx, double *totalMs, int64_t *launches);

/* Diagnostic builds only (-DRD_PERSIST_STAMPS): wall-clock stamps (100 MHz ticks) of the last persistent launch —
 * [0][w] wave w started, [1][w] it found the pixel supply dry, [2][w] it ended, [3][w] ticks it spent tracing literal-class rays
 * whole, [4][w] how many of those it had.  RDH_ERR_UNSUPPORTED otherwise. */
int rdh_debug_persist_stamps(rdh_ctx *ctx, uint64_t *out5x4096);
/* Diagnostic builds only (-DRD_PERSIST_PHASES): of the last persistent launch, summed over its waves — [0..5] s_memtime
 * ticks (100 MHz) spent in raygen / whole-wave rays / box loop / leaf tests / retire / shading, [6] total; [8] box
 * wave-steps, [9] box lane-steps, [10] leaf calls, [11] leaf lanes, [12] shade calls, [13] shade lanes, [14] raygen calls,
 * [15] raygen lanes.  RDH_ERR_UNSUPPORTED otherwise. */
int rdh_debug_persist_phases(rdh_ctx *ctx, uint64_t *out16);

/* Time of the most recent render call's kernels, measured with hipEvents on the context's stream (blocking);
 * the reference prints this figure from pathTrace (src/pathtrace.cu:364-374). */
int rdh_last_kernel_ms(rdh_ctx *ctx, float *ms);

#ifdef __cplusplus
}
#endif
#endif
