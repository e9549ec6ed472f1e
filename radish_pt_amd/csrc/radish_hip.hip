// radish_pt_amd/csrc/radish_hip.hip — C ABI (include/radish_hip.h) over the gfx950 kernels.
//
// Host side of the hot path: scene re-layout + upload (replaces DevScene::create, /root/reference/src/scene.cpp:461-551),
// launch wrappers (replace pathTrace / pathTraceDirect / GBuffer::render / ReSTIRInit|Free|Direct,
// /root/reference/src/pathtrace.cu:351-407, gBuffer.cu:83-103, restir.cu:205-251).  There is no CPU fallback: without a
// HIP device rdh_create fails with RDH_ERR_NO_DEVICE.
#include "radish_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "device/kernels_pt.h"
#include "device/kernels_restir.h"
#include "device/kernels_wave.h"
#include "device/kernels_persist.h"
#include "device/kernels_walk.h"
#include "device/kernels_display.h"
#include "device/kernels_denoise.h"

using namespace rd;

constexpr int kWfParts = 3;  // sub-frame pipelines of the wavefront path (RDH_PT_WF_SUBFRAMES)

struct rdh_ctx {
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    hipStream_t sideStream = nullptr;  // the wavefront path's second pipeline, k_persist_schedule
    hipStream_t litStream = nullptr;   // the workgroup-per-ray launches for literal-class rays (k_gbuffer_literal, k_trace_wg_list), beside the
                                       // kernel that walks everything else: the side stream.  RADISH_LIT_PRIORITY=1 (experiments): a high-
                                       // priority stream of its own — those launches then end earlier, frames do not, and the extra
                                       // hardware queue cost the three-stream wavefront path 4 % (profiles/r03_za_*, r03_zg_*)
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    bool timed = false;
    std::string err;

    // scene
    bool haveScene = false;
    DScene ds{};
    std::vector<void *> sceneAllocs;
    Counters *dCounters = nullptr;
    PersistCounters *dPersist = nullptr;
    unsigned persistGrid = 0;
    unsigned persistGridPair = 0;  // resident waves of k_pt_persistent<false, true>
    unsigned gbufGrid = 0;  // resident waves of k_gbuffer_persistent
    unsigned gbufGridPair = 0;  // ... of its sibling-pair variant
    unsigned walkGrid[2] = {0, 0};  // ... of k_walk_persistent<false, false / true>
    unsigned pairGrid[2] = {0, 0};  // ... of k_walk_pair<false, false / true>
    int *treeOvf = nullptr;         // the pair walkers' stacks beyond their LDS rings (traverse.h, pairPush), ray-batch and persistent kernels
    size_t treeOvfInts = 0;
    unsigned wfGrid[4] = {0, 0, 0, 0};  // resident workgroups of k_wf_trace<false> / <true>, (unused), k_wf_shade
    unsigned wfGridPair[2] = {0, 0};    // ... of k_wf_trace<false, true> / <true, true>
    int *wfTreeOvf[3] = {nullptr, nullptr, nullptr};  // per sub-frame workspace: the deep end of the pair walkers' stacks
    size_t wfTreeOvfInts[3] = {0, 0, 0};
    int pairMode = -1;  // -1: by scene size (usePairs); 0 / 1: RADISH_PAIRS in the environment (experiments)
    float *posPlane = nullptr;  // denoisers: Camera::getPosition of every pixel (k_position_plane)
    long long posPlanePixels = 0;
    // Longest-paths-first block order (k_persist_schedule), off the critical path: launch n writes blockCost[n & 1]; the
    // schedule of those costs runs on sideStream while launch n + 1 renders, and launch n + 2 reads its order from
    // blockOrder[n & 1] (and reuses the cost buffer the schedule has zeroed).
    unsigned *blockCost[2] = {nullptr, nullptr};
    unsigned *blockEma = nullptr;   // running mean of the costs over the launches so far
    int *blockOrder[2] = {nullptr, nullptr};
    int costBlocks = 0;             // blocks the arrays are sized for
    bool orderValid = false;        // false: restart the sequence (scene, camera size or partition changed)
    int persistFrame = 0;           // persistent launches since the last restart
    hipEvent_t evFrame = nullptr, evSched[2] = {nullptr, nullptr};

    // camera
    bool haveCamera = false;
    DCamera cam{};

    // partition
    int rank = 0, world = 1, tile = 64;
    int share = 1;  // contexts rendering concurrently on this GPU: persistent grids are divided by it (rdh_set_occupancy_share)
    bool forcePacked = false;  // render into packed tile buffers even with one rank (the *_gathered entries)

    // ReSTIR buffers (restir.cu:4-7)
    float *resvCur = nullptr, *resvLast = nullptr, *resvTemp = nullptr;
    float4 *restirState = nullptr;
    long long restirPixels = 0;
    // per-slot scratch of the split pass 1 (kernels_restir.h RestirSplit), grown on demand
    void *splitBuf = nullptr;
    long long splitSlots = 0;      // allocation high-water mark
    long long splitLastSlots = 0;  // slots (= plane stride) of the last split pass 1: what rdh_restir_read_scratch reads back
    unsigned risGrid = 0;
    LightPre *lightPre = nullptr;
    bool restirFirstFrame = true;

    // per-launch timing of the dominant (traversal) kernel: ring of hipEvent pairs (RDH_PT_PROFILE)
    std::vector<hipEvent_t> profEvents;
    size_t profUsed = 0;

    // RCCL communicator of this rank (rdh_comm_init) and the staging buffers of the collectives: this rank's packed tiles
    // (send side) and the gathered tiles of every rank (receive side), grown on demand
    void *comm = nullptr;
    // ReSTIR's exchanges run on a communication stream of their own, so that the G-buffer gather overlaps pass 1's ray
    // generation / walks / RIS and the reservoir gather overlaps the NEXT frame's G-buffer pass and pass 1.  Every
    // collective of those entries is issued on commStream, in one order; the render stream and it hand over through events.
    hipStream_t commStream = nullptr;
    hipEvent_t evToComm = nullptr, evGbuf = nullptr, evImg = nullptr, evResv = nullptr;
    bool commOverlap = true;   // rdh_comm_set_overlap
    bool gbufPending = false;  // a G-buffer exchange is in flight on commStream: wait for evGbuf before the planes are read
    bool resvPending = false;  // ... a reservoir exchange: wait for evResv before `last` reservoirs are read
    float *commSend[2] = {nullptr, nullptr};
    float *commRecv = nullptr;
    size_t commSendFloats[2] = {0, 0}, commRecvFloats = 0;

    // wavefront workspace
    WaveWorkspace wf[kWfParts] = {};  // one per sub-frame (RDH_PT_WF_SUBFRAMES: three pipelines on three streams)
    hipStream_t wfStream = nullptr;    // the third pipeline's stream (the first two use stream and sideStream)
    hipEvent_t evWfJoin3 = nullptr;
    long long wfCapacity = 0;   // path slots each workspace holds
    std::vector<void *> wfAllocs;
    hipEvent_t evWfFork = nullptr, evWfJoin = nullptr;
};

namespace {

int waitExchanges(rdh_ctx *c, bool gbuf, bool resv);  // defined with the collectives below
int joinComm(rdh_ctx *c);

int fail(rdh_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((c), (int)e_, "HIP error (%s:%d): %s: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int uploadVec(rdh_ctx *c, const std::vector<T> &v, const T **out) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIP_TRY(c, hipMalloc(&p, bytes));
    c->sceneAllocs.push_back(p);
    if (!v.empty()) HIP_TRY(c, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T *>(p);
    return RDH_OK;
}

float asFloat(int32_t i) {
    float f;
    memcpy(&f, &i, 4);
    return f;
}

// DScene::pairs (layouts.h) from the reference-layout arrays: the tree once, shared by the six orderings.  Returns false (no pairs;
// the kernels then walk the six threaded arrays) unless bvhNodes[0..5] are six pre-orders of one binary tree over boundingBoxes —
// which is what BVHBuilder::buildMTBVH emits (src/bvh.cpp:136-183) — that differ only in which child comes first.  The caller
// has range-checked every node already.  `tree` = one record per node (box, w = primitiveId or ~first child, ordering bits; record
// 0 is the root, a node's children are neighbours), from which the pairs are cut; `depth` = the most far children a walk can have
// pending at once.
bool buildSharedTree(const rdh_scene_desc *d, int S, std::vector<NodeRec> &tree, std::vector<PairRec> &pairs, int &depth) {
    if (S <= 0) return false;
    std::vector<int> canon((size_t)S, -1);
    tree.assign((size_t)S + 1, NodeRec{});
    tree[S].lo_prim = make_float4(0.f, 0.f, 0.f, asFloat(0));  // pad record, never visited
    tree[S].hi_next = make_float4(0.f, 0.f, 0.f, asFloat(0));
    std::vector<int> w3((size_t)S, 0), bits((size_t)S, 0);
    int nextFree = 1;
    std::vector<unsigned char> seen((size_t)S, 0);
    {
        const int32_t *src = d->bvhNodes[0];
        canon[src[1]] = 0;
        for (int p = 0; p < S; p++) {
            const int prim = src[3 * p], box = src[3 * p + 1], next = src[3 * p + 2];
            const int b = canon[box];
            if (b < 0 || seen[b]) return false;  // reached before its parent, or twice: not a pre-order
            seen[b] = 1;
            if (prim >= 0) {
                if (next != p + 1) return false;
                w3[b] = prim;
            } else {
                const int first = p + 1;
                if (first >= S) return false;
                const int second = src[3 * first + 2];
                if (second <= first || second >= S || src[3 * second + 2] != next) return false;
                const int bf = src[3 * first + 1], bs = src[3 * second + 1];
                if (bf == bs || canon[bf] >= 0 || canon[bs] >= 0 || nextFree + 2 > S) return false;
                canon[bf] = nextFree;
                canon[bs] = nextFree + 1;
                w3[b] = ~nextFree;
                nextFree += 2;
            }
        }
        if (nextFree != S) return false;
    }
    for (int k = 1; k < 6; k++) {
        const int32_t *src = d->bvhNodes[k];
        std::fill(seen.begin(), seen.end(), 0);
        for (int p = 0; p < S; p++) {
            const int prim = src[3 * p], box = src[3 * p + 1], next = src[3 * p + 2];
            const int b = canon[box];
            if (b < 0 || seen[b]) return false;
            seen[b] = 1;
            if (p == 0 && b != 0) return false;
            if (prim >= 0) {
                if (next != p + 1 || w3[b] != prim) return false;
            } else {
                if (w3[b] >= 0) return false;
                const int first = p + 1;
                if (first >= S) return false;
                const int second = src[3 * first + 2];
                if (second <= first || second >= S || src[3 * second + 2] != next) return false;
                const int c0 = ~w3[b], cf = canon[src[3 * first + 1]], cs = canon[src[3 * second + 1]];
                if (cf == c0 && cs == c0 + 1) {
                } else if (cf == c0 + 1 && cs == c0) {
                    bits[b] |= 1 << k;
                } else {
                    return false;
                }
            }
        }
    }
    // pending far children on the way down, per ordering (children have larger record numbers than their parent)
    depth = 0;
    std::vector<int> pend((size_t)S);
    for (int k = 0; k < 6; k++) {
        pend[0] = 0;
        for (int b = 0; b < S; b++) {
            if (w3[b] >= 0) continue;
            const int c0 = ~w3[b], nb = (bits[b] >> k) & 1;
            pend[c0 + nb] = pend[b] + 1;
            pend[c0 + (nb ^ 1)] = pend[b];
            if (pend[b] + 1 > depth) depth = pend[b] + 1;
        }
    }
    for (int box = 0; box < S; box++) {
        const int b = canon[box];
        const float *q = d->boundingBoxes + 6 * (size_t)box;
        tree[b].lo_prim = make_float4(q[0], q[1], q[2], asFloat(w3[b]));
        tree[b].hi_next = make_float4(q[3], q[4], q[5], asFloat(bits[b]));
    }
    // sibling pairs (layouts.h, PairRec): children c0 = 2q + 1 and c0 + 1 of the inner node that was the (q + 1)-th to be reached
    pairs.assign((size_t)(S - 1) / 2 + 1, PairRec{});  // + a pad record
    auto childW = [&](int b) { return w3[b] >= 0 ? w3[b] : ~((~w3[b] - 1) / 2); };
    for (int b = 0; b < S; b++) {
        if (w3[b] >= 0) continue;
        const int c0 = ~w3[b], q = (c0 - 1) / 2;
        const float4 l0 = tree[c0].lo_prim, h0 = tree[c0].hi_next, l1 = tree[c0 + 1].lo_prim, h1 = tree[c0 + 1].hi_next;
        pairs[q].lo0_w0 = make_float4(l0.x, l0.y, l0.z, asFloat(childW(c0)));
        pairs[q].hi0_bits = make_float4(h0.x, h0.y, h0.z, asFloat(bits[b]));
        pairs[q].lo1_w1 = make_float4(l1.x, l1.y, l1.z, asFloat(childW(c0 + 1)));
        pairs[q].hi1_pad = make_float4(h1.x, h1.y, h1.z, asFloat(0));
    }
    return true;
}

struct HostCamera {  // src/sceneStructs.h:118-130
    int32_t resx, resy;
    float position[3], rotation[3], view[3], up[3], right[3];
    float fov[2], pixelLength[2];
    float rotationMatInv[9];
    float viewProjection[16];
    float lensRadius, focalDist, tanFovY;
};
static_assert(sizeof(HostCamera) == 196, "Camera layout");

DCamera toDeviceCamera(const void *camera196) {
    HostCamera h;
    memcpy(&h, camera196, sizeof(h));
    DCamera d;
    d.resx = h.resx;
    d.resy = h.resy;
    d.position = {h.position[0], h.position[1], h.position[2]};
    d.view = {h.view[0], h.view[1], h.view[2]};
    d.up = {h.up[0], h.up[1], h.up[2]};
    d.right = {h.right[0], h.right[1], h.right[2]};
    d.rotationMatInv.c0 = {h.rotationMatInv[0], h.rotationMatInv[1], h.rotationMatInv[2]};
    d.rotationMatInv.c1 = {h.rotationMatInv[3], h.rotationMatInv[4], h.rotationMatInv[5]};
    d.rotationMatInv.c2 = {h.rotationMatInv[6], h.rotationMatInv[7], h.rotationMatInv[8]};
    d.lensRadius = h.lensRadius;
    d.focalDist = h.focalDist;
    // glm::tan(glm::radians(fov.y)) (sceneStructs.h:75), hoisted to the host: libm tanf of (deg * pi/180)
    d.tanFovY = tanf(h.fov[1] * 0.01745329251994329576923690768489f);
    return d;
}

PixelMap makePixelMap(const rdh_ctx *c) {
    PixelMap pm;
    pm.W = c->cam.resx;
    pm.H = c->cam.resy;
    pm.tile = c->tile;
    pm.tilesX = (pm.W + pm.tile - 1) / pm.tile;
    int tilesY = (pm.H + pm.tile - 1) / pm.tile;
    pm.numTiles = pm.tilesX * tilesY;
    pm.rank = c->rank;
    pm.world = c->world;
    pm.tilesPerRank = (pm.numTiles + pm.world - 1) / pm.world;
    pm.packed = (c->world > 1 || c->forcePacked) ? 1 : 0;
    int bpe = pm.tile / 8;
    pm.numBlocks = pm.tilesPerRank * bpe * bpe;
    return pm;
}

// Do the per-lane walks of this launch go over the sibling pairs (DScene::pairs) or over the six threaded arrays?  Pairs wherever
// the uploaded arrays allow: measured on each scene's own frame rays (profiles/r03_m_*) the walker needs 0.99x the time on Cornell
// (37 k nodes), 0.79x on teapots (201 k), 0.61x on the 1-M-triangle scene, and whole frames gain on all three in every structure
// (a sixth of the node footprint, half the dependent round trips).  RDH_PT_NO_PAIRS / RADISH_PAIRS=0 keep the threaded walks.
constexpr int kPairMinNodes = 0;
bool usePairs(const rdh_ctx *c, uint32_t flags) {
    if (!c->ds.pairs || (flags & RDH_PT_NO_PAIRS)) return false;
    if (flags & RDH_PT_PAIRS) return true;
    if (c->pairMode >= 0) return c->pairMode != 0;
    return c->ds.bvhSize >= kPairMinNodes;
}

// Primary rays as wave packets (traverse.h, packetWalk)?  RADISH_PACKETS=0 in the environment switches them off (experiments).
bool usePackets(uint32_t flags) {
    static const int env = getenv("RADISH_PACKETS") ? atoi(getenv("RADISH_PACKETS")) : 1;
    return env != 0 && !(flags & RDH_PT_NO_PACKETS);
}

// Grid for "4 waves per workgroup, one 8x8 block per wave", padded to a multiple of 8 workgroups (xcdSwizzle).
unsigned gridFor(const PixelMap &pm) {
    unsigned work = (unsigned)(pm.numBlocks + 3) / 4;
    return ((work + 7u) / 8u) * 8u;
}

unsigned gridBlocks(const PixelMap &pm) { return (((unsigned)pm.numBlocks + 7u) / 8u) * 8u; }  // single-wave workgroups

int requireReady(rdh_ctx *c) {
    if (!c) return RDH_ERR_ARGS;
    if (!c->haveScene) return fail(c, RDH_ERR_NO_SCENE, "no scene uploaded (rdh_scene_upload)");
    if (!c->haveCamera) return fail(c, RDH_ERR_NO_SCENE, "no camera set (rdh_set_camera)");
    return RDH_OK;
}

constexpr size_t kProfilePairs = 8192;

// Returns the index of a fresh event pair (2*i, 2*i+1) or -1 when profiling is off / the ring is full.
long profBegin(rdh_ctx *c, uint32_t flags) {
    if (!(flags & RDH_PT_PROFILE) || c->profUsed >= kProfilePairs) return -1;
    if (c->profEvents.size() < 2 * (c->profUsed + 1)) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        c->profEvents.push_back(a);
        c->profEvents.push_back(b);
    }
    long i = (long)c->profUsed++;
    hipEventRecord(c->profEvents[2 * i], c->stream);
    return i;
}
void profEnd(rdh_ctx *c, long i) {
    if (i >= 0) hipEventRecord(c->profEvents[2 * i + 1], c->stream);
}

// Persistent kernels: enough workgroups to fill every CU at full occupancy; surplus ones find the queues empty.
constexpr unsigned kPersistentGrid = 256u * 8u;

template <typename T>
int wfAlloc(rdh_ctx *c, T **out, size_t count) {
    void *p = nullptr;
    HIP_TRY(c, hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16)));
    c->wfAllocs.push_back(p);
    *out = static_cast<T *>(p);
    return RDH_OK;
}

// Workspace for the wavefront pipeline: 152 B of path state + the ray / shadow queue entries per path slot, per sub-frame, plus
// the two literal-class lists (litq) of kWfLitCap entries.
int wavefrontEnsure(rdh_ctx *c, const PixelMap &pm) {
    long long slots = (long long)pm.numBlocks * 64;  // workspace 0: the whole frame; 1 and 2: a third each (+ a block), sized for half
    if (c->wfCapacity >= slots) return RDH_OK;
    for (void *p : c->wfAllocs) hipFree(p);
    c->wfAllocs.clear();
    c->wfCapacity = 0;
    const size_t n0 = (size_t)slots, n1 = (size_t)((pm.numBlocks + 1) / 2) * 64;
    for (int h = 0; h < kWfParts; h++) {
        WaveWorkspace &w = c->wf[h];
        const size_t n = h == 0 ? n0 : n1;  // workspace 0 also serves the one-pipeline mode (whole frame)
        int rc;
        if ((rc = wfAlloc(c, &w.ro, n)) || (rc = wfAlloc(c, &w.rd, n)) || (rc = wfAlloc(c, &w.thr, n)) ||
            (rc = wfAlloc(c, &w.prevPos, n)) || (rc = wfAlloc(c, &w.accD, n)) || (rc = wfAlloc(c, &w.accI, n)) ||
            (rc = wfAlloc(c, &w.nee, n)) || (rc = wfAlloc(c, &w.sht, n)) || (rc = wfAlloc(c, &w.rng, n)) ||
            (rc = wfAlloc(c, &w.hit, n)) || (rc = wfAlloc(c, &w.rayq[0], n)) || (rc = wfAlloc(c, &w.rayq[1], n)) ||
            (rc = wfAlloc(c, &w.shadowq, n)) || (rc = wfAlloc(c, &w.ctr, 1)) || (rc = wfAlloc(c, &w.litq[0], (size_t)kWfLitCap)) ||
            (rc = wfAlloc(c, &w.litq[1], (size_t)kWfLitCap)))
            return rc;
        w.litCap = 0;
    }
    if (!c->evWfFork) {
        HIP_TRY(c, hipEventCreateWithFlags(&c->evWfFork, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->evWfJoin, hipEventDisableTiming));
        HIP_TRY(c, hipStreamCreateWithFlags(&c->wfStream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->evWfJoin3, hipEventDisableTiming));
    }
    c->wfCapacity = slots;
    return RDH_OK;
}

int wavefrontPathTrace(rdh_ctx *c, const PixelMap &pm, float *d_direct, float *d_indirect, int iter, int looper,
                       int maxDepth, uint32_t flags) {
    if (maxDepth > kMaxWaveDepth) return fail(c, RDH_ERR_ARGS, "maxDepth %d exceeds %d", maxDepth, kMaxWaveDepth);
    const bool count = (flags & RDH_PT_COUNT) != 0, sort = (flags & RDH_PT_SORT_MATERIAL) != 0;
    // Persistent kernels whose waves take a STATIC first packet: every workgroup must be resident from the start, or the
    // surplus ones run their static packets after everybody else has finished (k_wf_trace: 72 VGPRs, 7 waves per SIMD as 256-thread
    // workgroups allow; k_wf_shade 3 — tests/test_kernel_resources.py pins the budgets).
    if (c->wfGrid[0] == 0) {
        int cus = 0, per[4] = {0, 0, 1, 0};
        HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per[0], k_wf_trace<false>, 256, 0));
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per[1], k_wf_trace<true>, 256, 0));
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per[3], k_wf_shade, 256, 0));
        for (int q = 0; q < 4; q++) {
            unsigned g = (unsigned)((per[q] < 1 ? 1 : per[q]) * cus);
            c->wfGrid[q] = g < kPersistentGrid ? g : kPersistentGrid;
        }
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per[0], (k_wf_trace<false, true>), 256, 0));
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per[1], (k_wf_trace<true, true>), 256, 0));
        for (int q = 0; q < 2; q++) {
            unsigned g = (unsigned)((per[q] < 1 ? 1 : per[q]) * cus);
            c->wfGridPair[q] = g < kPersistentGrid ? g : kPersistentGrid;
        }
    }
    const bool tree = usePairs(c, flags);
    // Three sub-frames (8x8 blocks dealt round robin) as three pipelines on three streams: every stage of one pipeline ends on its
    // longest ray while the stages of the others fill the chip.  Each pipeline launches a third of the resident grid.  Measured on
    // the teapots / Cornell frame: one pipeline 12.3 / 5.14 ms, two 10.5 / 4.70, three 10.2 / 4.58, four 10.4 / 5.0.  Small frames
    // stay one pipeline.
    static const int partsEnv = getenv("RADISH_WF_PARTS") ? atoi(getenv("RADISH_WF_PARTS")) : kWfParts;  // experiments: 2
    const int parts = ((flags & RDH_PT_WF_SUBFRAMES) && pm.numBlocks >= 2048) ? (partsEnv >= 1 && partsEnv <= kWfParts ? partsEnv : kWfParts) : 1;
    hipStream_t sts[kWfParts] = {c->stream, c->sideStream, c->wfStream};
    // persistent grids: what stays resident, shared between the sub-frame pipelines and between the contexts that render side by
    // side on this GPU (rdh_set_occupancy_share)
    const unsigned div = (unsigned)parts * (unsigned)c->share;
    const unsigned traceGrid = std::max(8u, (tree ? c->wfGridPair[count ? 1 : 0] : c->wfGrid[count ? 1 : 0]) / div);
    const unsigned shadeGrid = std::max(8u, c->wfGrid[3] / div);
    if (tree) {  // the deep end of the walkers' stacks (anything that can fail comes before the fork)
        const int ovfDepth = c->ds.treeDepth + 1;  // entry numbers, not rows beyond the ring (traverse.h, pairPush)
        const size_t need = (size_t)traceGrid * 4 * 64 * (size_t)ovfDepth * 2;  // int2 entries
        for (int h = 0; h < parts; h++) {
            if (need > c->wfTreeOvfInts[h]) {
                HIP_TRY(c, hipDeviceSynchronize());
                if (c->wfTreeOvf[h]) hipFree(c->wfTreeOvf[h]);
                c->wfTreeOvf[h] = nullptr;
                c->wfTreeOvfInts[h] = 0;
                HIP_TRY(c, hipMalloc((void **)&c->wfTreeOvf[h], need * sizeof(int)));
                c->wfTreeOvfInts[h] = need;
            }
            c->wf[h].treeOvf = c->wfTreeOvf[h];
            c->wf[h].treeOvfDepth = ovfDepth;
        }
    }
    // RDH_PT_PROFILE with sub-frames: the pipelines' launches overlap by design, so what is timed is the FRAME — one event pair
    // from before the fork to after the join, on the context's stream (one pipeline: a pair around every k_wf_trace launch)
    const long pf = parts >= 2 ? profBegin(c, flags) : -1;
    if (parts >= 2) {
        HIP_TRY(c, hipEventRecord(c->evWfFork, c->stream));
        for (int h = 1; h < parts; h++) HIP_TRY(c, hipStreamWaitEvent(sts[h], c->evWfFork, 0));
    }
    // From here on the side streams may hold work: no early return before the join below (an error is carried in `rc`).
    int rc = RDH_OK;
    auto keep = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == RDH_OK) rc = fail(c, (int)e, "HIP error (%s:%d): %s: %s", __FILE__, __LINE__, what, hipGetErrorString(e));
    };
    for (int h = 0; h < parts; h++) {  // what a stage's literal-class list may hold: one entry per lane of the trace grid
        const unsigned waves = traceGrid * 4u;
        c->wf[h].litCap = (int)(waves * 64u < (unsigned)kWfLitCap ? waves * 64u : (unsigned)kWfLitCap);
        if (flags & RDH_PT_WF_SMALL_LISTS) c->wf[h].litCap = 4;  // tests: what does not fit stays in the ordinary queues
    }
    for (int h = 0; h < parts; h++) {
        hipStream_t st = sts[h];
        WaveWorkspace &w = c->wf[h];
        const unsigned nLocal = (unsigned)(pm.numBlocks - h + parts - 1) / (unsigned)parts;
        const unsigned gridBlk = (((nLocal + 3u) / 4u + 7u) / 8u) * 8u;
        keep(hipMemsetAsync(w.ctr, 0, sizeof(WaveCounters), st), "hipMemsetAsync(wavefront counters)");
        hipLaunchKernelGGL(k_wf_raygen, dim3(gridBlk), dim3(256), 0, st, c->ds, c->cam, pm, w, looper, h, parts);
    }
    for (int k = 0; k <= maxDepth && rc == RDH_OK; k++) {
        for (int h = 0; h < parts; h++) {  // interleaved issue: stage k of every pipeline before stage k + 1 of any
            hipStream_t st = sts[h];
            WaveWorkspace &w = c->wf[h];
            long pe = (h == 0 && parts == 1) ? profBegin(c, flags) : -1;
            if (tree && count) hipLaunchKernelGGL((k_wf_trace<true, true>), dim3(traceGrid), dim3(256), 0, st, c->ds, w, k);
            else if (tree) hipLaunchKernelGGL((k_wf_trace<false, true>), dim3(traceGrid), dim3(256), 0, st, c->ds, w, k);
            else if (count) hipLaunchKernelGGL(k_wf_trace<true>, dim3(traceGrid), dim3(256), 0, st, c->ds, w, k);
            else hipLaunchKernelGGL(k_wf_trace<false>, dim3(traceGrid), dim3(256), 0, st, c->ds, w, k);
            profEnd(c, pe);
            hipLaunchKernelGGL(k_wf_shade, dim3(shadeGrid), dim3(256), 0, st, c->ds, w, k, maxDepth, sort ? 1 : 0);
        }
        keep(hipGetLastError(), "wavefront stage launch");
    }
    for (int h = 0; h < parts && rc == RDH_OK; h++) {
        hipStream_t st = sts[h];
        const unsigned nLocal = (unsigned)(pm.numBlocks - h + parts - 1) / (unsigned)parts;
        const unsigned gridBlk = (((nLocal + 3u) / 4u + 7u) / 8u) * 8u;
        hipLaunchKernelGGL(k_wf_finish, dim3(gridBlk), dim3(256), 0, st, pm, c->wf[h], iter, d_direct, d_indirect, h, parts);
    }
    keep(hipGetLastError(), "k_wf_finish");
    if (parts >= 2) {  // the join is unconditional; if it cannot be expressed with events the side streams are drained
        hipError_t e1 = hipEventRecord(c->evWfJoin, c->sideStream);
        hipError_t e2 = hipStreamWaitEvent(c->stream, c->evWfJoin, 0);
        hipError_t e3 = hipEventRecord(c->evWfJoin3, c->wfStream);
        hipError_t e4 = hipStreamWaitEvent(c->stream, c->evWfJoin3, 0);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
            hipStreamSynchronize(c->sideStream);
            hipStreamSynchronize(c->wfStream);
            keep(e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : (e3 != hipSuccess ? e3 : e4)), "wavefront join");
        }
    }
    profEnd(c, pf);
    return rc;
}

// The walkers' stacks beyond their LDS rings (traverse.h, pairPush): one allocation of the context, grown on demand.
int ensureStackOverflow(rdh_ctx *c, size_t ints) {
    if (ints <= c->treeOvfInts) return RDH_OK;
    HIP_TRY(c, hipDeviceSynchronize());
    if (c->treeOvf) hipFree(c->treeOvf);
    c->treeOvf = nullptr;
    c->treeOvfInts = 0;
    HIP_TRY(c, hipMalloc((void **)&c->treeOvf, ints * sizeof(int)));
    c->treeOvfInts = ints;
    return RDH_OK;
}

// k_walk_persistent over a ray list (d_hits xor d_occ): as many single-wave workgroups as stay resident, lane refill.
// deferCount / deferList (ReSTIR's lists): literal-class rays have been listed by the producer of `d_rays`; they are traced one
// per workgroup on the side stream beside the walker, which skips them; the stream waits for both before it goes on.
int launchWalk(rdh_ctx *c, const float *d_rays, long long n, int4 *d_hits, int *d_occ, bool count, const int *deferCount = nullptr,
               const int *deferList = nullptr, int slotList = 0, bool pairs = false, bool packets = false, int packetBudget = kPacketBudget) {
    const int any = d_occ ? 1 : 0;
    packets = packets && !any && n > 0 && (n + 255) / 256 <= 0x7fffffffll;  // closest hits of a list the caller calls coherent
    // everything that can fail comes BEFORE the fork, so that no error path leaves the side stream un-joined
    if (pairs && !c->ds.pairs) return fail(c, RDH_ERR_UNSUPPORTED, "RDH_PT_PAIRS: the uploaded node arrays are not six orderings of one binary tree, or that tree is more than 2 048 levels deep");
    unsigned &resGrid = pairs ? c->pairGrid[any] : c->walkGrid[any];
    if (resGrid == 0) {
        int perCU = 0, cus = 0;
        if (pairs && any) HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_walk_pair<false, true>), 64, 0));
        else if (pairs) HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_walk_pair<false, false>), 64, 0));
        else if (any) HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_walk_persistent<false, true>), 64, 0));
        else HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_walk_persistent<false, false>), 64, 0));
        HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        resGrid = (unsigned)((perCU < 1 ? 1 : perCU) * cus);
    }
    const unsigned long long chunks = (unsigned long long)((n + 63) / 64);
    unsigned resident = resGrid / (unsigned)c->share;  // rdh_set_occupancy_share: contexts side by side on one GPU
    if (resident < 8u) resident = 8u;
    const unsigned grid = chunks < resident ? (unsigned)chunks : resident;
    const int ovfDepth = c->ds.treeDepth + 1;  // entry numbers (traverse.h, pairPush)
    if (pairs && !packets) {
        int rc = ensureStackOverflow(c, (size_t)resGrid * 64 * (size_t)ovfDepth * 2);
        if (rc) return rc;
    }
    int2 *const ovf = reinterpret_cast<int2 *>(c->treeOvf);
    if (!packets) HIP_TRY(c, hipMemsetAsync(c->dPersist, 0, offsetof(PersistCounters, deferred), c->stream));  // the walkers' chunk counter
    hipError_t ew = hipSuccess;
    if (deferCount) {
        HIP_TRY(c, hipEventRecord(c->evFork, c->stream));
        ew = hipStreamWaitEvent(c->litStream, c->evFork, 0);
        if (ew == hipSuccess) {
            if (any && count) hipLaunchKernelGGL((k_trace_wg_list<true, true>), dim3(kWalkDeferCap), dim3(kWgTraceThreads), 0, c->litStream, c->ds, d_rays, deferList, deferCount, d_hits, d_occ);
            else if (any) hipLaunchKernelGGL((k_trace_wg_list<false, true>), dim3(kWalkDeferCap), dim3(kWgTraceThreads), 0, c->litStream, c->ds, d_rays, deferList, deferCount, d_hits, d_occ);
            else if (count) hipLaunchKernelGGL((k_trace_wg_list<true, false>), dim3(kWalkDeferCap), dim3(kWgTraceThreads), 0, c->litStream, c->ds, d_rays, deferList, deferCount, d_hits, d_occ);
            else hipLaunchKernelGGL((k_trace_wg_list<false, false>), dim3(kWalkDeferCap), dim3(kWgTraceThreads), 0, c->litStream, c->ds, d_rays, deferList, deferCount, d_hits, d_occ);
        }
    }
#define RD_LAUNCH_WALK(CNT, ANYHIT, DEF)                                                                                                      \
    do {                                                                                                                                     \
        if (pairs)                                                                                                                           \
            hipLaunchKernelGGL((k_walk_pair<CNT, ANYHIT, DEF>), dim3(grid), dim3(64), 0, c->stream, c->ds, d_rays, n, d_hits, d_occ,          \
                               c->dPersist, ovf, ovfDepth, deferCount, slotList);                                                            \
        else                                                                                                                                 \
            hipLaunchKernelGGL((k_walk_persistent<CNT, ANYHIT, DEF>), dim3(grid), dim3(64), 0, c->stream, c->ds, d_rays, n, d_hits, d_occ,    \
                               c->dPersist, deferCount, slotList);                                                                           \
    } while (0)
    if (packets) {  // one 64-ray chunk per wave, walked as a packet (kernels_walk.h, k_walk_packet)
        const unsigned pgrid = (unsigned)((n + 255) / 256);
        if (deferCount && count) hipLaunchKernelGGL((k_walk_packet<true, true>), dim3(pgrid), dim3(256), 0, c->stream, c->ds, d_rays, n, d_hits, deferCount, slotList, packetBudget);
        else if (deferCount) hipLaunchKernelGGL((k_walk_packet<false, true>), dim3(pgrid), dim3(256), 0, c->stream, c->ds, d_rays, n, d_hits, deferCount, slotList, packetBudget);
        else if (count) hipLaunchKernelGGL((k_walk_packet<true, false>), dim3(pgrid), dim3(256), 0, c->stream, c->ds, d_rays, n, d_hits, deferCount, slotList, packetBudget);
        else hipLaunchKernelGGL((k_walk_packet<false, false>), dim3(pgrid), dim3(256), 0, c->stream, c->ds, d_rays, n, d_hits, deferCount, slotList, packetBudget);
        if (deferCount) {
            hipError_t e1 = hipEventRecord(c->evJoin, c->litStream);
            hipError_t e2 = hipStreamWaitEvent(c->stream, c->evJoin, 0);
            if (e1 != hipSuccess || e2 != hipSuccess) hipStreamSynchronize(c->litStream);
            HIP_TRY(c, ew);
        }
    } else if (deferCount) {
        if (any && count) RD_LAUNCH_WALK(true, true, true);
        else if (any) RD_LAUNCH_WALK(false, true, true);
        else if (count) RD_LAUNCH_WALK(true, false, true);
        else RD_LAUNCH_WALK(false, false, true);
        // the join is unconditional: whatever failed above, the stream goes on only after the side stream's work
        hipError_t e1 = hipEventRecord(c->evJoin, c->litStream);
        hipError_t e2 = hipStreamWaitEvent(c->stream, c->evJoin, 0);
        if (e1 != hipSuccess || e2 != hipSuccess) hipStreamSynchronize(c->litStream);
        HIP_TRY(c, ew);
    } else {
        if (any && count) RD_LAUNCH_WALK(true, true, false);
        else if (any) RD_LAUNCH_WALK(false, true, false);
        else if (count) RD_LAUNCH_WALK(true, false, false);
        else RD_LAUNCH_WALK(false, false, false);
    }
#undef RD_LAUNCH_WALK
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

void timeBegin(rdh_ctx *c) {
    hipEventRecord(c->evStart, c->stream);
}
int timeEnd(rdh_ctx *c, const char *what) {
    hipEventRecord(c->evStop, c->stream);
    c->timed = true;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, (int)e, "HIP error: %s: %s", what, hipGetErrorString(e));
    return RDH_OK;
}

}  // namespace

extern "C" {

int rdh_create(rdh_ctx **out, int device) {
    if (!out) return RDH_ERR_ARGS;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return RDH_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return RDH_ERR_NO_DEVICE;
    rdh_ctx *c = new rdh_ctx;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->evStart) != hipSuccess || hipEventCreate(&c->evStop) != hipSuccess ||
        hipStreamCreateWithFlags(&c->sideStream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evFrame, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evSched[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evSched[1], hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void **)&c->dCounters, sizeof(Counters)) != hipSuccess ||
        hipMalloc((void **)&c->dPersist, sizeof(PersistCounters)) != hipSuccess) {
        delete c;
        return RDH_ERR_NO_DEVICE;
    }
    c->stream = c->ownStream;
    c->litStream = c->sideStream;
    {
        const char *e = getenv("RADISH_LIT_PRIORITY");
        int least = 0, greatest = 0;
        if (e && atoi(e) != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least) {
            hipStream_t hs = nullptr;
            if (hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, greatest) == hipSuccess) c->litStream = hs;
            else (void)hipGetLastError();
        }
    }
    // cleared on the context's own (non-blocking) stream: a null-stream memset is not ordered against it
    if (hipMemsetAsync(c->dCounters, 0, sizeof(Counters), c->stream) != hipSuccess ||
        hipMemsetAsync(c->dPersist, 0, sizeof(PersistCounters), c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        delete c;
        return RDH_ERR_NO_DEVICE;
    }
    if (const char *e = getenv("RADISH_PAIRS")) c->pairMode = atoi(e) != 0 ? 1 : 0;  // experiments: force the sibling-pair walks on / off
    *out = c;
    return RDH_OK;
}

int rdh_scene_free(rdh_ctx *c) {
    if (!c) return RDH_ERR_ARGS;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (void *p : c->sceneAllocs) hipFree(p);
    c->sceneAllocs.clear();
    c->haveScene = false;
    return RDH_OK;
}

void rdh_destroy(rdh_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    rdh_scene_free(c);
    rdh_restir_free(c);
    rdh_comm_destroy(c);
    for (void *p : c->wfAllocs) hipFree(p);
    for (hipEvent_t e : c->profEvents) hipEventDestroy(e);
    if (c->dCounters) hipFree(c->dCounters);
    if (c->dPersist) hipFree(c->dPersist);
    if (c->posPlane) hipFree(c->posPlane);
    if (c->treeOvf) hipFree(c->treeOvf);
    for (int h = 0; h < 3; h++)
        if (c->wfTreeOvf[h]) hipFree(c->wfTreeOvf[h]);
    if (c->sideStream) hipStreamSynchronize(c->sideStream);
    if (c->litStream && c->litStream != c->sideStream) {
        hipStreamSynchronize(c->litStream);
        hipStreamDestroy(c->litStream);
    }
    for (int k = 0; k < 2; k++) {
        if (c->blockCost[k]) hipFree(c->blockCost[k]);
        if (c->blockOrder[k]) hipFree(c->blockOrder[k]);
        if (c->evSched[k]) hipEventDestroy(c->evSched[k]);
    }
    if (c->blockEma) hipFree(c->blockEma);
    if (c->evFrame) hipEventDestroy(c->evFrame);
    if (c->wfStream) {
        hipStreamSynchronize(c->wfStream);
        hipStreamDestroy(c->wfStream);
    }
    if (c->commStream) {
        hipStreamSynchronize(c->commStream);
        hipStreamDestroy(c->commStream);
        for (hipEvent_t e : {c->evToComm, c->evGbuf, c->evImg, c->evResv})
            if (e) hipEventDestroy(e);
    }
    if (c->evWfJoin3) hipEventDestroy(c->evWfJoin3);
    if (c->evWfFork) hipEventDestroy(c->evWfFork);
    if (c->evWfJoin) hipEventDestroy(c->evWfJoin);
    if (c->evStart) hipEventDestroy(c->evStart);
    if (c->evStop) hipEventDestroy(c->evStop);
    if (c->evFork) hipEventDestroy(c->evFork);
    if (c->evJoin) hipEventDestroy(c->evJoin);
    if (c->sideStream) hipStreamDestroy(c->sideStream);
    if (c->ownStream) hipStreamDestroy(c->ownStream);
    delete c;
}

const char *rdh_last_error(const rdh_ctx *c) { return c ? c->err.c_str() : "null context"; }

int rdh_set_stream(rdh_ctx *c, void *s) {
    if (!c) return RDH_ERR_ARGS;
    hipStream_t ns = static_cast<hipStream_t>(s);  // NULL is HIP's default (null) stream, which is what torch uses by default
    if (ns != c->stream) {
        // The persistent kernels' workspace (block reservation counter, cost / order double buffers, schedule events) and the
        // fork/join with the side stream assume that every launch of a context is ordered on ONE stream: drain the old one
        // and restart the block-order sequence before launches move to the new one.
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->sideStream) HIP_TRY(c, hipStreamSynchronize(c->sideStream));
        if (c->litStream != c->sideStream) HIP_TRY(c, hipStreamSynchronize(c->litStream));
        if (c->wfStream) HIP_TRY(c, hipStreamSynchronize(c->wfStream));
        if (c->commStream) HIP_TRY(c, hipStreamSynchronize(c->commStream));
        c->gbufPending = c->resvPending = false;
        c->orderValid = false;
        c->stream = ns;
    }
    return RDH_OK;
}

int rdh_synchronize(rdh_ctx *c) {
    if (!c) return RDH_ERR_ARGS;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->commStream) {  // exchanges that rdh_gbuffer_exchange / rdh_restir_direct_gathered left on the communication stream
        HIP_TRY(c, hipStreamSynchronize(c->commStream));
        c->gbufPending = c->resvPending = false;
    }
    return RDH_OK;
}

int rdh_scene_upload(rdh_ctx *c, const rdh_scene_desc *d) {
    if (!c || !d) return RDH_ERR_ARGS;
    if (!d->vertices || !d->normals || !d->texcoords || !d->boundingBoxes || !d->materialIds || !d->materials ||
        !d->sampleSequence || d->numPrims <= 0 || d->bvhSize != 2 * d->numPrims - 1 || d->numMaterials <= 0)
        return fail(c, RDH_ERR_ARGS, "rdh_scene_upload: inconsistent descriptor");
    for (int k = 0; k < 6; k++)
        if (!d->bvhNodes[k]) return fail(c, RDH_ERR_ARGS, "rdh_scene_upload: bvhNodes[%d] is null", k);
    if (d->lightSamplerLength > 0 && (!d->lightSampler || !d->lightPrimIds || !d->lightUnitRadiance))
        return fail(c, RDH_ERR_ARGS, "rdh_scene_upload: light arrays missing");
    const bool hasEnv = d->envMapTexId >= 0;
    if (d->lightSamplerLength != d->numLights + (hasEnv ? 1 : 0))
        return fail(c, RDH_ERR_ARGS, "lightSamplerLength %d != numLights %d (+1 with an env map)", d->lightSamplerLength, d->numLights);
    if (d->numTextures < 0 || (d->numTextures > 0 && !d->textures) || d->envMapTexId >= d->numTextures)
        return fail(c, RDH_ERR_ARGS, "rdh_scene_upload: texture table inconsistent");
    if (hasEnv) {
        const rdh_texture &e = d->textures[d->envMapTexId];
        if (!d->envMapSampler || d->envMapSamplerLength != e.width * e.height)
            return fail(c, RDH_ERR_ARGS, "envMapSamplerLength %d != env map %dx%d", d->envMapSamplerLength, e.width, e.height);
    } else if (d->envMapSamplerLength != 0) {
        return fail(c, RDH_ERR_ARGS, "envMapSampler given without envMapTexId");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    rdh_scene_free(c);

    struct HostMaterial { int32_t type; float color[3]; float metallic, roughness, ior; int32_t map[4]; };
    static_assert(sizeof(HostMaterial) == 44, "Material layout");
    const HostMaterial *hm = static_cast<const HostMaterial *>(d->materials);
    std::vector<MatRec> mats(d->numMaterials);
    for (int i = 0; i < d->numMaterials; i++) {
        for (int k = 0; k < 4; k++) {  // baseColor may be procedural (-2); the others are -1 or a texture index
            int id = hm[i].map[k];
            if (id >= d->numTextures || id < (k == 0 ? -2 : -1))
                return fail(c, RDH_ERR_ARGS, "material %d: map id %d out of range (numTextures %d)", i, id, d->numTextures);
        }
        mats[i].a = make_float4(asFloat(hm[i].type), hm[i].color[0], hm[i].color[1], hm[i].color[2]);
        mats[i].b = make_float4(hm[i].metallic, hm[i].roughness, hm[i].ior, 0.f);
        // Material field order: baseColorMapId, normalMapId, metallicMapId, roughnessMapId (src/material.h:282-285)
        mats[i].maps = make_int4(hm[i].map[0], hm[i].map[1], hm[i].map[2], hm[i].map[3]);
    }
    // textures: one blob + {width, height, first texel} per texture (src/scene.cpp:463-486)
    std::vector<float> texBlob;
    std::vector<int4> texInfo(std::max(d->numTextures, 1), make_int4(1, 1, 0, 0));
    for (int i = 0; i < d->numTextures; i++) {
        const rdh_texture &t = d->textures[i];
        if (t.width <= 0 || t.height <= 0 || !t.data) return fail(c, RDH_ERR_ARGS, "texture %d is empty", i);
        texInfo[i] = make_int4(t.width, t.height, (int)(texBlob.size() / 3), 0);
        texBlob.insert(texBlob.end(), t.data, t.data + 3 * (size_t)t.width * t.height);
    }
    if (texBlob.empty()) texBlob.assign(4, 0.f);

    const int N = d->numPrims, S = d->bvhSize;
    std::vector<TriRec> tris(N);
    std::vector<AttrRec> attrs(N);
    for (int p = 0; p < N; p++) {
        const float *v = d->vertices + 9 * (size_t)p;
        const float *n = d->normals + 9 * (size_t)p;
        const float *t = d->texcoords + 6 * (size_t)p;
        int mid = d->materialIds[p];
        if (mid < 0 || mid >= d->numMaterials) return fail(c, RDH_ERR_ARGS, "materialIds[%d] = %d out of range", p, mid);
        tris[p].a = make_float4(v[0], v[1], v[2], v[3]);
        tris[p].b = make_float4(v[4], v[5], v[6], v[7]);
        tris[p].c = make_float4(v[8], asFloat(mid), 0.f, 0.f);
        attrs[p].a = make_float4(n[0], n[1], n[2], n[3]);
        attrs[p].b = make_float4(n[4], n[5], n[6], n[7]);
        attrs[p].c = make_float4(n[8], t[0], t[1], t[2]);
        attrs[p].d = make_float4(t[3], t[4], t[5], 0.f);
    }
    int rc;
    if ((rc = uploadVec(c, tris, &c->ds.tris))) return rc;
    if ((rc = uploadVec(c, attrs, &c->ds.attrs))) return rc;
    if ((rc = uploadVec(c, mats, &c->ds.mats))) return rc;
    {  // shading class per triangle (Material::Type of its material, src/material.h:129): the key of the wavefront material sort
        std::vector<unsigned char> cls(N);
        for (int p = 0; p < N; p++) {
            const int type = hm[d->materialIds[p]].type;
            cls[p] = type == 0 ? 1 : (type == 1 ? 2 : (type == 2 ? 3 : 0));  // Lambertian, MetallicWorkflow, Dielectric; else terminal
        }
        if ((rc = uploadVec(c, cls, &c->ds.primClass))) return rc;
    }
    if ((rc = uploadVec(c, texBlob, &c->ds.texData))) return rc;
    if ((rc = uploadVec(c, texInfo, &c->ds.texInfo))) return rc;
    std::vector<AliasRec> envAlias(d->envMapSamplerLength);
    if (d->envMapSamplerLength) memcpy(envAlias.data(), d->envMapSampler, sizeof(AliasRec) * envAlias.size());
    for (auto &e : envAlias)
        if (e.failId < 0 || e.failId >= d->envMapSamplerLength) return fail(c, RDH_ERR_ARGS, "env-map alias table failId out of range");
    if ((rc = uploadVec(c, envAlias, &c->ds.envAlias))) return rc;
    c->ds.envTex = d->envMapTexId;
    c->ds.envSamplerLength = d->envMapSamplerLength;

    // The six orderings live in ONE allocation, (S + 1) records each (+1: a readable pad record at index S for speculative
    // next-node loads), so that a kernel can address any record as base + 32-bit byte offset (saddr form, one VALU) instead
    // of a per-lane 64-bit pointer; nodes[k] point into it.
    if ((size_t)6 * (S + 1) * sizeof(NodeRec) >= 0xffffffffull) return fail(c, RDH_ERR_ARGS, "scene too large: %d BVH nodes", S);
    std::vector<NodeRec> nodes((size_t)6 * (S + 1));
    for (int k = 0; k < 6; k++) {
        const int32_t *src = d->bvhNodes[k];
        NodeRec *dst = nodes.data() + (size_t)k * (S + 1);
        for (int i = 0; i < S; i++) {
            int prim = src[3 * i], box = src[3 * i + 1], next = src[3 * i + 2];
            if (box < 0 || box >= S || next < 0 || next > S || prim < -1 || prim >= N || next <= i)
                return fail(c, RDH_ERR_ARGS, "bvhNodes[%d][%d] = {%d,%d,%d} is not a valid threaded node", k, i, prim, box, next);
            const float *b = d->boundingBoxes + 6 * (size_t)box;
            for (int q = 0; q < 6; q++)  // aabbFast (device/traverse.h) assumes finite, moderately sized coordinates
                if (!(std::fabs(b[q]) < 1e30f)) return fail(c, RDH_ERR_ARGS, "boundingBoxes[%d] is not finite / exceeds 1e30", box);
            dst[i].lo_prim = make_float4(b[0], b[1], b[2], asFloat(prim));
            dst[i].hi_next = make_float4(b[3], b[4], b[5], asFloat(next));
        }
        dst[S].lo_prim = make_float4(0.f, 0.f, 0.f, asFloat(-1));
        dst[S].hi_next = make_float4(0.f, 0.f, 0.f, asFloat(S));
    }
    if ((rc = uploadVec(c, nodes, &c->ds.nodes[0]))) return rc;
    for (int k = 1; k < 6; k++) c->ds.nodes[k] = c->ds.nodes[0] + (size_t)k * (S + 1);
    {  // the same tree once, as sibling pairs (layouts.h, DScene::pairs) — when the six arrays ARE six pre-orders of one binary tree
        std::vector<NodeRec> tree;
        std::vector<PairRec> pairs;
        int depth = 0;
        c->ds.pairs = nullptr;
        c->ds.treeDepth = 0;
        // (a walker keeps a strip of treeDepth entries per lane in global memory for the deep end of its stack: a degenerate tree — a
        // chain thousands of levels deep — keeps the threaded walk, which needs no stack)
        constexpr int kMaxPairDepth = 2048;
        if (buildSharedTree(d, S, tree, pairs, depth) && depth <= kMaxPairDepth) {
            if ((rc = uploadVec(c, pairs, &c->ds.pairs))) return rc;
            c->ds.treeDepth = depth;
            c->ds.rootLo = make_float4(tree[0].lo_prim.x, tree[0].lo_prim.y, tree[0].lo_prim.z,
                                       asFloat(__builtin_bit_cast(int, tree[0].lo_prim.w) >= 0 ? __builtin_bit_cast(int, tree[0].lo_prim.w) : ~0));
            c->ds.rootHi = tree[0].hi_next;
        }
    }

    std::vector<LightRec> lights(d->numLights);
    for (int i = 0; i < d->numLights; i++) {
        int p = d->lightPrimIds[i];
        if (p < 0 || p >= N) return fail(c, RDH_ERR_ARGS, "lightPrimIds[%d] = %d out of range", i, p);
        const float *v = d->vertices + 9 * (size_t)p;
        const float *r = d->lightUnitRadiance + 3 * (size_t)i;
        lights[i].a = make_float4(v[0], v[1], v[2], v[3]);
        lights[i].b = make_float4(v[4], v[5], v[6], v[7]);
        lights[i].c = make_float4(v[8], r[0], r[1], r[2]);
    }
    if ((rc = uploadVec(c, lights, &c->ds.lights))) return rc;
    {  // per-light constants of the RIS loop, computed on the device with the functions the per-candidate code uses
        void *pre = nullptr;
        HIP_TRY(c, hipMalloc(&pre, std::max<size_t>(sizeof(LightPre) * lights.size(), 64)));
        c->sceneAllocs.push_back(pre);
        c->ds.lightPre = static_cast<const LightPre *>(pre);
        if (!lights.empty()) {
            hipLaunchKernelGGL(k_light_precompute, dim3((unsigned)((lights.size() + 255) / 256)), dim3(256), 0, c->stream, c->ds.lights,
                               static_cast<LightPre *>(pre), (int)lights.size(), d->sumLightPowerInv);
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    std::vector<AliasRec> alias(d->lightSamplerLength);
    if (d->lightSamplerLength) memcpy(alias.data(), d->lightSampler, sizeof(AliasRec) * alias.size());
    for (auto &e : alias)
        if (e.failId < 0 || e.failId >= d->lightSamplerLength) return fail(c, RDH_ERR_ARGS, "alias table failId out of range");
    if ((rc = uploadVec(c, alias, &c->ds.lightAlias))) return rc;
    std::vector<uint32_t> sobol(d->sampleSequence, d->sampleSequence + 10000 * 200);
    if ((rc = uploadVec(c, sobol, &c->ds.sobol))) return rc;

    c->ds.counters = c->dCounters;
    c->ds.bvhSize = S;
    c->ds.numPrims = N;
    c->ds.lightSamplerLength = d->lightSamplerLength;
    c->ds.sumLightPowerInv = d->sumLightPowerInv;
    c->haveScene = true;
    c->orderValid = false;
    return RDH_OK;
}

int rdh_set_camera(rdh_ctx *c, const void *camera196) {
    if (!c || !camera196) return RDH_ERR_ARGS;
    DCamera d = toDeviceCamera(camera196);
    if (d.resx <= 0 || d.resy <= 0) return fail(c, RDH_ERR_ARGS, "camera resolution %dx%d", d.resx, d.resy);
    if (d.resx != c->cam.resx || d.resy != c->cam.resy) c->orderValid = false;
    c->cam = d;
    c->haveCamera = true;
    return RDH_OK;
}

int rdh_set_partition(rdh_ctx *c, int rank, int world, int tileSize) {
    if (!c || world < 1 || rank < 0 || rank >= world || tileSize < 8 || (tileSize % 8) != 0)
        return fail(c, RDH_ERR_ARGS, "rdh_set_partition(rank=%d, world=%d, tile=%d)", rank, world, tileSize);
    if (rank != c->rank || world != c->world || tileSize != c->tile) c->orderValid = false;
    c->rank = rank;
    c->world = world;
    c->tile = tileSize;
    return RDH_OK;
}

int rdh_tiles_per_rank(const rdh_ctx *c) {
    if (!c || !c->haveCamera) return RDH_ERR_NO_SCENE;
    return makePixelMap(c).tilesPerRank;
}

int rdh_untile(rdh_ctx *c, const float *d_gathered, float *d_frame) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_gathered || !d_frame) return fail(c, RDH_ERR_ARGS, "rdh_untile: null buffer");
    PixelMap pm = makePixelMap(c);
    long long total = (long long)pm.W * pm.H;
    hipLaunchKernelGGL(k_untile, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_gathered, d_frame, pm.W,
                       pm.H, pm.tile, pm.tilesX, pm.numTiles, pm.world, pm.tilesPerRank, 3);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_path_trace(rdh_ctx *c, float *d_direct, float *d_indirect, int iter, int looper, int maxDepth, uint32_t flags) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_direct || !d_indirect || iter < 0 || looper < 0 || looper >= 10000 || maxDepth < 0)
        return fail(c, RDH_ERR_ARGS, "rdh_path_trace: bad arguments");
    if (4 + 7 * maxDepth > 200) return fail(c, RDH_ERR_ARGS, "maxDepth %d needs more than 200 Sobol dimensions", maxDepth);
    HIP_TRY(c, hipSetDevice(c->device));
    PixelMap pm = makePixelMap(c);
    const bool count = (flags & RDH_PT_COUNT) != 0;
    if (flags & RDH_PT_AUTO) {  // the structure by what this launch holds (radish_hip.h; measured: DESIGN 5d, 8)
        flags &= ~(RDH_PT_AUTO | RDH_PT_WAVEFRONT | RDH_PT_SORT_MATERIAL | RDH_PT_WF_SUBFRAMES | RDH_PT_PERSISTENT);
        const bool wavefront = c->ds.bvhSize >= 100000 && pm.numBlocks >= 12000;
        flags |= wavefront ? (RDH_PT_WAVEFRONT | RDH_PT_SORT_MATERIAL | RDH_PT_WF_SUBFRAMES) : RDH_PT_PERSISTENT;
    }
    if (flags & RDH_PT_WAVEFRONT) {
        rc = wavefrontEnsure(c, pm);
        if (rc) return rc;
        timeBegin(c);
        rc = wavefrontPathTrace(c, pm, d_direct, d_indirect, iter, looper, maxDepth, flags);
        if (rc) return rc;
        return timeEnd(c, "pathTrace (wavefront)");
    }
    if (flags & RDH_PT_PERSISTENT) {
        // one persistent launch: as many single-wave workgroups as stay resident (3 per SIMD at 165 VGPRs), never more than
        // there are 4-block groups of work
        // The grid must be fully resident: a workgroup that starts late would start its static first blocks late.
        unsigned groups = (((unsigned)(pm.numBlocks + 3) / 4 + 7u) / 8u) * 8u * 4u;  // one wave per workgroup
        if (c->persistGrid == 0) {
            int perCU = 0, cus = 0;
            HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_pt_persistent<false>, 64, 0));
            HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            if (perCU < 1) perCU = 1;
            c->persistGrid = (unsigned)(perCU * cus);
        }
        const bool pairs = usePairs(c, flags);
        if (pairs && c->persistGridPair == 0) {
            int perCU = 0, cus = 0;
            HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_pt_persistent<false, true>), 64, 0));
            HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            if (perCU < 1) perCU = 1;
            c->persistGridPair = (unsigned)(perCU * cus);
        }
        unsigned residentGrid = (pairs ? c->persistGridPair : c->persistGrid) / (unsigned)c->share;
        if (residentGrid < 8u) residentGrid = 8u;
        unsigned grid = groups < residentGrid ? groups : residentGrid;
        const int ovfDepth = c->ds.treeDepth + 1;
        if (pairs && (rc = ensureStackOverflow(c, (size_t)c->persistGridPair * 64 * (size_t)ovfDepth * 2))) return rc;
        // longest-paths-first schedule from the previous launch's per-block costs (same partition and resolution only)
        if (c->costBlocks != pm.numBlocks) {
            HIP_TRY(c, hipStreamSynchronize(c->sideStream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (int k = 0; k < 2; k++) {
                if (c->blockCost[k]) hipFree(c->blockCost[k]);
                if (c->blockOrder[k]) hipFree(c->blockOrder[k]);
                c->blockCost[k] = nullptr;
                c->blockOrder[k] = nullptr;
            }
            if (c->blockEma) hipFree(c->blockEma);
            c->blockEma = nullptr;
            c->costBlocks = 0;
            for (int k = 0; k < 2; k++) {
                HIP_TRY(c, hipMalloc((void **)&c->blockCost[k], sizeof(unsigned) * (size_t)pm.numBlocks));
                HIP_TRY(c, hipMalloc((void **)&c->blockOrder[k], sizeof(int) * (size_t)pm.numBlocks));
            }
            HIP_TRY(c, hipMalloc((void **)&c->blockEma, sizeof(unsigned) * (size_t)pm.numBlocks));
            HIP_TRY(c, hipMemsetAsync(c->blockEma, 0, sizeof(unsigned) * (size_t)pm.numBlocks, c->stream));
            c->costBlocks = pm.numBlocks;
            c->orderValid = false;
        }
        if (!c->orderValid) {  // restart: no schedule in flight, both cost buffers zero, two launches in plain order
            HIP_TRY(c, hipStreamSynchronize(c->sideStream));
            for (int k = 0; k < 2; k++)
                HIP_TRY(c, hipMemsetAsync(c->blockCost[k], 0, sizeof(unsigned) * (size_t)pm.numBlocks, c->stream));
            c->persistFrame = 0;
            c->orderValid = true;
        }
        const int n = c->persistFrame++;
        const int cb = n & 1;
        timeBegin(c);
        // the schedule of launch n - 2 (it ran beside launch n - 1) has written blockOrder[cb] and zeroed blockCost[cb]
        if (n >= 2) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evSched[cb], 0));
        const bool useOrder = n >= 2 && !(flags & RDH_PT_NO_SCHEDULE);
        HIP_TRY(c, hipMemsetAsync(c->dPersist, 0, offsetof(PersistCounters, deferred), c->stream));
        const int *order = useOrder ? c->blockOrder[cb] : nullptr;
        long pp = profBegin(c, flags);
        int2 *const ovf = reinterpret_cast<int2 *>(c->treeOvf);
        if (pairs && count)
            hipLaunchKernelGGL((k_pt_persistent<true, true>), dim3(grid), dim3(64), 0, c->stream, c->ds, c->cam, pm, looper, iter,
                               maxDepth, d_direct, d_indirect, c->dPersist, order, c->blockCost[cb], ovf, ovfDepth);
        else if (pairs)
            hipLaunchKernelGGL((k_pt_persistent<false, true>), dim3(grid), dim3(64), 0, c->stream, c->ds, c->cam, pm, looper, iter,
                               maxDepth, d_direct, d_indirect, c->dPersist, order, c->blockCost[cb], ovf, ovfDepth);
        else if (count)
            hipLaunchKernelGGL(k_pt_persistent<true>, dim3(grid), dim3(64), 0, c->stream, c->ds, c->cam, pm, looper, iter,
                               maxDepth, d_direct, d_indirect, c->dPersist, order, c->blockCost[cb], (int2 *)nullptr, 0);
        else
            hipLaunchKernelGGL(k_pt_persistent<false>, dim3(grid), dim3(64), 0, c->stream, c->ds, c->cam, pm, looper, iter,
                               maxDepth, d_direct, d_indirect, c->dPersist, order, c->blockCost[cb], (int2 *)nullptr, 0);
        profEnd(c, pp);
        // this launch's costs -> running means -> block order for launch n + 2, on the side stream
        HIP_TRY(c, hipEventRecord(c->evFrame, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->sideStream, c->evFrame, 0));
        hipLaunchKernelGGL(k_persist_schedule, dim3(1), dim3(kScheduleThreads), 0, c->sideStream, c->blockCost[cb], c->blockEma,
                           c->blockOrder[cb], pm.numBlocks);
        HIP_TRY(c, hipEventRecord(c->evSched[cb], c->sideStream));
        return timeEnd(c, "pathTrace (persistent)");
    }
    timeBegin(c);
    long pe = profBegin(c, flags);
    if (count)
        hipLaunchKernelGGL(k_path_trace_mega<true>, dim3(gridBlocks(pm)), dim3(64), 0, c->stream, c->ds, c->cam, pm, looper,
                           iter, maxDepth, d_direct, d_indirect);
    else
        hipLaunchKernelGGL(k_path_trace_mega<false>, dim3(gridBlocks(pm)), dim3(64), 0, c->stream, c->ds, c->cam, pm, looper,
                           iter, maxDepth, d_direct, d_indirect);
    profEnd(c, pe);
    return timeEnd(c, "pathTrace");
}

int rdh_path_trace_direct(rdh_ctx *c, float *d_direct, int iter, int looper, uint32_t flags) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_direct || iter < 0 || looper < 0 || looper >= 10000) return fail(c, RDH_ERR_ARGS, "rdh_path_trace_direct: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    PixelMap pm = makePixelMap(c);
    timeBegin(c);
    if (flags & RDH_PT_COUNT)
        hipLaunchKernelGGL(k_path_trace_direct<true>, dim3(gridBlocks(pm)), dim3(64), 0, c->stream, c->ds, c->cam, pm, looper,
                           iter, d_direct);
    else
        hipLaunchKernelGGL(k_path_trace_direct<false>, dim3(gridBlocks(pm)), dim3(64), 0, c->stream, c->ds, c->cam, pm, looper,
                           iter, d_direct);
    return timeEnd(c, "pathTraceDirect");
}

int rdh_gbuffer_render(rdh_ctx *c, const rdh_gbuffer *gb, uint32_t flags) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!gb || !gb->albedo || !gb->motion || (gb->frameIdx & ~1)) return fail(c, RDH_ERR_ARGS, "rdh_gbuffer_render: bad G-buffer");
    for (int k = 0; k < 2; k++)
        if (!gb->normal[k] || !gb->depth[k] || !gb->primId[k]) return fail(c, RDH_ERR_ARGS, "rdh_gbuffer_render: null plane");
    if (gb->width != c->cam.resx || gb->height != c->cam.resy)
        return fail(c, RDH_ERR_ARGS, "G-buffer %dx%d does not match camera %dx%d", gb->width, gb->height, c->cam.resx, c->cam.resy);
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = waitExchanges(c, true, false))) return rc;  // an exchange still writing `albedo` / `motion` (single-buffered planes)
    // ReSTIR needs the WHOLE frame's G-buffer on every rank: its temporal lookup follows motion vectors to arbitrary pixels
    // of the previous frame and its spatial lookup crosses tile borders.  Default on a partition: every rank renders the
    // whole frame (no exchange).  RDH_PT_PARTITION_GBUFFER: this rank renders the records of ITS tiles only (frame layout),
    // and rdh_gbuffer_exchange[_pack/_unpack] completes the planes with one all-gather of 36 B per pixel.
    PixelMap pm = makePixelMap(c);
    pm.packed = 0;  // G-buffer planes are always in frame layout
    if (c->world > 1 && !(flags & RDH_PT_PARTITION_GBUFFER)) {
        pm.rank = 0;
        pm.world = 1;
        pm.tilesPerRank = pm.numTiles;
        pm.packed = 0;
        pm.numBlocks = pm.numTiles * (pm.tile / 8) * (pm.tile / 8);
    }
    DCamera last = toDeviceCamera(gb->lastCam);
    GBufPtrs p{gb->albedo, gb->normal[gb->frameIdx], gb->motion, gb->depth[gb->frameIdx], gb->primId[gb->frameIdx],
               gb->width, gb->height};
    if (flags & RDH_PT_ONE_LANE_PER_PIXEL) {  // one lane per pixel for the whole launch (the reference's structure)
        timeBegin(c);
        if (flags & RDH_PT_COUNT)
            hipLaunchKernelGGL(k_gbuffer<true>, dim3(gridBlocks(pm)), dim3(64), 0, c->stream, c->ds, c->cam, last, pm, p);
        else
            hipLaunchKernelGGL(k_gbuffer<false>, dim3(gridBlocks(pm)), dim3(64), 0, c->stream, c->ds, c->cam, last, pm, p);
        return timeEnd(c, "renderGBuffer");
    }
    // persistent launch with lane refill: as many single-wave workgroups as stay resident, never more than there is work
    if (c->gbufGrid == 0) {
        int perCU = 0, cus = 0;
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_gbuffer_persistent<false, true>), 64, 0));
        HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        c->gbufGrid = (unsigned)((perCU < 1 ? 1 : perCU) * cus);
    }
    // Primary rays are coherent: while the six threaded arrays sit in the caches the pair walk buys them nothing (teapots, 201 k nodes:
    // 0.78 ms either way, profiles/r03_o_*); it is taken where those arrays outgrow the 256-MiB Infinity Cache (6 x 32 B x nodes).
    static const int gbufPairsEnv = getenv("RADISH_GBUFFER_PAIRS") ? atoi(getenv("RADISH_GBUFFER_PAIRS")) : -1;  // experiments
    const bool pairs = usePairs(c, flags) && (gbufPairsEnv >= 0 ? gbufPairsEnv != 0 : (size_t)c->ds.bvhSize * 192 > ((size_t)256 << 20));
    if (pairs && c->gbufGridPair == 0) {
        int perCU = 0, cus = 0;
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_gbuffer_persistent<false, true, true>), 64, 0));
        HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        c->gbufGridPair = (unsigned)((perCU < 1 ? 1 : perCU) * cus);
    }
    const unsigned resGrid = pairs ? c->gbufGridPair : c->gbufGrid;
    unsigned groups = (((unsigned)(pm.numBlocks + 3) / 4 + 7u) / 8u) * 8u * 4u;
    unsigned grid = groups < resGrid ? groups : resGrid;
    const int ovfDepth = c->ds.treeDepth + 1;
    if (pairs) {
        int rc2 = ensureStackOverflow(c, (size_t)resGrid * 64 * (size_t)ovfDepth * 2);
        if (rc2) return rc2;
    }
    int2 *const ovf = reinterpret_cast<int2 *>(c->treeOvf);
    timeBegin(c);
    HIP_TRY(c, hipMemsetAsync(c->dPersist, 0, offsetof(PersistCounters, deferred), c->stream));
    // literal-class rays are listed first and traced on a second stream beside the rest, each by a whole workgroup
    // (kernels_persist.h, k_gbuffer_literal): traced in place, ONE such ray (1.2 ms on the teapots scene) was the duration of
    // the pass.  RDH_PT_NO_DEFER keeps them in the persistent kernel.
    const bool defer = !(flags & RDH_PT_NO_DEFER);
    const bool count = (flags & RDH_PT_COUNT) != 0;
    if (defer) {
        const unsigned pixels = (unsigned)(c->cam.resx * c->cam.resy);
        hipLaunchKernelGGL(k_gbuffer_find_literal, dim3((pixels + 255u) / 256u), dim3(256), 0, c->stream, c->ds, c->cam, pm, c->dPersist);
        HIP_TRY(c, hipEventRecord(c->evFork, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->litStream, c->evFork, 0));
        if (count)
            hipLaunchKernelGGL(k_gbuffer_literal<true>, dim3(kDeferCap), dim3(kWgTraceThreads), 0, c->litStream, c->ds, c->cam, last, p, c->dPersist);
        else
            hipLaunchKernelGGL(k_gbuffer_literal<false>, dim3(kDeferCap), dim3(kWgTraceThreads), 0, c->litStream, c->ds, c->cam, last, p, c->dPersist);
        HIP_TRY(c, hipEventRecord(c->evJoin, c->litStream));
    }
    if (usePackets(flags)) {  // one 8x8 block per wave, its 64 centre rays walked as a packet (kernels_persist.h, k_gbuffer_packet)
        const unsigned pgrid = ((unsigned)pm.numBlocks + 3u) / 4u;
        if (count && defer) hipLaunchKernelGGL((k_gbuffer_packet<true, true>), dim3(pgrid), dim3(256), 0, c->stream, c->ds, c->cam, last, pm, p, c->dPersist,
                                          (flags & RDH_PT_WF_SMALL_LISTS) ? 4 : kPacketBudget);
        else if (count) hipLaunchKernelGGL((k_gbuffer_packet<true, false>), dim3(pgrid), dim3(256), 0, c->stream, c->ds, c->cam, last, pm, p, c->dPersist,
                                          (flags & RDH_PT_WF_SMALL_LISTS) ? 4 : kPacketBudget);
        else if (defer) hipLaunchKernelGGL((k_gbuffer_packet<false, true>), dim3(pgrid), dim3(256), 0, c->stream, c->ds, c->cam, last, pm, p, c->dPersist,
                                          (flags & RDH_PT_WF_SMALL_LISTS) ? 4 : kPacketBudget);
        else hipLaunchKernelGGL((k_gbuffer_packet<false, false>), dim3(pgrid), dim3(256), 0, c->stream, c->ds, c->cam, last, pm, p, c->dPersist,
                                          (flags & RDH_PT_WF_SMALL_LISTS) ? 4 : kPacketBudget);
        if (defer) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evJoin, 0));
        return timeEnd(c, "renderGBuffer");
    }
#define RD_LAUNCH_GBUF(CNT, DEF)                                                                                                             \
    do {                                                                                                                                     \
        if (pairs)                                                                                                                           \
            hipLaunchKernelGGL((k_gbuffer_persistent<CNT, DEF, true>), dim3(grid), dim3(64), 0, c->stream, c->ds, c->cam, last, pm, p,         \
                               c->dPersist, ovf, ovfDepth);                                                                                  \
        else                                                                                                                                 \
            hipLaunchKernelGGL((k_gbuffer_persistent<CNT, DEF, false>), dim3(grid), dim3(64), 0, c->stream, c->ds, c->cam, last, pm, p,        \
                               c->dPersist, (int2 *)nullptr, 0);                                                                             \
    } while (0)
    if (count && defer) RD_LAUNCH_GBUF(true, true);
    else if (count) RD_LAUNCH_GBUF(true, false);
    else if (defer) RD_LAUNCH_GBUF(false, true);
    else RD_LAUNCH_GBUF(false, false);
#undef RD_LAUNCH_GBUF
    if (defer) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evJoin, 0));
    return timeEnd(c, "renderGBuffer");
}

int rdh_restir_free(rdh_ctx *c) {
    if (!c) return RDH_ERR_ARGS;
    hipSetDevice(c->device);
    if (c->commStream) hipStreamSynchronize(c->commStream);  // a reservoir exchange may still be writing `last`
    c->resvPending = false;
    if (c->resvCur) hipFree(c->resvCur);
    if (c->resvLast) hipFree(c->resvLast);
    if (c->resvTemp) hipFree(c->resvTemp);
    if (c->restirState) hipFree(c->restirState);
    if (c->splitBuf) hipFree(c->splitBuf);
    c->splitBuf = nullptr;
    c->splitSlots = 0;
    c->splitLastSlots = 0;
    c->resvCur = c->resvLast = c->resvTemp = nullptr;
    c->restirState = nullptr;
    c->restirPixels = 0;
    return RDH_OK;
}

int rdh_restir_init(rdh_ctx *c) {
    if (!c) return RDH_ERR_ARGS;
    if (!c->haveCamera) return fail(c, RDH_ERR_NO_SCENE, "rdh_restir_init needs the camera resolution (rdh_set_camera)");
    HIP_TRY(c, hipSetDevice(c->device));
    rdh_restir_free(c);
    long long n = (long long)c->cam.resx * c->cam.resy;
    size_t bytes = (size_t)n * 36;
    HIP_TRY(c, hipMalloc((void **)&c->resvCur, bytes));
    HIP_TRY(c, hipMalloc((void **)&c->resvLast, bytes));
    HIP_TRY(c, hipMalloc((void **)&c->resvTemp, bytes));
    HIP_TRY(c, hipMalloc((void **)&c->restirState, (size_t)n * 48));
    HIP_TRY(c, hipMemsetAsync(c->resvCur, 0, bytes, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->resvLast, 0, bytes, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->resvTemp, 0, bytes, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->restirState, 0, (size_t)n * 48, c->stream));
    c->restirPixels = n;
    c->restirFirstFrame = true;
    return RDH_OK;
}

int rdh_restir_direct(rdh_ctx *c, float *d_direct, int iter, int looper, const rdh_gbuffer *gb, const rdh_restir_params *p,
                      uint32_t flags) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_direct || !gb || !p || iter < 0 || looper < 0 || looper >= 10000) return fail(c, RDH_ERR_ARGS, "rdh_restir_direct: bad arguments");
    if (!c->resvCur || c->restirPixels != (long long)c->cam.resx * c->cam.resy)
        return fail(c, RDH_ERR_STATE, "rdh_restir_direct before rdh_restir_init (or resolution changed)");
    if (gb->width != c->cam.resx || gb->height != c->cam.resy) return fail(c, RDH_ERR_ARGS, "G-buffer size mismatch");
    if (p->risCount < 0 || p->numSpatial < 0 || p->temporalClamp < 1) return fail(c, RDH_ERR_ARGS, "bad ReSTIR parameters");
    int dims = 4 + p->risCount * 5 + 1 + ((p->reuseMask & 1) ? 1 : 0) + ((p->reuseMask & 2) ? p->numSpatial * 3 + 1 : 0);
    if (dims > 200) return fail(c, RDH_ERR_ARGS, "ReSTIR parameters need %d Sobol dimensions (max 200)", dims);
    HIP_TRY(c, hipSetDevice(c->device));
    PixelMap pm = makePixelMap(c);
    int f = gb->frameIdx & 1;
    RestirArgs a;
    a.reservoirOut = c->resvCur;
    a.reservoirIn = c->resvLast;
    a.reservoirTemp = c->resvTemp;
    a.state = c->restirState;
    a.albedo = gb->albedo;
    a.normalCur = gb->normal[f];
    a.normalLast = gb->normal[f ^ 1];
    a.depthCur = gb->depth[f];
    a.motion = gb->motion;
    a.primIdCur = gb->primId[f];
    a.primIdLast = gb->primId[f ^ 1];
    a.gbWidth = gb->width;
    a.gbHeight = gb->height;
    a.firstFrame = c->restirFirstFrame ? 1 : 0;
    a.reuseMask = p->reuseMask;
    a.risCount = p->risCount;
    a.numSpatial = p->numSpatial;
    a.temporalClamp = p->temporalClamp;
    a.faithfulRIS = p->faithfulRIS;
    // On a tile partition with spatial reuse, pass 1 also covers an 8-pixel apron around this rank's tiles.
    int apronBlocks = 0;
    if (c->world > 1 && (p->reuseMask & 2)) {
        int bpe = pm.tile / 8 + 2;
        apronBlocks = pm.tilesPerRank * bpe * bpe;
    }
    const unsigned nBlocks1 = (unsigned)(apronBlocks > 0 ? apronBlocks : pm.numBlocks);
    timeBegin(c);
    if (flags & RDH_PT_RESTIR_FUSED) {
        // Round 1's pass 1: one lane per pixel for the whole launch, both walks inside it (k_restir_pass1).  A persistent
        // lane-refill version of THAT kernel was measured slower (3.21 against 2.90 ms): the RIS loop dominates it.
        const unsigned p1Threads = 64;  // single-wave workgroups, one 8x8 block each (kernels_restir.h)
        const unsigned grid1 = ((nBlocks1 + 7u) / 8u) * 8u;
        if ((rc = waitExchanges(c, true, true))) return rc;  // pass 1 reads the G-buffer and last frame's reservoirs (temporal reuse)
        if (flags & RDH_PT_COUNT)
            hipLaunchKernelGGL(k_restir_pass1<true>, dim3(grid1), dim3(p1Threads), 0, c->stream, c->ds, c->cam, pm, looper, iter, a, d_direct, apronBlocks);
        else
            hipLaunchKernelGGL(k_restir_pass1<false>, dim3(grid1), dim3(p1Threads), 0, c->stream, c->ds, c->cam, pm, looper, iter, a, d_direct, apronBlocks);
    } else {
        // raygen -> closest-hit walk -> RIS (LDS light table) -> any-hit walk -> resolve (kernels_restir.h)
        const long long slots = (long long)nBlocks1 * 64;
        constexpr size_t kSlotBytes = 24 + 16 + 24 + 4 + 36 + 48;
        constexpr size_t kDeferBytes = 16 + 2 * sizeof(int) * kRestirDeferCap;
        if (slots > c->splitSlots) {
            if (c->splitBuf) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                hipFree(c->splitBuf);
                c->splitBuf = nullptr;
                c->splitSlots = 0;
            }
            HIP_TRY(c, hipMalloc(&c->splitBuf, (size_t)slots * kSlotBytes + kDeferBytes));
            c->splitSlots = slots;
        }
        c->splitLastSlots = slots;
        RestirSplit sp;
        char *base = static_cast<char *>(c->splitBuf);
        sp.st = reinterpret_cast<float4 *>(base);                                  // 48 B, 16-B aligned first
        sp.hits = reinterpret_cast<int4 *>(base + (size_t)slots * 48);
        sp.rays = reinterpret_cast<float *>(base + (size_t)slots * (48 + 16));
        sp.segs = reinterpret_cast<float *>(base + (size_t)slots * (48 + 16 + 24));
        sp.rawResv = reinterpret_cast<float *>(base + (size_t)slots * (48 + 16 + 24 + 24));
        sp.occ = reinterpret_cast<int *>(base + (size_t)slots * (48 + 16 + 24 + 24 + 36));
        sp.deferCount = reinterpret_cast<int *>(base + (size_t)slots * kSlotBytes);
        sp.deferList = sp.deferCount + 4;
        const bool defer = !(flags & RDH_PT_NO_DEFER);
        HIP_TRY(c, hipMemsetAsync(sp.deferCount, 0, 16, c->stream));
        const bool count = (flags & RDH_PT_COUNT) != 0;
        const unsigned gridBlk = (nBlocks1 + 3u) / 4u;
        hipLaunchKernelGGL(k_restir_raygen, dim3(gridBlk), dim3(256), 0, c->stream, c->ds, c->cam, pm, looper, apronBlocks, sp.rays,
                           sp.deferCount, sp.deferList);
        const bool pairs = usePairs(c, flags);
        if ((rc = launchWalk(c, sp.rays, slots, sp.hits, nullptr, count, defer ? sp.deferCount : nullptr, sp.deferList, 1, pairs, usePackets(flags),
                             (flags & RDH_PT_WF_SMALL_LISTS) ? 4 : kPacketBudget))) return rc;
        const int nLights = c->ds.lightSamplerLength - (c->ds.envSamplerLength != 0 ? 1 : 0);
        const size_t ldsBytes = (size_t)nLights * sizeof(LightPre) + (size_t)c->ds.lightSamplerLength * sizeof(AliasRec);
        const bool staged = nLights > 0 && ldsBytes <= kRisLdsBytes;
        if (c->risGrid == 0) {
            int cus = 0;
            HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            c->risGrid = (unsigned)(2 * cus);  // two 512-thread workgroups per CU (74 KB of LDS each at 1 026 lights)
        }
        const unsigned wgNeeded = (nBlocks1 + (kRisThreads / 64) - 1) / (kRisThreads / 64);
        const unsigned risGrid = wgNeeded < c->risGrid ? wgNeeded : c->risGrid;
        if (staged)
            hipLaunchKernelGGL(k_restir_ris<true>, dim3(risGrid), dim3(kRisThreads), ldsBytes, c->stream, c->ds, c->cam, pm, looper, a, apronBlocks, sp);
        else
            hipLaunchKernelGGL(k_restir_ris<false>, dim3(risGrid), dim3(kRisThreads), 0, c->stream, c->ds, c->cam, pm, looper, a, apronBlocks, sp);
        if ((rc = launchWalk(c, sp.segs, slots, nullptr, sp.occ, count, defer ? sp.deferCount + 1 : nullptr, sp.deferList + kRestirDeferCap, 1, pairs)))
            return rc;
        // Only from here on are the G-buffer planes and last frame's reservoirs read (temporal reuse in the resolve step, spatial
        // reuse in pass 2): exchanges of either that are still in flight on the communication stream have had the ray generation,
        // both walks and the RIS launch to hide behind.
        if ((rc = waitExchanges(c, true, true))) return rc;
        hipLaunchKernelGGL(k_restir_resolve, dim3(gridBlk), dim3(256), 0, c->stream, c->ds, pm, iter, a, apronBlocks, sp, d_direct);
    }
    if (p->reuseMask & 2)
        hipLaunchKernelGGL(k_restir_pass2, dim3(gridFor(pm)), dim3(256), 0, c->stream, c->ds, pm, iter, a, d_direct);
    rc = timeEnd(c, "ReSTIR Direct");
    std::swap(c->resvCur, c->resvLast);  // restir.cu:221
    c->restirFirstFrame = false;         // :223-225
    return rc;
}

int rdh_restir_exchange_pack(rdh_ctx *c, float *d_packed) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_packed) return fail(c, RDH_ERR_ARGS, "rdh_restir_exchange_pack: null buffer");
    if (!c->resvLast) return fail(c, RDH_ERR_STATE, "ReSTIR not initialised");
    PixelMap pm = makePixelMap(c);
    hipLaunchKernelGGL(k_pack_tiles, dim3((unsigned)(pm.numBlocks + 3) / 4), dim3(256), 0, c->stream, c->resvLast, d_packed, pm, 9);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_restir_exchange_unpack(rdh_ctx *c, const float *d_gathered) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_gathered) return fail(c, RDH_ERR_ARGS, "rdh_restir_exchange_unpack: null buffer");
    if (!c->resvLast) return fail(c, RDH_ERR_STATE, "ReSTIR not initialised");
    PixelMap pm = makePixelMap(c);
    long long total = (long long)pm.W * pm.H;
    hipLaunchKernelGGL(k_untile, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_gathered, c->resvLast, pm.W,
                       pm.H, pm.tile, pm.tilesX, pm.numTiles, pm.world, pm.tilesPerRank, 9);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_restir_read(rdh_ctx *c, int which, void *hostOut) {
    if (!c || !hostOut || which < 0 || which > 2) return RDH_ERR_ARGS;
    if (!c->resvCur) return fail(c, RDH_ERR_STATE, "ReSTIR not initialised");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->commStream) HIP_TRY(c, hipStreamSynchronize(c->commStream));
    const float *src = which == 0 ? c->resvCur : (which == 1 ? c->resvLast : c->resvTemp);
    HIP_TRY(c, hipMemcpy(hostOut, src, (size_t)c->restirPixels * 36, hipMemcpyDeviceToHost));
    return RDH_OK;
}

long long rdh_restir_read_scratch(rdh_ctx *c, int which, void *hostOut, long long maxBytes) {
    if (!c || which < 0 || which > 2) return RDH_ERR_ARGS;
    if (!c->splitBuf) return fail(c, RDH_ERR_STATE, "rdh_restir_read_scratch: no split pass 1 has run");
    if (c->splitLastSlots <= 0) return fail(c, RDH_ERR_STATE, "rdh_restir_read_scratch: no split pass 1 has run");
    const size_t slots = (size_t)c->splitLastSlots;  // the planes were laid out with THIS stride (not the allocation's)
    const char *base = static_cast<const char *>(c->splitBuf);
    const char *src = which == 0 ? base + slots * (48 + 16) : (which == 1 ? base + slots * (48 + 16 + 24) : base + slots * 152);
    const size_t bytes = which == 2 ? 16 + 2 * sizeof(int) * kRestirDeferCap : slots * 24;
    if (!hostOut) return (long long)bytes;
    if (maxBytes < (long long)bytes) return fail(c, RDH_ERR_ARGS, "rdh_restir_read_scratch: buffer too small");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(hostOut, src, bytes, hipMemcpyDeviceToHost));
    return (long long)bytes;
}

// ---- collectives: RCCL over xGMI, called from C++ (no reference counterpart) -----------------------
// One process per GPU: every rank creates its context, rank 0 makes a 128-byte id (rdh_comm_unique_id) and hands it to the
// others by whatever channel the host has (the Python harness broadcasts it over torch.distributed's store, a C++ host can
// use a file or MPI), every rank calls rdh_comm_init.  RCCL is bound at run time (dlopen), not at link time, so that the
// library loads — and every single-GPU entry point works — on a machine without RCCL, and so that inside a PyTorch process
// the communicator lives in the RCCL copy that process already holds.  Data path: ONE ncclAllGather per image (or per
// reservoir / G-buffer set) per frame on the context's stream, then a local un-tile kernel.  xGMI is point-to-point (7 links
// per GPU), so the gather's per-peer shards travel on different links concurrently; bytes per rank are in DESIGN §8.
struct Id128 { char internal[128]; };  // == ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value
extern "C++" {
namespace {
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
RcclApi g_rccl;
constexpr int kNcclFloat = 7;  // ncclFloat32 (rccl.h ncclDataType_t)

// Resolved once per process (std::call_once: a host that drives several GPUs from one thread each may get here concurrently).
// Order: RADISH_RCCL_LIB; the nccl* symbols the process already holds in its global scope (a statically linked RCCL, or one
// loaded RTLD_GLOBAL) — resolved with dlsym(RTLD_DEFAULT, ...), no second copy is loaded; a librccl the process has loaded
// privately (PyTorch's: RTLD_NOLOAD finds it); finally ROCm's own.
std::once_flag g_rcclOnce;
const char *g_rcclWhy = nullptr;
void rcclLoadOnce() {
    void *h = nullptr;
    bool global = false;
    const char *env = getenv("RADISH_RCCL_LIB");
    if (env && *env) {
        h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            g_rcclWhy = "RADISH_RCCL_LIB could not be loaded";
            return;
        }
    }
    if (!h && dlsym(RTLD_DEFAULT, "ncclAllGather") && dlsym(RTLD_DEFAULT, "ncclCommInitRank")) global = true;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (int k = 0; !h && !global && k < 2; k++) h = dlopen(names[k], RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    for (int k = 0; !h && !global && k < 2; k++) h = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
    if (!h && !global) {
        g_rcclWhy = "librccl.so not found (set RADISH_RCCL_LIB)";
        return;
    }
    void *scope = global ? RTLD_DEFAULT : h;  // RTLD_DEFAULT is a null handle on glibc: `global` carries the decision
    g_rccl.handle = h;
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(scope, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(scope, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(scope, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(scope, "ncclGetErrorString"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(dlsym(scope, "ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(dlsym(scope, "ncclGroupEnd"));
    auto ag = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(scope, "ncclAllGather"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd || !ag) {
        g_rcclWhy = "librccl.so lacks the nccl* entry points";
        return;
    }
    g_rccl.AllGather = ag;
}
const char *rcclLoad() {
    std::call_once(g_rcclOnce, rcclLoadOnce);
    return g_rccl.AllGather ? nullptr : (g_rcclWhy ? g_rcclWhy : "RCCL could not be loaded");
}

int rcclFail(rdh_ctx *c, int r, const char *what) {
    return fail(c, RDH_ERR_COMM, "RCCL error: %s: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", r);
}

int commEnsure(rdh_ctx *c, float **buf, size_t *have, size_t floats) {
    if (*have >= floats) return RDH_OK;
    if (*buf) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->commStream) HIP_TRY(c, hipStreamSynchronize(c->commStream));
        hipFree(*buf);
        *buf = nullptr;
        *have = 0;
    }
    HIP_TRY(c, hipMalloc((void **)buf, floats * sizeof(float)));
    *have = floats;
    return RDH_OK;
}

// ---- the communication stream of ReSTIR's exchanges ----
int commStreamEnsure(rdh_ctx *c) {
    if (c->commStream) return RDH_OK;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->commStream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->evToComm, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->evGbuf, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->evImg, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->evResv, hipEventDisableTiming));
    return RDH_OK;
}
bool overlapOn(const rdh_ctx *c) { return c->commOverlap && c->commStream != nullptr; }
// Launches between construction and destruction go to the communication stream (every helper launches on c->stream); with the
// overlap switched off this is a no-op and everything stays on the render stream, in the same order.
struct OnCommStream {
    rdh_ctx *c;
    hipStream_t saved;
    explicit OnCommStream(rdh_ctx *ctx) : c(ctx), saved(ctx->stream) {
        if (overlapOn(c)) c->stream = c->commStream;
    }
    ~OnCommStream() { c->stream = saved; }
};
// render stream -> communication stream: what was enqueued on the render stream so far happens before what follows on commStream
int toComm(rdh_ctx *c) {
    if (!overlapOn(c)) return RDH_OK;
    HIP_TRY(c, hipEventRecord(c->evToComm, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->commStream, c->evToComm, 0));
    return RDH_OK;
}
// the render stream waits for exchanges still in flight (before it reads G-buffer planes / `last` reservoirs they complete)
int waitExchanges(rdh_ctx *c, bool gbuf, bool resv) {
    if (gbuf && c->gbufPending) {
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evGbuf, 0));
        c->gbufPending = false;
    }
    if (resv && c->resvPending) {
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evResv, 0));
        c->resvPending = false;
    }
    return RDH_OK;
}
int joinComm(rdh_ctx *c) { return c ? waitExchanges(c, true, true) : RDH_OK; }

// send: this rank's packed tiles, `channels` floats per pixel -> c->commRecv: the packed tiles of every rank, rank-major
int gatherEnsure(rdh_ctx *c, int channels, const PixelMap &pm) {  // may allocate (and then synchronises): never inside a group
    if (!c->comm) return fail(c, RDH_ERR_STATE, "no communicator: call rdh_comm_init first");
    const size_t shard = (size_t)pm.tilesPerRank * pm.tile * pm.tile * channels;
    return commEnsure(c, &c->commRecv, &c->commRecvFloats, shard * (size_t)pm.world);
}
int gatherEnqueue(rdh_ctx *c, const float *d_send, int channels, const PixelMap &pm) {
    const size_t shard = (size_t)pm.tilesPerRank * pm.tile * pm.tile * channels;
    int r = g_rccl.AllGather(d_send, c->commRecv, shard, kNcclFloat, c->comm, c->stream);
    if (r != 0) return rcclFail(c, r, "ncclAllGather");
    return RDH_OK;
}
int allGatherPacked(rdh_ctx *c, const float *d_send, int channels, const PixelMap &pm) {
    int rc = gatherEnsure(c, channels, pm);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    return gatherEnqueue(c, d_send, channels, pm);
}
}  // namespace
}  // extern "C++"

int rdh_comm_unique_id(void *id128) {
    if (!id128) return RDH_ERR_ARGS;
    if (rcclLoad()) return RDH_ERR_COMM;
    return g_rccl.GetUniqueId(id128) == 0 ? RDH_OK : RDH_ERR_COMM;
}

int rdh_comm_init(rdh_ctx *c, const void *id128, int rank, int world) {
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return c ? fail(c, RDH_ERR_ARGS, "rdh_comm_init(rank=%d, world=%d)", rank, world) : RDH_ERR_ARGS;
    const char *why = rcclLoad();
    if (why) return fail(c, RDH_ERR_COMM, "%s", why);
    HIP_TRY(c, hipSetDevice(c->device));
    rdh_comm_destroy(c);
    Id128 id;
    memcpy(&id, id128, sizeof(id));
    void *comm = nullptr;
    int r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != 0) return rcclFail(c, r, "ncclCommInitRank");
    c->comm = comm;
    return rdh_set_partition(c, rank, world, c->tile);
}

// One PROCESS, n devices (the reference's host is one process with one frame loop, main.cpp:163-202): the
// communicators of all n contexts are created inside one RCCL group, so a single host thread can do it (ncclCommInitRank by
// itself blocks until every rank has arrived).  ctxs[i] becomes rank i of n; its tile partition is set accordingly.
int rdh_comm_init_all(rdh_ctx **ctxs, int n) {
    if (!ctxs || n < 1) return RDH_ERR_ARGS;
    for (int i = 0; i < n; i++)
        if (!ctxs[i]) return RDH_ERR_ARGS;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (ctxs[i] == ctxs[j] || ctxs[i]->device == ctxs[j]->device)
                return fail(ctxs[0], RDH_ERR_ARGS, "rdh_comm_init_all: contexts %d and %d share a device (RCCL wants one rank per GPU)", j, i);
    rdh_ctx *c0 = ctxs[0];
    const char *why = rcclLoad();
    if (why) return fail(c0, RDH_ERR_COMM, "%s", why);
    Id128 id;
    int r = g_rccl.GetUniqueId(&id);
    if (r != 0) return rcclFail(c0, r, "ncclGetUniqueId");
    for (int i = 0; i < n; i++) rdh_comm_destroy(ctxs[i]);
    std::vector<void *> comms((size_t)n, nullptr);
    if ((r = g_rccl.GroupStart()) != 0) return rcclFail(c0, r, "ncclGroupStart");
    int rInit = 0;
    for (int i = 0; i < n && rInit == 0; i++) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) rInit = -1;
        else rInit = g_rccl.CommInitRank(&comms[(size_t)i], n, id, i);
    }
    r = g_rccl.GroupEnd();  // the group is always closed
    if (rInit != 0 || r != 0) {
        for (int i = 0; i < n; i++)
            if (comms[(size_t)i]) g_rccl.CommDestroy(comms[(size_t)i]);
        return rInit == -1 ? fail(c0, RDH_ERR_NO_DEVICE, "rdh_comm_init_all: hipSetDevice failed") : rcclFail(c0, rInit != 0 ? rInit : r, "ncclCommInitRank (grouped)");
    }
    for (int i = 0; i < n; i++) {
        ctxs[i]->comm = comms[(size_t)i];
        int rc = rdh_set_partition(ctxs[i], i, n, ctxs[i]->tile);
        if (rc) return rc;
    }
    return RDH_OK;
}

int rdh_comm_destroy(rdh_ctx *c) {
    if (!c) return RDH_ERR_ARGS;
    hipSetDevice(c->device);
    if (c->commStream) hipStreamSynchronize(c->commStream);
    c->gbufPending = c->resvPending = false;
    if (c->comm) {
        hipStreamSynchronize(c->stream);
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    for (int k = 0; k < 2; k++) {
        if (c->commSend[k]) hipFree(c->commSend[k]);
        c->commSend[k] = nullptr;
        c->commSendFloats[k] = 0;
    }
    if (c->commRecv) hipFree(c->commRecv);
    c->commRecv = nullptr;
    c->commRecvFloats = 0;
    return RDH_OK;
}

int rdh_allgather_tiles(rdh_ctx *c, const float *d_packed, float *d_frame) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_packed || !d_frame) return fail(c, RDH_ERR_ARGS, "rdh_allgather_tiles: null buffer");
    HIP_TRY(c, hipSetDevice(c->device));
    PixelMap pm = makePixelMap(c);
    if ((rc = allGatherPacked(c, d_packed, 3, pm))) return rc;
    return rdh_untile(c, c->commRecv, d_frame);
}

extern "C++" {
namespace {
// frame-layout image -> this rank's packed tiles in commSend[k] (the running mean of pathtrace.cu:287-290 reads the caller's
// image, so the packed render must start from it)
int packFrame(rdh_ctx *c, const float *d_frame, int k, int channels, const PixelMap &pm) {
    const size_t shard = (size_t)pm.tilesPerRank * pm.tile * pm.tile * channels;
    int rc = commEnsure(c, &c->commSend[k], &c->commSendFloats[k], shard);
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack_tiles, dim3((unsigned)(pm.numBlocks + 3) / 4), dim3(256), 0, c->stream, d_frame, c->commSend[k], pm, channels);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}
}  // namespace
}

extern "C++" {
namespace {
// rdh_path_trace_gathered, first half: the caller's frames -> this rank's packed tiles (the running mean reads them), render.
int gatheredRender(rdh_ctx *c, const float *d_directFrame, const float *d_indirectFrame, int iter, int looper, int maxDepth, uint32_t flags) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_directFrame || !d_indirectFrame) return fail(c, RDH_ERR_ARGS, "rdh_path_trace_gathered: null image");
    if (!c->comm) return fail(c, RDH_ERR_STATE, "no communicator: call rdh_comm_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    c->forcePacked = true;  // one rank: still render into packed tiles, so that the same path runs at every world size
    PixelMap pm = makePixelMap(c);
    if (!(rc = packFrame(c, d_directFrame, 0, 3, pm)) && !(rc = packFrame(c, d_indirectFrame, 1, 3, pm)))
        rc = rdh_path_trace(c, c->commSend[0], c->commSend[1], iter, looper, maxDepth, flags);
    c->forcePacked = false;
    if (rc) return rc;
    return gatherEnsure(c, 3, pm);
}
PixelMap packedMap(rdh_ctx *c) {
    PixelMap pm = makePixelMap(c);
    pm.packed = 1;
    return pm;
}
}  // namespace
}

int rdh_path_trace_gathered(rdh_ctx *c, float *d_directFrame, float *d_indirectFrame, int iter, int looper, int maxDepth,
                            uint32_t flags) {
    int rc = gatheredRender(c, d_directFrame, d_indirectFrame, iter, looper, maxDepth, flags);
    if (rc) return rc;
    const PixelMap pm = packedMap(c);
    if ((rc = gatherEnqueue(c, c->commSend[0], 3, pm))) return rc;
    if ((rc = rdh_untile(c, c->commRecv, d_directFrame))) return rc;
    if ((rc = gatherEnqueue(c, c->commSend[1], 3, pm))) return rc;
    return rdh_untile(c, c->commRecv, d_indirectFrame);
}

// The same for n contexts of ONE process (rdh_comm_init_all): every context's render is enqueued first (asynchronous, one stream
// per device: the GPUs render side by side), then one RCCL group per image holds the n all-gathers — issued one after the other
// outside a group, the first would wait for peers whose call the same thread has not made yet.
int rdh_path_trace_gathered_all(rdh_ctx **ctxs, int n, float *const *d_directFrames, float *const *d_indirectFrames, int iter, int looper,
                                int maxDepth, uint32_t flags) {
    if (!ctxs || n < 1 || !d_directFrames || !d_indirectFrames) return RDH_ERR_ARGS;
    for (int i = 0; i < n; i++)
        if (!ctxs[i]) return RDH_ERR_ARGS;
    for (int i = 0; i < n; i++) {
        if (ctxs[i]->world != n || ctxs[i]->rank != i)
            return fail(ctxs[i], RDH_ERR_STATE, "rdh_path_trace_gathered_all: context %d is rank %d of %d (rdh_comm_init_all first)", i, ctxs[i]->rank, ctxs[i]->world);
        int rc = gatheredRender(ctxs[i], d_directFrames[i], d_indirectFrames[i], iter, looper, maxDepth, flags);
        if (rc) return rc;
    }
    for (int k = 0; k < 2; k++) {
        int r = g_rccl.GroupStart(), rc = RDH_OK;
        if (r != 0) return rcclFail(ctxs[0], r, "ncclGroupStart");
        for (int i = 0; i < n && rc == RDH_OK; i++) {
            if (hipSetDevice(ctxs[i]->device) != hipSuccess) rc = fail(ctxs[i], RDH_ERR_NO_DEVICE, "hipSetDevice(%d)", ctxs[i]->device);
            else rc = gatherEnqueue(ctxs[i], ctxs[i]->commSend[k], 3, packedMap(ctxs[i]));
        }
        r = g_rccl.GroupEnd();
        if (rc) return rc;
        if (r != 0) return rcclFail(ctxs[0], r, "ncclGroupEnd");
        for (int i = 0; i < n; i++)
            if ((rc = rdh_untile(ctxs[i], ctxs[i]->commRecv, k == 0 ? d_directFrames[i] : d_indirectFrames[i]))) return rc;
    }
    return RDH_OK;
}

extern "C++" {
namespace {
// ---- ReSTIRDirect on a partition, in phases (so that the one-process form can put each collective of n contexts into one RCCL
// group).  Render stream: pack the caller's image, the ReSTIR kernels.  Communication stream (overlap on; else the render stream):
// image gather + un-tile, then the reservoir exchange for the NEXT frame's temporal reuse. ----
int restirRender(rdh_ctx *c, float *d_directFrame, int iter, int looper, const rdh_gbuffer *gb, const rdh_restir_params *p, uint32_t flags) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_directFrame) return fail(c, RDH_ERR_ARGS, "rdh_restir_direct_gathered: null image");
    if (!c->comm) return fail(c, RDH_ERR_STATE, "no communicator: call rdh_comm_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->commOverlap && (rc = commStreamEnsure(c))) return rc;
    const PixelMap pm = packedMap(c);
    const size_t shard = (size_t)pm.tilesPerRank * pm.tile * pm.tile;
    if ((rc = commEnsure(c, &c->commSend[0], &c->commSendFloats[0], shard * 9))) return rc;  // allocations never happen inside a group
    if ((rc = commEnsure(c, &c->commSend[1], &c->commSendFloats[1], shard * 3))) return rc;
    if ((rc = gatherEnsure(c, 9, pm))) return rc;
    c->forcePacked = true;
    if (!(rc = packFrame(c, d_directFrame, 1, 3, makePixelMap(c)))) rc = rdh_restir_direct(c, c->commSend[1], iter, looper, gb, p, flags);
    c->forcePacked = false;
    if (rc) return rc;
    return toComm(c);
}
int restirImageGather(rdh_ctx *c) {
    OnCommStream on(c);
    return gatherEnqueue(c, c->commSend[1], 3, packedMap(c));
}
int restirImageEnd(rdh_ctx *c, float *d_directFrame) {  // + the send side of the reservoir exchange
    OnCommStream on(c);
    int rc = rdh_untile(c, c->commRecv, d_directFrame);
    if (rc) return rc;
    if (overlapOn(c)) HIP_TRY(c, hipEventRecord(c->evImg, c->commStream));
    return packFrame(c, c->resvLast, 0, 9, packedMap(c));
}
int restirResvGather(rdh_ctx *c) {
    OnCommStream on(c);
    return gatherEnqueue(c, c->commSend[0], 9, packedMap(c));
}
int restirResvEnd(rdh_ctx *c, bool imageToo) {
    {
        OnCommStream on(c);
        int rc = rdh_restir_exchange_unpack(c, c->commRecv);
        if (rc) return rc;
    }
    if (overlapOn(c)) {
        HIP_TRY(c, hipEventRecord(c->evResv, c->commStream));
        c->resvPending = true;  // the next rdh_restir_direct waits where it first reads `last` reservoirs
        // the caller's image is complete in render-stream order on return; the reservoir exchange goes on beside what comes next
        if (imageToo) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evImg, 0));
    }
    return RDH_OK;
}
int checkAll(rdh_ctx **ctxs, int n, const char *what) {
    if (!ctxs || n < 1) return RDH_ERR_ARGS;
    for (int i = 0; i < n; i++)
        if (!ctxs[i]) return RDH_ERR_ARGS;
    for (int i = 0; i < n; i++)
        if (ctxs[i]->world != n || ctxs[i]->rank != i || !ctxs[i]->comm)
            return fail(ctxs[i], RDH_ERR_STATE, "%s: context %d is rank %d of %d (rdh_comm_init_all first)", what, i, ctxs[i]->rank, ctxs[i]->world);
    return RDH_OK;
}
// one RCCL group around `enqueue(ctx)` for every context
template <typename F>
int grouped(rdh_ctx **ctxs, int n, F enqueue) {
    int r = g_rccl.GroupStart(), rc = RDH_OK;
    if (r != 0) return rcclFail(ctxs[0], r, "ncclGroupStart");
    for (int i = 0; i < n && rc == RDH_OK; i++) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) rc = fail(ctxs[i], RDH_ERR_NO_DEVICE, "hipSetDevice(%d)", ctxs[i]->device);
        else rc = enqueue(ctxs[i]);
    }
    r = g_rccl.GroupEnd();  // always closed
    if (rc) return rc;
    if (r != 0) return rcclFail(ctxs[0], r, "ncclGroupEnd");
    return RDH_OK;
}
}  // namespace
}

int rdh_restir_exchange(rdh_ctx *c) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!c->resvLast) return fail(c, RDH_ERR_STATE, "ReSTIR not initialised");
    if (!c->comm) return fail(c, RDH_ERR_STATE, "no communicator: call rdh_comm_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->commOverlap && (rc = commStreamEnsure(c))) return rc;
    const PixelMap pm = packedMap(c);
    if ((rc = commEnsure(c, &c->commSend[0], &c->commSendFloats[0], (size_t)pm.tilesPerRank * pm.tile * pm.tile * 9))) return rc;
    if ((rc = gatherEnsure(c, 9, pm))) return rc;
    if ((rc = toComm(c))) return rc;
    {
        OnCommStream on(c);
        if ((rc = packFrame(c, c->resvLast, 0, 9, pm))) return rc;
    }
    if ((rc = restirResvGather(c))) return rc;
    return restirResvEnd(c, false);
}

int rdh_restir_direct_gathered(rdh_ctx *c, float *d_directFrame, int iter, int looper, const rdh_gbuffer *gb,
                               const rdh_restir_params *p, uint32_t flags) {
    int rc = restirRender(c, d_directFrame, iter, looper, gb, p, flags);
    if (rc) return rc;
    if ((rc = restirImageGather(c))) return rc;
    if ((rc = restirImageEnd(c, d_directFrame))) return rc;
    if ((rc = restirResvGather(c))) return rc;  // next frame's temporal reuse reads the whole frame's reservoirs
    return restirResvEnd(c, true);
}

int rdh_restir_direct_gathered_all(rdh_ctx **ctxs, int n, float *const *d_directFrames, int iter, int looper, const rdh_gbuffer *gbs,
                                   const rdh_restir_params *p, uint32_t flags) {
    int rc = checkAll(ctxs, n, "rdh_restir_direct_gathered_all");
    if (rc) return rc;
    if (!d_directFrames || !gbs) return RDH_ERR_ARGS;
    for (int i = 0; i < n; i++)
        if ((rc = restirRender(ctxs[i], d_directFrames[i], iter, looper, &gbs[i], p, flags))) return rc;
    if ((rc = grouped(ctxs, n, restirImageGather))) return rc;
    for (int i = 0; i < n; i++)
        if ((rc = restirImageEnd(ctxs[i], d_directFrames[i]))) return rc;
    if ((rc = grouped(ctxs, n, restirResvGather))) return rc;
    for (int i = 0; i < n; i++)
        if ((rc = restirResvEnd(ctxs[i], true))) return rc;
    return RDH_OK;
}

int rdh_comm_set_overlap(rdh_ctx *c, int enable) {
    if (!c) return RDH_ERR_ARGS;
    int rc = rdh_synchronize(c);  // nothing of the other mode is left in flight
    if (rc) return rc;
    c->commOverlap = enable != 0;
    return RDH_OK;
}

int rdh_comm_join(rdh_ctx *c) {
    if (!c) return RDH_ERR_ARGS;
    return joinComm(c);
}

extern "C++" {
namespace {
int gbufCheck(rdh_ctx *c, const rdh_gbuffer *gb, const char *what) {
    if (!gb || !gb->albedo || !gb->motion || (gb->frameIdx & ~1)) return fail(c, RDH_ERR_ARGS, "%s: bad G-buffer", what);
    for (int k = 0; k < 2; k++)
        if (!gb->normal[k] || !gb->depth[k] || !gb->primId[k]) return fail(c, RDH_ERR_ARGS, "%s: null plane", what);
    if (gb->width != c->cam.resx || gb->height != c->cam.resy) return fail(c, RDH_ERR_ARGS, "%s: G-buffer size does not match the camera", what);
    return RDH_OK;
}
}  // namespace
}

int rdh_gbuffer_exchange_pack(rdh_ctx *c, const rdh_gbuffer *gb, float *d_packed) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_packed) return fail(c, RDH_ERR_ARGS, "rdh_gbuffer_exchange_pack: null buffer");
    if ((rc = gbufCheck(c, gb, "rdh_gbuffer_exchange_pack"))) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    PixelMap pm = makePixelMap(c);
    const int f = gb->frameIdx;
    hipLaunchKernelGGL(k_gbuf_pack, dim3((unsigned)(pm.numBlocks + 3) / 4), dim3(256), 0, c->stream, gb->albedo, gb->normal[f], gb->motion,
                       gb->depth[f], gb->primId[f], d_packed, pm);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_gbuffer_exchange_unpack(rdh_ctx *c, const rdh_gbuffer *gb, const float *d_gathered) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!d_gathered) return fail(c, RDH_ERR_ARGS, "rdh_gbuffer_exchange_unpack: null buffer");
    if ((rc = gbufCheck(c, gb, "rdh_gbuffer_exchange_unpack"))) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    PixelMap pm = makePixelMap(c);
    const int f = gb->frameIdx;
    const long long total = (long long)pm.W * pm.H;
    hipLaunchKernelGGL(k_gbuf_unpack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_gathered, gb->albedo, gb->normal[f],
                       gb->motion, gb->depth[f], gb->primId[f], pm.W, pm.H, pm.tile, pm.tilesX, pm.world, pm.tilesPerRank);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

extern "C++" {
namespace {
int gbufExchangeBegin(rdh_ctx *c, const rdh_gbuffer *gb) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!c->comm) return fail(c, RDH_ERR_STATE, "no communicator: call rdh_comm_init first");
    if ((rc = gbufCheck(c, gb, "rdh_gbuffer_exchange"))) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->commOverlap && (rc = commStreamEnsure(c))) return rc;
    const PixelMap pm = packedMap(c);
    if ((rc = commEnsure(c, &c->commSend[0], &c->commSendFloats[0], (size_t)pm.tilesPerRank * pm.tile * pm.tile * 9))) return rc;
    if ((rc = gatherEnsure(c, 9, pm))) return rc;
    if ((rc = toComm(c))) return rc;  // after the G-buffer pass that filled this rank's records
    OnCommStream on(c);
    return rdh_gbuffer_exchange_pack(c, gb, c->commSend[0]);
}
int gbufExchangeGather(rdh_ctx *c) {
    OnCommStream on(c);
    return gatherEnqueue(c, c->commSend[0], 9, packedMap(c));
}
int gbufExchangeEnd(rdh_ctx *c, const rdh_gbuffer *gb) {
    {
        OnCommStream on(c);
        int rc = rdh_gbuffer_exchange_unpack(c, gb, c->commRecv);
        if (rc) return rc;
    }
    if (overlapOn(c)) {
        HIP_TRY(c, hipEventRecord(c->evGbuf, c->commStream));
        c->gbufPending = true;  // rdh_restir_direct waits where it first reads the planes; other readers: rdh_comm_join
    }
    return RDH_OK;
}
}  // namespace
}

int rdh_gbuffer_exchange(rdh_ctx *c, const rdh_gbuffer *gb) {
    int rc = gbufExchangeBegin(c, gb);
    if (rc) return rc;
    if ((rc = gbufExchangeGather(c))) return rc;
    return gbufExchangeEnd(c, gb);
}

int rdh_gbuffer_exchange_all(rdh_ctx **ctxs, int n, const rdh_gbuffer *gbs) {
    int rc = checkAll(ctxs, n, "rdh_gbuffer_exchange_all");
    if (rc) return rc;
    if (!gbs) return RDH_ERR_ARGS;
    for (int i = 0; i < n; i++)
        if ((rc = gbufExchangeBegin(ctxs[i], &gbs[i]))) return rc;
    if ((rc = grouped(ctxs, n, gbufExchangeGather))) return rc;
    for (int i = 0; i < n; i++)
        if ((rc = gbufExchangeEnd(ctxs[i], &gbs[i]))) return rc;
    return RDH_OK;
}

// RDH_PT_PERSISTENT on the ray-batch entries: the walk-only lane-refill kernel (device/kernels_walk.h); d_hits xor d_occ
extern "C++" {
static int walkPersistent(rdh_ctx *c, const float *d_rays, int64_t n, int4 *d_hits, int *d_occ, uint32_t flags, const char *what) {
    const int any = d_occ ? 1 : 0;
    if (flags & RDH_PT_WG_PER_RAY) {  // one workgroup per ray (device/kernels_walk.h, k_trace_wg): test access to wgTraceWhole
        const unsigned g = (unsigned)(n < 4096 ? n : 4096);
        const bool cnt = (flags & RDH_PT_COUNT) != 0;
        timeBegin(c);
        if (any && cnt) hipLaunchKernelGGL((k_trace_wg<true, true>), dim3(g), dim3(kWgTraceThreads), 0, c->stream, c->ds, d_rays, (long long)n, d_hits, d_occ);
        else if (any) hipLaunchKernelGGL((k_trace_wg<false, true>), dim3(g), dim3(kWgTraceThreads), 0, c->stream, c->ds, d_rays, (long long)n, d_hits, d_occ);
        else if (cnt) hipLaunchKernelGGL((k_trace_wg<true, false>), dim3(g), dim3(kWgTraceThreads), 0, c->stream, c->ds, d_rays, (long long)n, d_hits, d_occ);
        else hipLaunchKernelGGL((k_trace_wg<false, false>), dim3(g), dim3(kWgTraceThreads), 0, c->stream, c->ds, d_rays, (long long)n, d_hits, d_occ);
        return timeEnd(c, what);
    }
    timeBegin(c);
    int rc = launchWalk(c, d_rays, n, d_hits, d_occ, (flags & RDH_PT_COUNT) != 0, nullptr, nullptr, 0, (flags & RDH_PT_PAIRS) || usePairs(c, flags));
    if (rc) return rc;
    return timeEnd(c, what);
}
}

int rdh_trace_closest(rdh_ctx *c, const float *d_rays, int64_t n, rdh_hit *d_hits, uint32_t flags) {
    if (!c) return RDH_ERR_ARGS;
    if (!c->haveScene) return fail(c, RDH_ERR_NO_SCENE, "no scene uploaded");
    if (n < 0 || (n > 0 && (!d_rays || !d_hits))) return fail(c, RDH_ERR_ARGS, "rdh_trace_closest: bad arguments");
    if (n == 0) return RDH_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (flags & (RDH_PT_PERSISTENT | RDH_PT_WG_PER_RAY)) return walkPersistent(c, d_rays, n, (int4 *)d_hits, nullptr, flags, "trace_closest (persistent)");
    unsigned grid = (unsigned)((n + 255) / 256);
    timeBegin(c);
    if (flags & RDH_PT_COUNT)
        hipLaunchKernelGGL(k_trace_closest<true>, dim3(grid), dim3(256), 0, c->stream, c->ds, d_rays, (long long)n, (int4 *)d_hits);
    else
        hipLaunchKernelGGL(k_trace_closest<false>, dim3(grid), dim3(256), 0, c->stream, c->ds, d_rays, (long long)n, (int4 *)d_hits);
    return timeEnd(c, "trace_closest");
}

int rdh_trace_occluded(rdh_ctx *c, const float *d_seg, int64_t n, int32_t *d_occ, uint32_t flags) {
    if (!c) return RDH_ERR_ARGS;
    if (!c->haveScene) return fail(c, RDH_ERR_NO_SCENE, "no scene uploaded");
    if (n < 0 || (n > 0 && (!d_seg || !d_occ))) return fail(c, RDH_ERR_ARGS, "rdh_trace_occluded: bad arguments");
    if (n == 0) return RDH_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (flags & (RDH_PT_PERSISTENT | RDH_PT_WG_PER_RAY)) return walkPersistent(c, d_seg, n, nullptr, d_occ, flags, "trace_occluded (persistent)");
    unsigned grid = (unsigned)((n + 255) / 256);
    timeBegin(c);
    if (flags & RDH_PT_COUNT)
        hipLaunchKernelGGL(k_trace_occluded<true>, dim3(grid), dim3(256), 0, c->stream, c->ds, d_seg, (long long)n, d_occ);
    else
        hipLaunchKernelGGL(k_trace_occluded<false>, dim3(grid), dim3(256), 0, c->stream, c->ds, d_seg, (long long)n, d_occ);
    return timeEnd(c, "trace_occluded");
}

int rdh_dump_rays(rdh_ctx *c, int looper, int maxDepth, float *d_closest, int64_t capClosest, float *d_any, int64_t capAny,
                  int64_t *nClosest, int64_t *nAny) {
    int rc = requireReady(c);
    if (rc) return rc;
    if (!nClosest || !nAny || capClosest < 0 || capAny < 0 || (capClosest > 0 && !d_closest) || (capAny > 0 && !d_any) || looper < 0 ||
        looper >= 10000 || maxDepth < 0 || 4 + 7 * maxDepth > 200)
        return fail(c, RDH_ERR_ARGS, "rdh_dump_rays: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    PixelMap pm = makePixelMap(c);
    // scratch images (the frame's pixels are not wanted) + the two list counters
    float *img = nullptr;
    unsigned long long *cnt = nullptr;
    const size_t px = (size_t)(pm.packed ? (long long)pm.tilesPerRank * pm.tile * pm.tile : (long long)pm.W * pm.H);
    HIP_TRY(c, hipMalloc((void **)&img, px * 6 * sizeof(float)));
    if (hipMalloc((void **)&cnt, 16) != hipSuccess) {
        hipFree(img);
        return fail(c, RDH_ERR_ARGS, "rdh_dump_rays: out of device memory");
    }
    hipMemsetAsync(cnt, 0, 16, c->stream);
    hipMemsetAsync(img, 0, px * 6 * sizeof(float), c->stream);
    RayDump d{d_closest, d_any, (long long)capClosest, (long long)capAny, cnt};
    hipLaunchKernelGGL((k_path_trace_mega<false, true>), dim3(gridBlocks(pm)), dim3(64), 0, c->stream, c->ds, c->cam, pm, looper, 0,
                       maxDepth, img, img + px * 3, d);
    unsigned long long h[2] = {0, 0};
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
    hipFree(img);
    hipFree(cnt);
    if (e != hipSuccess) return fail(c, (int)e, "HIP error: rdh_dump_rays: %s", hipGetErrorString(e));
    *nClosest = (int64_t)h[0];
    *nAny = (int64_t)h[1];
    return RDH_OK;
}

int rdh_set_occupancy_share(rdh_ctx *c, int share) {
    if (!c || share < 1 || share > 64) return c ? fail(c, RDH_ERR_ARGS, "rdh_set_occupancy_share(%d)", share) : RDH_ERR_ARGS;
    if (share != c->share) c->orderValid = false;
    c->share = share;
    return RDH_OK;
}

int rdh_counters_reset(rdh_ctx *c) {
    if (!c) return RDH_ERR_ARGS;
    HIP_TRY(c, hipMemsetAsync(c->dCounters, 0, sizeof(Counters), c->stream));
    return RDH_OK;
}

int rdh_counters_read(rdh_ctx *c, rdh_counters *out) {
    if (!c || !out) return RDH_ERR_ARGS;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    Counters h;
    HIP_TRY(c, hipMemcpy(&h, c->dCounters, sizeof(h), hipMemcpyDeviceToHost));
    out->closestRays = h.closestRays;
    out->anyRays = h.anyRays;
    out->nodeVisits = h.nodeVisits;
    out->triTests = h.triTests;
    out->closestHits = h.closestHits;
    return RDH_OK;
}

int rdh_profile_reset(rdh_ctx *c) {
    if (!c) return RDH_ERR_ARGS;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->profUsed = 0;
    return RDH_OK;
}

int rdh_profile_read(rdh_ctx *c, double *totalMs, int64_t *launches) {
    if (!c || !totalMs || !launches) return RDH_ERR_ARGS;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double sum = 0.0;
    for (size_t i = 0; i < c->profUsed; i++) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->profEvents[2 * i], c->profEvents[2 * i + 1]));
        sum += ms;
    }
    *totalMs = sum;
    *launches = (int64_t)c->profUsed;
    return RDH_OK;
}

int rdh_debug_persist_stamps(rdh_ctx *c, uint64_t *out5x4096) {
    if (!c || !out5x4096) return RDH_ERR_ARGS;
#ifdef RD_PERSIST_STAMPS
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out5x4096, c->dPersist->stamp, sizeof(unsigned long long) * 5 * 4096, hipMemcpyDeviceToHost));
    return RDH_OK;
#else
    return fail(c, RDH_ERR_UNSUPPORTED, "library built without RD_PERSIST_STAMPS");
#endif
}

int rdh_copy_image_to_pbo(rdh_ctx *c, void *d_pbo, const void *d_image, int width, int height, int kind, int toneMapping,
                          float scale) {
    if (!c) return RDH_ERR_ARGS;
    if (!d_pbo || !d_image || width <= 0 || height <= 0 || kind < 0 || kind > 3 || toneMapping < 0 || toneMapping > 2)
        return fail(c, RDH_ERR_ARGS, "rdh_copy_image_to_pbo: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rj = joinComm(c)) return rj;  // d_image may be a G-buffer plane that an exchange is still completing
    long long total = (long long)width * height;
    hipLaunchKernelGGL(k_send_image_to_pbo, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                       static_cast<uint32_t *>(d_pbo), d_image, width, height, kind, toneMapping, scale);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

// ---- denoisers (src/denoiser.cu) -----------------------------------------------------------------------------------------
extern "C++" {
namespace {
int denoiseGB(rdh_ctx *c, const rdh_gbuffer *gb, DenoiseGB &d, const char *what) {
    if (int rj = joinComm(c)) return rj;  // the planes may still be completed by an exchange on the communication stream
    if (!gb || !gb->albedo || !gb->motion || gb->frameIdx < 0 || gb->frameIdx > 1 || gb->width <= 0 || gb->height <= 0)
        return fail(c, RDH_ERR_ARGS, "%s: bad G-buffer", what);
    for (int k = 0; k < 2; k++)
        if (!gb->normal[k] || !gb->depth[k] || !gb->primId[k]) return fail(c, RDH_ERR_ARGS, "%s: null G-buffer plane", what);
    const int f = gb->frameIdx;
    d = DenoiseGB{gb->albedo, gb->normal[f], gb->normal[f ^ 1], gb->depth[f], nullptr, gb->motion, gb->primId[f], gb->primId[f ^ 1],
                  gb->width, gb->height};
    return RDH_OK;
}
dim3 denoiseGrid(int w, int h) { return dim3((unsigned)((w + 31) / 32), (unsigned)((h + 7) / 8)); }
// world-space position of every pixel for the filter launched next (recomputed per launch: the depth plane behind the
// pointer changes from frame to frame, and 10 us buy 40 % of the filter's arithmetic)
int denoisePositions(rdh_ctx *c, DenoiseGB &d, const DCamera &cam) {
    const long long n = (long long)cam.resx * cam.resy;
    if (n > c->posPlanePixels) {
        if (c->posPlane) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            hipFree(c->posPlane);
            c->posPlane = nullptr;
            c->posPlanePixels = 0;
        }
        HIP_TRY(c, hipMalloc((void **)&c->posPlane, sizeof(float) * 3 * (size_t)n));
        c->posPlanePixels = n;
    }
    hipLaunchKernelGGL(k_position_plane, denoiseGrid(cam.resx, cam.resy), dim3(256), 0, c->stream, c->posPlane, d.depth, cam);
    d.position = c->posPlane;
    return RDH_OK;
}
}  // namespace
}  // extern "C++"

int rdh_denoise_eaw(rdh_ctx *c, float *d_colorOut, const float *d_colorIn, const rdh_gbuffer *gb, const void *camera196,
                    float sigLumin, float sigNormal, float sigDepth, int level) {
    if (!c || !d_colorOut || !d_colorIn || !camera196 || level < 0 || level > 16) return c ? fail(c, RDH_ERR_ARGS, "rdh_denoise_eaw: bad arguments") : RDH_ERR_ARGS;
    DenoiseGB d;
    int rc = denoiseGB(c, gb, d, "rdh_denoise_eaw");
    if (rc) return rc;
    DCamera cam = toDeviceCamera(camera196);
    if (cam.resx != gb->width || cam.resy != gb->height) return fail(c, RDH_ERR_ARGS, "rdh_denoise_eaw: camera / G-buffer size mismatch");
    HIP_TRY(c, hipSetDevice(c->device));
    rc = denoisePositions(c, d, cam);
    if (rc) return rc;
    {
        const dim3 grid = denoiseTileGrid(cam.resx, cam.resy, level);
        const int v = (sigmaIsPow2(sigLumin) ? 4 : 0) | (sigmaIsPow2(sigNormal) ? 2 : 0) | (sigmaIsPow2(sigDepth) ? 1 : 0);
#define RD_EAW(L, N, D) hipLaunchKernelGGL((k_eaw_filter<L, N, D>), grid, dim3(256), 0, c->stream, d_colorOut, d_colorIn, d, sigDepth, sigNormal, sigLumin, cam, level)
        switch (v) {
            case 0: RD_EAW(false, false, false); break;
            case 1: RD_EAW(false, false, true); break;
            case 2: RD_EAW(false, true, false); break;
            case 3: RD_EAW(false, true, true); break;
            case 4: RD_EAW(true, false, false); break;
            case 5: RD_EAW(true, false, true); break;
            case 6: RD_EAW(true, true, false); break;
            default: RD_EAW(true, true, true); break;
        }
#undef RD_EAW
    }
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_denoise_svgf(rdh_ctx *c, float *d_colorOut, const float *d_colorIn, float *d_varianceOut, const float *d_varianceIn,
                     const float *d_filteredVar, const rdh_gbuffer *gb, const void *camera196, float sigLumin, float sigNormal,
                     float sigDepth, int level) {
    if (!c || !d_colorOut || !d_colorIn || !d_varianceOut || !d_varianceIn || !d_filteredVar || !camera196 || level < 0 || level > 16)
        return c ? fail(c, RDH_ERR_ARGS, "rdh_denoise_svgf: bad arguments") : RDH_ERR_ARGS;
    DenoiseGB d;
    int rc = denoiseGB(c, gb, d, "rdh_denoise_svgf");
    if (rc) return rc;
    DCamera cam = toDeviceCamera(camera196);
    if (cam.resx != gb->width || cam.resy != gb->height) return fail(c, RDH_ERR_ARGS, "rdh_denoise_svgf: camera / G-buffer size mismatch");
    HIP_TRY(c, hipSetDevice(c->device));
    rc = denoisePositions(c, d, cam);
    if (rc) return rc;
    if (sigmaIsPow2(sigDepth + 1e-4f))
        hipLaunchKernelGGL((k_svgf_filter<true>), denoiseTileGrid(cam.resx, cam.resy, level), dim3(256), 0, c->stream, d_colorOut, d_colorIn,
                           d_varianceOut, d_varianceIn, d_filteredVar, d, sigDepth, sigNormal, sigLumin, cam, level);
    else
        hipLaunchKernelGGL((k_svgf_filter<false>), denoiseTileGrid(cam.resx, cam.resy, level), dim3(256), 0, c->stream, d_colorOut, d_colorIn,
                           d_varianceOut, d_varianceIn, d_filteredVar, d, sigDepth, sigNormal, sigLumin, cam, level);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_denoise_modulate(rdh_ctx *c, float *d_image, const rdh_gbuffer *gb) {
    if (!c || !d_image) return c ? fail(c, RDH_ERR_ARGS, "rdh_denoise_modulate: bad arguments") : RDH_ERR_ARGS;
    DenoiseGB d;
    int rc = denoiseGB(c, gb, d, "rdh_denoise_modulate");
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    long long n = (long long)gb->width * gb->height;
    hipLaunchKernelGGL(k_modulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_image, d.albedo, n);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_denoise_add(rdh_ctx *c, float *d_out, const float *d_in1, const float *d_in2, int width, int height) {
    if (!c || !d_out || !d_in1 || !d_in2 || width <= 0 || height <= 0) return c ? fail(c, RDH_ERR_ARGS, "rdh_denoise_add: bad arguments") : RDH_ERR_ARGS;
    HIP_TRY(c, hipSetDevice(c->device));
    long long n3 = 3ll * width * height;
    hipLaunchKernelGGL(k_add_images, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, c->stream, d_out, d_in1, d_in2, n3);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_denoise_temporal_accumulate(rdh_ctx *c, float *d_colorAccumOut, const float *d_colorAccumIn, float *d_momentAccumOut,
                                    const float *d_momentAccumIn, const float *d_colorIn, const rdh_gbuffer *gb, int first) {
    if (!c || !d_colorAccumOut || !d_colorAccumIn || !d_momentAccumOut || !d_momentAccumIn || !d_colorIn)
        return c ? fail(c, RDH_ERR_ARGS, "rdh_denoise_temporal_accumulate: bad arguments") : RDH_ERR_ARGS;
    DenoiseGB d;
    int rc = denoiseGB(c, gb, d, "rdh_denoise_temporal_accumulate");
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_temporal_accumulate, denoiseGrid(gb->width, gb->height), dim3(256), 0, c->stream, d_colorAccumOut,
                       d_colorAccumIn, d_momentAccumOut, d_momentAccumIn, d_colorIn, d, first);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_denoise_estimate_variance(rdh_ctx *c, float *d_variance, const float *d_moment, int width, int height) {
    if (!c || !d_variance || !d_moment || width <= 0 || height <= 0) return c ? fail(c, RDH_ERR_ARGS, "rdh_denoise_estimate_variance: bad arguments") : RDH_ERR_ARGS;
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_estimate_variance, denoiseGrid(width, height), dim3(256), 0, c->stream, d_variance, d_moment, width, height);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_denoise_filter_variance(rdh_ctx *c, float *d_varianceOut, const float *d_varianceIn, int width, int height) {
    if (!c || !d_varianceOut || !d_varianceIn || width <= 0 || height <= 0) return c ? fail(c, RDH_ERR_ARGS, "rdh_denoise_filter_variance: bad arguments") : RDH_ERR_ARGS;
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_filter_variance, denoiseGrid(width, height), dim3(256), 0, c->stream, d_varianceOut, d_varianceIn, width, height);
    HIP_TRY(c, hipGetLastError());
    return RDH_OK;
}

int rdh_debug_persist_phases(rdh_ctx *c, uint64_t *out16) {
    if (!c || !out16) return RDH_ERR_ARGS;
#ifdef RD_PERSIST_PHASES
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out16, c->dPersist->phase, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost));
    return RDH_OK;
#else
    return fail(c, RDH_ERR_UNSUPPORTED, "library built without RD_PERSIST_PHASES");
#endif
}

int rdh_last_kernel_ms(rdh_ctx *c, float *ms) {
    if (!c || !ms) return RDH_ERR_ARGS;
    if (!c->timed) return fail(c, RDH_ERR_STATE, "no timed launch yet");
    HIP_TRY(c, hipEventSynchronize(c->evStop));
    HIP_TRY(c, hipEventElapsedTime(ms, c->evStart, c->evStop));
    return RDH_OK;
}

}  // extern "C"
