// radish_pt_amd/csrc/host/scene_build.cpp — host-side producers of the DevScene arrays (include/radish_host.h).
//
// CPU-only C++ (g++).  Restates, in this project's own structure, what the reference computes on the host
// before DevScene::create uploads it: the binned-SAH builder and its six threaded orderings
// (/root/reference/src/bvh.cpp), the alias-table construction (/root/reference/src/sampler.h:81-125),
// the emissive-triangle list (/root/reference/src/scene.cpp:192-223) and Camera::update
// (/root/reference/src/sceneStructs.h:93-107).  Built with -ffp-contract=off so the float results do not
// depend on whether the host has FMA.
#include "../../../include/radish_host.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct F3 {
    float x, y, z;
    float at(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline F3 f3min(F3 a, F3 b) { return {b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }
inline F3 f3max(F3 a, F3 b) { return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z}; }
inline F3 sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 crossf(F3 a, F3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float dotf(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline F3 normalizef(F3 v) {
    float s = 1.f / sqrtf(dotf(v, v));
    return {v.x * s, v.y * s, v.z * s};
}

struct Box {  // src/bvh.h:15-159 — starts empty (pMin = +FLT_MAX, pMax = -FLT_MAX)
    F3 lo{FLT_MAX, FLT_MAX, FLT_MAX}, hi{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(F3 p) { lo = f3min(lo, p); hi = f3max(hi, p); }
    void grow(const Box &b) { lo = f3min(lo, b.lo); hi = f3max(hi, b.hi); }
    F3 center() const { return {(lo.x + hi.x) * .5f, (lo.y + hi.y) * .5f, (lo.z + hi.z) * .5f}; }
    float surfaceArea() const {  // bvh.h:52-55
        F3 s = sub(hi, lo);
        return 2.f * (s.x * s.y + s.y * s.z + s.z * s.x);
    }
    int longestAxis() const {  // bvh.h:60-67
        F3 s = sub(hi, lo);
        if (s.x < s.y) return s.y > s.z ? 1 : 2;
        return s.x > s.z ? 0 : 2;
    }
};

struct PrimRef {
    int primId;
    Box bound;
    F3 center;
};

struct SubtreeInfo {  // per depth-first slot: leaf -> primitive id, inner -> number of slots in the subtree
    bool leaf;
    int primOrSize;
};

constexpr int kBuckets = 16;  // bvh.cpp:37

// Bucket of a centroid coordinate (bvh.cpp:82-84).  A zero-width centroid range gives 0/0 = NaN, whose
// float→int conversion is undefined in C++; x86 returns INT_MIN, which the reference's clamp turns into 0.
inline int bucketOf(float c, float lo, float hi) {
    float f = (c - lo) / (hi - lo) * kBuckets;
    if (f != f) return 0;
    if (f >= 2147483648.f) return 0;   // cvttss2si overflow → INT_MIN → clamp → 0
    if (f <= -2147483648.f) return 0;
    int b = (int)f;
    return b < 0 ? 0 : (b > kBuckets - 1 ? kBuckets - 1 : b);
}

}  // namespace

extern "C" int32_t rdh_build_bvh(const float *vertices, int32_t numPrims, float *boxesOut,
                                 int32_t *const nodesOut[6]) {
    if (!vertices || numPrims <= 0 || !boxesOut || !nodesOut) return RDH_HOST_ERR_ARGS;
    const int bvhSize = numPrims * 2 - 1;
    const F3 *v = reinterpret_cast<const F3 *>(vertices);

    std::vector<PrimRef> prims(numPrims);
    for (int i = 0; i < numPrims; i++) {
        prims[i].primId = i;
        Box b;  // AABB(va, vb, vc): min/max of the three vertices (bvh.h:22-25)
        b.lo = f3min(f3min(v[i * 3], v[i * 3 + 1]), v[i * 3 + 2]);
        b.hi = f3max(f3max(v[i * 3], v[i * 3 + 1]), v[i * 3 + 2]);
        prims[i].bound = b;
        prims[i].center = b.center();
    }
    std::vector<SubtreeInfo> info(bvhSize);
    std::vector<Box> boxes(bvhSize);

    // Depth-first flattened top-down build (bvh.cpp:32-129): the left subtree occupies the slots right
    // after its parent, the right subtree follows it.
    struct Job { int slot, first, last; };
    std::vector<Job> todo;
    todo.reserve(64);
    todo.push_back({0, 0, numPrims - 1});
    std::vector<PrimRef> scratch;
    while (!todo.empty()) {
        Job job = todo.back();
        todo.pop_back();
        const int count = job.last - job.first + 1;
        const int slots = count * 2 - 1;
        const bool leaf = (slots == 1);
        info[job.slot] = {leaf, leaf ? prims[job.first].primId : slots};

        Box nodeBound, centerBound;
        for (int i = job.first; i <= job.last; i++) {
            nodeBound.grow(prims[i].bound);
            centerBound.grow(prims[i].center);
        }
        boxes[job.slot] = nodeBound;
        if (leaf) continue;

        const int axis = centerBound.longestAxis();
        // bvh.cpp:64-72 (`if (nodeSize == 2)`) is unreachable: nodeSize = 2*count-1 is always odd, so a
        // two-primitive node (nodeSize 3) goes through the bucket code like every other node.

        Box bucketBound[kBuckets];
        int bucketCount[kBuckets] = {0};
        const float lo = centerBound.lo.at(axis), hi = centerBound.hi.at(axis);
        for (int i = job.first; i <= job.last; i++) {
            int b = bucketOf(prims[i].center.at(axis), lo, hi);
            bucketBound[b].grow(prims[i].bound);
            bucketCount[b]++;
        }
        // bvh.cpp:93-102.  NB the reference does NOT accumulate prefix/suffix unions: it merges an EMPTY box
        // with the neighbouring bucket, so left[i] = bucket[i-1] (left[0] = bucket[0]) and
        // right[j] = bucket[j+1] (right[15] = bucket[15]).  Reproduced as is: the tree shape — and with it
        // the visit counts the roofline accounting uses — depends on it.
        Box left[kBuckets], right[kBuckets];
        int prefix[kBuckets];
        left[0] = bucketBound[0];
        right[kBuckets - 1] = bucketBound[kBuckets - 1];
        prefix[0] = bucketCount[0];
        for (int i = 1, j = kBuckets - 2; i < kBuckets; i++, j--) {
            left[i].grow(bucketBound[i - 1]);
            right[j].grow(bucketBound[j + 1]);
            prefix[i] = prefix[i - 1] + bucketCount[i];
        }
        float best = FLT_MAX;
        int splitBucket = 0;
        for (int i = 0; i < kBuckets - 1; i++) {
            // glm::mix(SA_left, SA_right, n_left / n) (bvh.cpp:104-106) = x*(1-a) + y*a
            float a = float(prefix[i]) / count;
            float cost = left[i].surfaceArea() * (1.f - a) + right[i + 1].surfaceArea() * a;
            if (cost < best) {
                best = cost;
                splitBucket = i;
            }
        }
        // Partition: left part fills forward, right part fills backward from the end (bvh.cpp:115-124).
        scratch.assign(prims.begin() + job.first, prims.begin() + job.last + 1);
        int fwd = job.first, bwd = job.last;
        for (int i = 0; i < count; i++) {
            int b = bucketOf(scratch[i].center.at(axis), lo, hi);
            if (b <= splitBucket) prims[fwd++] = scratch[i];
            else prims[bwd--] = scratch[i];
        }
        int mid = std::min(std::max(fwd - 1, job.first), job.last - 1);
        int leftSlots = 2 * (mid - job.first + 1) - 1;
        todo.push_back({job.slot + 1 + leftSlots, mid + 1, job.last});
        todo.push_back({job.slot + 1, job.first, mid});
    }

    for (int i = 0; i < bvhSize; i++) {
        boxesOut[i * 6 + 0] = boxes[i].lo.x; boxesOut[i * 6 + 1] = boxes[i].lo.y; boxesOut[i * 6 + 2] = boxes[i].lo.z;
        boxesOut[i * 6 + 3] = boxes[i].hi.x; boxesOut[i * 6 + 4] = boxes[i].hi.y; boxesOut[i * 6 + 5] = boxes[i].hi.z;
    }

    // Six threaded orderings (bvh.cpp:136-183): array k serves rays whose dominant travel axis is k/2;
    // of the two children the one nearer along that axis is emitted first.
    std::vector<int> stack;
    stack.reserve(256);
    for (int k = 0; k < 6; k++) {
        int32_t *out = nodesOut[k];
        if (!out) return RDH_HOST_ERR_ARGS;
        const int dim = k / 2;
        const bool lesser = (k & 1) != 0;
        stack.clear();
        stack.push_back(0);
        int emitted = 0;
        while (!stack.empty()) {
            int slot = stack.back();
            stack.pop_back();
            const bool leaf = info[slot].leaf;
            const int slots = leaf ? 1 : info[slot].primOrSize;
            out[emitted * 3 + 0] = leaf ? info[slot].primOrSize : -1;
            out[emitted * 3 + 1] = slot;
            out[emitted * 3 + 2] = emitted + slots;
            emitted++;
            if (leaf) continue;
            int a = slot + 1;
            int b = slot + 1 + (info[a].leaf ? 1 : info[a].primOrSize);
            if ((boxes[a].center().at(dim) < boxes[b].center().at(dim)) != lesser) std::swap(a, b);
            stack.push_back(b);
            stack.push_back(a);
        }
    }
    return bvhSize;
}

extern "C" int32_t rdh_build_alias_table(const float *valuesIn, int32_t n, void *tableOut, float *sumOut) {
    if (n < 0 || (n > 0 && (!valuesIn || !tableOut))) return RDH_HOST_ERR_ARGS;
    struct Entry { float prob; int32_t failId; };
    Entry *table = static_cast<Entry *>(tableOut);
    std::vector<float> val(valuesIn, valuesIn + n);
    float sum = 0.f;
    for (float x : val) sum += x;
    if (sumOut) *sumOut = sum;
    float scale = static_cast<float>(n) / sum;
    for (float &x : val) x *= scale;

    // Two LIFO piles (sampler.h:93-117): entries above the mean donate to entries at or below it.
    std::vector<Entry> over, under;
    over.reserve(2 * n);
    under.reserve(2 * n);
    for (int i = 0; i < n; i++) (val[i] > 1.f ? over : under).push_back({val[i], i});
    while (!over.empty() && !under.empty()) {
        Entry big = over.back(); over.pop_back();
        Entry small = under.back(); under.pop_back();
        table[small.failId] = {small.prob, big.failId};
        big.prob -= (1.f - small.prob);
        (big.prob > 1.f ? over : under).push_back(big);
    }
    for (int i = (int)over.size() - 1; i >= 0; --i) table[over[i].failId] = over[i];
    for (int i = (int)under.size() - 1; i >= 0; --i) table[under[i].failId] = under[i];
    return 0;
}

extern "C" int32_t rdh_build_light_list(const float *vertices, const int32_t *materialIds, int32_t numPrims,
                                        const void *materials, int32_t numMaterials, int32_t *lightPrimIdsOut,
                                        float *lightUnitRadianceOut, float *lightPowerOut) {
    if (!vertices || !materialIds || !materials || numPrims < 0) return RDH_HOST_ERR_ARGS;
    struct Mat { int32_t type; float r, g, b; float metallic, roughness, ior; int32_t map[4]; };
    static_assert(sizeof(Mat) == 44, "Material layout");
    const Mat *m = static_cast<const Mat *>(materials);
    const F3 *v = reinterpret_cast<const F3 *>(vertices);
    const float pi = 3.14159265358979323846264338327950288f;
    int count = 0;
    for (int p = 0; p < numPrims; p++) {
        int id = materialIds[p];
        if (id < 0 || id >= numMaterials) return RDH_HOST_ERR_ARGS;
        if (m[id].type != 4) continue;  // Material::Type::Light
        float lum = 0.2126f * m[id].r + 0.7152f * m[id].g + 0.0722f * m[id].b;
        float powerUnitArea = lum * 2.f * pi;
        F3 c = crossf(sub(v[p * 3 + 1], v[p * 3]), sub(v[p * 3 + 2], v[p * 3]));
        float area = sqrtf(dotf(c, c)) * 0.5f;
        lightPrimIdsOut[count] = p;
        lightUnitRadianceOut[count * 3 + 0] = m[id].r;
        lightUnitRadianceOut[count * 3 + 1] = m[id].g;
        lightUnitRadianceOut[count * 3 + 2] = m[id].b;
        lightPowerOut[count] = powerUnitArea * area;
        count++;
    }
    return count;
}

extern "C" int32_t rdh_build_envmap_pdf(const float *texels, int32_t width, int32_t height, float *pdfOut) {
    if (!texels || !pdfOut || width <= 0 || height <= 0) return RDH_HOST_ERR_ARGS;
    const float PI = 3.1415926535897932384626422832795028841971f;
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            int idx = i * width + j;
            const float *c = texels + 3 * (size_t)idx;
            float lum = 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2];
            pdfOut[idx] = lum * sinf((.5f + i) / height * PI);
        }
    return 0;
}

extern "C" void rdh_camera_update(void *camera196) {
    struct Cam {
        int32_t resx, resy;
        F3 position, rotation, view, up, right;
        float fovx, fovy, pixLenX, pixLenY;
        float rotInv[9];
        float viewProj[16];
        float lensRadius, focalDist, tanFovY;
    };
    static_assert(sizeof(Cam) == 196, "Camera layout");
    Cam c;
    memcpy(&c, camera196, sizeof(c));
    const float PI = 3.1415926535897932384626422832795028841971f;
    const float rad = 0.01745329251994329576923690768489f;
    // Scene::loadCamera (scene.cpp:378-383)
    float yscaled = tanf(c.fovy * (PI / 180));
    float xscaled = (yscaled * c.resx) / c.resy;
    c.fovx = (atanf(xscaled) * 180) / PI;
    c.tanFovY = tanf((c.fovy * 0.5f) * rad);
    // Camera::update (sceneStructs.h:93-107)
    float yaw = c.rotation.x * rad, pitch = c.rotation.y * rad, roll = c.rotation.z * rad;
    F3 view{cosf(yaw) * cosf(pitch), sinf(pitch) * cosf(roll), sinf(yaw) * cosf(pitch)};
    view = normalizef(view);
    F3 right = normalizef(crossf(view, F3{0.f, 1.f, 0.f}));
    F3 up = normalizef(crossf(right, view));
    c.view = view;
    c.right = right;
    c.up = up;
    // glm::inverse(mat3(right, up, view)), column-major
    float m00 = right.x, m01 = right.y, m02 = right.z;
    float m10 = up.x, m11 = up.y, m12 = up.z;
    float m20 = view.x, m21 = view.y, m22 = view.z;
    float ood = 1.f / ((m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02)) + m20 * (m01 * m12 - m11 * m02));
    c.rotInv[0] = +(m11 * m22 - m21 * m12) * ood;  // [0][0]
    c.rotInv[1] = -(m01 * m22 - m21 * m02) * ood;  // [0][1]
    c.rotInv[2] = +(m01 * m12 - m11 * m02) * ood;  // [0][2]
    c.rotInv[3] = -(m10 * m22 - m20 * m12) * ood;  // [1][0]
    c.rotInv[4] = +(m00 * m22 - m20 * m02) * ood;  // [1][1]
    c.rotInv[5] = -(m00 * m12 - m10 * m02) * ood;  // [1][2]
    c.rotInv[6] = +(m10 * m21 - m20 * m11) * ood;  // [2][0]
    c.rotInv[7] = -(m00 * m21 - m20 * m01) * ood;  // [2][1]
    c.rotInv[8] = +(m00 * m11 - m10 * m01) * ood;  // [2][2]
    // viewProjection = perspective(radians(2*fovy), aspect, .01, 1000) * lookAt(pos, pos+view, up); it is not read
    // anywhere on the hot path (Camera::getRasterUV has that code commented out, sceneStructs.h:23-27).
    {
        F3 f = view, s = normalizef(crossf(f, up)), u = crossf(s, f);
        float V[16] = {s.x, u.x, -f.x, 0, s.y, u.y, -f.y, 0, s.z, u.z, -f.z, 0,
                       -dotf(s, c.position), -dotf(u, c.position), dotf(f, c.position), 1};
        float aspect = float(c.resx) / c.resy, zn = .01f, zf = 1000.f;
        float th = tanf((c.fovy * 2.f) * rad / 2.f);
        float P[16] = {0};
        P[0] = 1.f / (aspect * th);
        P[5] = 1.f / th;
        P[10] = -(zf + zn) / (zf - zn);
        P[11] = -1.f;
        P[14] = -(2.f * zf * zn) / (zf - zn);
        for (int col = 0; col < 4; col++)
            for (int row = 0; row < 4; row++) {
                float acc = 0.f;
                for (int k = 0; k < 4; k++) acc += P[k * 4 + row] * V[col * 4 + k];
                c.viewProj[col * 4 + row] = acc;
            }
    }
    memcpy(camera196, &c, sizeof(c));
}
