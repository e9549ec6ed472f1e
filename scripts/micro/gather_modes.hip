// scripts/micro/gather_modes.hip — how does the vector L1 (TCP) price a divergent gather?  Per lane, per unique line,
// or per byte?  All kernels: 12 waves per CU (3 per SIMD), dependent chains, records of 32 B in a table of `recs`.
//  mode 0: every lane its own record, two dwordx4 loads               (what k_pt_persistent's box loop does)
//  mode 1: every lane its own record, one dwordx4 load
//  mode 2: lane PAIRS share a record: even lane loads .a, odd lane .b — one instruction, 32 records per wave
//  mode 3: every lane its own record, two dwordx2 loads (first 8 B of each half)
//  mode 4: every lane its own record, one dword load
//  mode 5: lane QUADS share a 64-B record pair: lane q loads 16 B at offset 16q — one instruction, 16 chains per wave
//  mode 6: every lane its own record, four dwordx4 loads from a 64-B-aligned pair of records (same 128-B line)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Rec { float4 a, b; };

template <int MODE>
__global__ __launch_bounds__(64) void chase(const Rec *__restrict__ tab, unsigned mask, int steps, unsigned *out) {
    const unsigned lane = threadIdx.x;
    unsigned chain = blockIdx.x * 64u + lane;
    if (MODE == 2) chain = blockIdx.x * 64u + (lane >> 1);
    if (MODE == 5) chain = blockIdx.x * 64u + (lane >> 2);
    unsigned idx = chain * 2654435761u & mask;
    unsigned acc = 0;
    const float4 *t4 = (const float4 *)tab;
    for (int s = 0; s < steps; s++) {
        unsigned nxt;
        if (MODE == 0) {
            float4 a = tab[idx].a, b = tab[idx].b;
            nxt = __float_as_uint(a.w) + __float_as_uint(b.w);
        } else if (MODE == 1) {
            float4 a = tab[idx].a;
            nxt = __float_as_uint(a.w);
        } else if (MODE == 2) {
            float4 v = t4[idx * 2u + (lane & 1u)];
            unsigned w = __float_as_uint(v.w);
            unsigned o = __shfl_xor((int)w, 1, 64);
            nxt = w + o;
        } else if (MODE == 3) {
            const float2 *t2 = (const float2 *)tab;
            float2 a = t2[idx * 4u], b = t2[idx * 4u + 2u];
            nxt = __float_as_uint(a.y) * 2654435761u + __float_as_uint(b.y);
        } else if (MODE == 4) {
            const float *t1 = (const float *)tab;
            nxt = __float_as_uint(t1[idx * 8u + 3u]);
        } else if (MODE == 5) {
            unsigned base = (idx & ~1u) * 2u;
            float4 v = t4[base + (lane & 3u)];
            unsigned w = __float_as_uint(v.w);
            w += __shfl_xor((int)w, 1, 64);
            w += __shfl_xor((int)w, 2, 64);
            nxt = w;
        } else {
            unsigned base = (idx & ~1u) * 2u;
            float4 a = t4[base], b = t4[base + 1], c = t4[base + 2], d = t4[base + 3];
            nxt = __float_as_uint(a.w) + __float_as_uint(b.w) + __float_as_uint(c.w) + __float_as_uint(d.w);
        }
        acc += nxt;
        idx = (nxt + chain * 0x9E3779B9u + (unsigned)s * 0x85EBCA6Bu) & mask;  // depends on the loaded word; chains never merge
    }
    if (acc == 0x12345u) out[0] = idx;
}

// LDS-resident table shared by a whole workgroup (one workgroup per CU)
template <int LOADS>
__global__ __launch_bounds__(1024) void chaseLds(const Rec *__restrict__ tab, unsigned ldsRecs, int steps, unsigned *out) {
    extern __shared__ Rec lds[];
    for (unsigned i = threadIdx.x; i < ldsRecs; i += blockDim.x) lds[i] = tab[i];
    __syncthreads();
    const unsigned mask = ldsRecs - 1;
    unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
    unsigned acc = 0;
    for (int s = 0; s < steps; s++) {
        float4 a = lds[idx].a;
        unsigned nxt = __float_as_uint(a.w);
        if (LOADS == 2) {
            float4 b = lds[idx].b;
            nxt += __float_as_uint(b.w);
        }
        acc += nxt;
        idx = (nxt + (blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B9u + (unsigned)s * 0x85EBCA6Bu) & mask;
    }
    if (acc == 0x12345u) out[0] = idx;
}

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int CUs = p.multiProcessorCount;
    const size_t maxRecs = 1u << 21;  // 64 MB
    std::vector<Rec> h(maxRecs);
    uint32_t st = 12345;
    for (size_t i = 0; i < maxRecs; i++) {
        uint32_t r[8];
        for (int k = 0; k < 8; k++) { st = st * 1664525u + 1013904223u; r[k] = st >> 3; }
        memcpy(&h[i], r, 32);
    }
    Rec *d;
    unsigned *out;
    CK(hipMalloc(&d, maxRecs * sizeof(Rec)));
    CK(hipMalloc(&out, 16));
    CK(hipMemcpy(d, h.data(), maxRecs * sizeof(Rec), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int steps = 2000;
    const char *names[] = {"0: 2 x dwordx4 / lane", "1: 1 x dwordx4 / lane", "2: lane pairs share 32 B", "3: 2 x dwordx2 / lane",
                           "4: 1 x dword / lane", "5: lane quads share 64 B", "6: 4 x dwordx4 / lane (64 B)"};
    const int chainsPerWave[] = {64, 64, 32, 64, 64, 16, 64};
    auto run = [&](int mode, auto kern, unsigned recs, int wavesPerSimd) {
        int grid = CUs * 4 * wavesPerSimd;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, recs - 1, 100, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, recs - 1, steps, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double waveSteps = (double)grid * steps;
        printf("%-30s table %9.3f MB w/SIMD %d: %7.3f ms %7.1f ns/wave-step/CU  lat/step %7.1f ns  %8.1f G chain-steps/s\n", names[mode],
               recs * 32.0 / 1e6, wavesPerSimd, ms, ms * 1e6 / (waveSteps / CUs), ms * 1e6 / steps,
               waveSteps * chainsPerWave[mode] / (ms * 1e-3) / 1e9);
    };
    unsigned sizes[] = {256, 512, 1024, 4096, 1u << 15, 1u << 18, 1u << 21};  // 8 KB, 16 KB, 32 KB, 128 KB, 1 MB, 8 MB, 64 MB
    for (unsigned recs : sizes) {
        for (int w : {3, 6}) {
            run(0, chase<0>, recs, w);
            run(1, chase<1>, recs, w);
            run(2, chase<2>, recs, w);
            run(3, chase<3>, recs, w);
            run(4, chase<4>, recs, w);
            run(5, chase<5>, recs, w);
            run(6, chase<6>, recs, w);
        }
        printf("\n");
    }
    auto runLds = [&](const char *name, auto kern, unsigned ldsRecs, int wavesPerCU) {
        int grid = CUs;
        size_t sh = (size_t)ldsRecs * 32;
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wavesPerCU), sh, 0, d, ldsRecs, 100, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wavesPerCU), sh, 0, d, ldsRecs, steps, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double waveSteps = (double)grid * wavesPerCU * steps;
        printf("%-30s LDS %4u KB, waves/CU %2d : %7.3f ms  %7.1f ns/wave-step/CU  lat/step %7.1f ns  %8.1f G chain-steps/s\n", name,
               (unsigned)(sh >> 10), wavesPerCU, ms, ms * 1e6 / (waveSteps / CUs), ms * 1e6 / steps, waveSteps * 64 / (ms * 1e-3) / 1e9);
    };
    for (unsigned recs : {2048u, 4096u})
        for (int w : {4, 8, 12, 16}) {
            runLds("LDS 2 x b128 / lane", chaseLds<2>, recs, w);
            runLds("LDS 1 x b128 / lane", chaseLds<1>, recs, w);
        }
    return 0;
}
