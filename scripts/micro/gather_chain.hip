// scripts/micro/gather_chain.hip — what does one dependent 32-byte-record gather step cost on MI355X?
// Each lane chases its own pseudo-random chain through a table of 32-byte records (the NodeRec shape), two dwordx4
// loads per step (or one), optionally followed by a block of dependent VALU work the size of the box test.
// Reports ns per wave-step per CU for several occupancies and table sizes: the ceiling k_pt_persistent's box loop runs under.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Rec { float4 a, b; };

template <int LOADS, int ALU>
__global__ __launch_bounds__(64) void chase(const Rec *__restrict__ tab, unsigned mask, int steps, unsigned *out) {
    unsigned idx = (blockIdx.x * 64u + threadIdx.x) * 2654435761u & mask;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        float4 a = tab[idx].a;
        float4 b = make_float4(0, 0, 0, 0);
        if (LOADS == 2) b = tab[idx].b;
        float x = a.x + b.y;
#pragma unroll
        for (int k = 0; k < ALU; k++) x = x * 1.0001f + a.y;   // dependent VALU chain
        acc += x;
        idx = (__float_as_uint(a.w) + (LOADS == 2 ? __float_as_uint(b.w) : 0u) + (x > 1e30f ? 1u : 0u)) & mask;
    }
    if (acc == 123.456f) out[0] = idx;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = idx;
}

template <int LOADS, int ALU>
__global__ __launch_bounds__(64) void chaseLds(const Rec *__restrict__ tab, unsigned mask, int steps, unsigned *out) {
    __shared__ Rec lds[2048];  // 64 KB
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = tab[i];
    __syncthreads();
    unsigned idx = (blockIdx.x * 64u + threadIdx.x) * 2654435761u & 2047u;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        float4 a = lds[idx].a;
        float4 b = make_float4(0, 0, 0, 0);
        if (LOADS == 2) b = lds[idx].b;
        float x = a.x + b.y;
#pragma unroll
        for (int k = 0; k < ALU; k++) x = x * 1.0001f + a.y;
        acc += x;
        idx = (__float_as_uint(a.w) + (LOADS == 2 ? __float_as_uint(b.w) : 0u) + (x > 1e30f ? 1u : 0u)) & 2047u;
    }
    if (acc == 123.456f) out[0] = idx;
}

int main() {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, dev));
    const int CUs = p.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", p.name, CUs, p.clockRate);
    const size_t maxRecs = 1u << 23;  // 256 MB
    std::vector<Rec> h(maxRecs);
    uint32_t st = 12345;
    for (size_t i = 0; i < maxRecs; i++) {
        st = st * 1664525u + 1013904223u;
        uint32_t r1 = st >> 4;
        st = st * 1664525u + 1013904223u;
        uint32_t r2 = st >> 4;
        h[i].a = make_float4(1.f, 0.5f, 0.f, 0.f);
        h[i].b = make_float4(0.f, 2.f, 0.f, 0.f);
        ((uint32_t *)&h[i].a)[3] = r1;
        ((uint32_t *)&h[i].b)[3] = r2;
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
    auto run = [&](const char *name, auto kern, unsigned recs, int wavesPerSimd) {
        int grid = CUs * 4 * wavesPerSimd;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, recs - 1, 200, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, recs - 1, steps, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double waveSteps = (double)grid * steps;
        double nsPerWaveStepPerCU = ms * 1e6 / (waveSteps / CUs);
        double glane = waveSteps * 64 / (ms * 1e-3) / 1e9;
        printf("%-28s table %8.2f MB  waves/SIMD %d : %7.3f ms  %7.1f ns/wave-step/CU  lat/step %7.1f ns  %8.1f G lane-steps/s\n", name,
               recs * 32.0 / 1e6, wavesPerSimd, ms, nsPerWaveStepPerCU, ms * 1e6 / steps, glane);
    };
    unsigned sizes[] = {1u << 15, 1u << 17, 1u << 19, 1u << 21, 1u << 23};  // 1 MB, 4 MB, 16 MB, 64 MB, 256 MB
    int occ[] = {1, 2, 3, 4, 6, 8};
    for (unsigned recs : sizes)
        for (int w : occ) {
            run("2 loads, no ALU", chase<2, 0>, recs, w);
        }
    for (int w : occ) run("1 load, no ALU", chase<1, 0>, 1u << 17, w);
    for (int w : occ) run("2 loads + 64 dep VALU", chase<2, 32>, 1u << 17, w);
    for (int w : occ) run("2 loads + 64 dep VALU", chase<2, 32>, 1u << 21, w);
    for (int w : {1, 2}) run("LDS 2 loads, no ALU", chaseLds<2, 0>, 2048, w);
    for (int w : {1, 2}) run("LDS 2 loads + 64 dep VALU", chaseLds<2, 32>, 2048, w);
    return 0;
}
