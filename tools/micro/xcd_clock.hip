// Round 4: why does a latency-bound sweep on XCDs 0-3 take twice as long while a GEMM runs on XCDs 4-7?
// Candidate: the package power cap is ONE budget -- matrix work on half the chip pulls the shader clock of every XCD down.
// A probe wave per XCD times a fixed chain of dependent VALU instructions against the constant 100 MHz counter (s_memrealtime)
// and the shader-cycle counter (s_memtime), (a) alone, (b) beside a bf16 MFMA register loop confined to XCDs 4-7, (c) beside an HBM
// streaming read confined to XCDs 4-7, (d) beside the MFMA loop on every XCD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_clock.hip -o tools/micro/xcd_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }

// out[8 * blk + {0: xcc, 1: shader cycles, 2: 100 MHz ticks, 3: chain length}]
__global__ __launch_bounds__(64) void probe(long long* out, int n, float* sink) {
    float v = (float)threadIdx.x;
    const long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v = __builtin_fmaf(v, 1.0000001f, 0.5f);      // 16 dependent v_fma_f32
    }
    const long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (v == 1.2345e30f) sink[threadIdx.x] = v;
    if (threadIdx.x == 0) { long long* o = out + 8 * blockIdx.x; o[0] = xcc_id(); o[1] = c1 - c0; o[2] = r1 - r0; o[3] = (long long)n * 16; }
}

__global__ __launch_bounds__(512) void mfma_load(const unsigned* seed, int iters, float* out, int xlo, int xn) {
    const int x = (int)xcc_id();
    if (x < xlo || x >= xlo + xn) return;
    unsigned s0 = seed[threadIdx.x & 63], s1 = seed[64 + (threadIdx.x & 63)];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 ua = {s0, s1, s0 * 3u, s1 * 5u}, ub = {s1, s0 * 7u, s1 * 11u, s0};
    for (int i = 0; i < 4; ++i) { ua[i] &= 0xbfffbfffu; ub[i] &= 0xbfffbfffu; }
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < 4; ++n) s += acc[n][0];
    if (s == 1.2345e30f) out[threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void hbm_load(const f32x4* src, size_t n4, int reps, float* out, int xlo, int xn) {
    const int x = (int)xcc_id();
    if (x < xlo || x >= xlo + xn) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) acc += src[i];
    if (acc[0] == 1.2345e30f) out[threadIdx.x] = acc[0];
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("%s, %d CUs\n", pr.name, pr.multiProcessorCount);
    long long* out; float* sink; unsigned* seed; f32x4* big;
    const size_t big_bytes = (size_t)2 << 30;
    hipMalloc(&out, 64 * 8 * 8); hipMalloc(&sink, 4096); hipMalloc(&seed, 512); hipMalloc(&big, big_bytes);
    unsigned h[128]; for (int i = 0; i < 128; ++i) h[i] = (unsigned)rand() * 2654435761u + (unsigned)rand();
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemset(big, 0, big_bytes);
    hipStream_t s_load, s_probe; hipStreamCreate(&s_load); hipStreamCreate(&s_probe);
    const int n = 40000;                  // 640k dependent FMAs: ~1 ms at 2.4 GHz with 4-cycle issue
    struct Case { const char* name; int kind, xlo, xn; } cases[] = {
        {"probe alone", 0, 0, 0}, {"beside bf16 MFMA loop on XCDs 4-7", 1, 4, 4}, {"beside HBM streaming read on XCDs 4-7", 2, 4, 4},
        {"beside bf16 MFMA loop on ALL XCDs", 1, 0, 8}, {"beside HBM streaming read on ALL XCDs", 2, 0, 8}, {"probe alone (again)", 0, 0, 0}};
    for (auto& c : cases) {
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        if (c.kind == 1) { hipEventRecord(e0, s_load); hipLaunchKernelGGL(mfma_load, dim3(pr.multiProcessorCount), dim3(512), 0, s_load, seed, 60000, sink, c.xlo, c.xn); hipEventRecord(e1, s_load); }
        if (c.kind == 2) { hipEventRecord(e0, s_load); hipLaunchKernelGGL(hbm_load, dim3(pr.multiProcessorCount * 8), dim3(256), 0, s_load, big, big_bytes / 16, c.xn == 8 ? 8 : 16, sink, c.xlo, c.xn); hipEventRecord(e1, s_load); }
        // let the load get going, then probe twice (the second probe sits well inside the load)
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(8), dim3(64), 0, s_probe, out, n, sink);
            hipStreamSynchronize(s_probe);
        }
        long long ho[64];
        hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
        hipDeviceSynchronize();
        float lms = 0.f; if (c.kind) hipEventElapsedTime(&lms, e0, e1);
        printf("%-42s load kernel %7.3f ms |", c.name, lms);
        for (int b = 0; b < 8; ++b) {
            const double us = ho[8 * b + 2] / 100.0;         // 100 MHz ticks -> us
            printf(" x%lld: %.0f MHz (%.2f cyc/fma)", ho[8 * b], ho[8 * b + 1] / us, (double)ho[8 * b + 1] / ho[8 * b + 3]);
        }
        printf("\n");
    }
    return 0;
}
