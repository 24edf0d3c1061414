// Round 4: the shader clock is per XCD (xcd_clock.hip) -- so why does a sweep on XCDs 0-3 slow down 2x beside a
// GEMM on XCDs 4-7?  This probe times, on one lane per XCD, the three memory operations of the sweeps' hand-off:
//   (a) a dependent chain of sc1 (agent-scope, L1-bypassing) loads that hit in THIS XCD's L2,
//   (b) a plain store followed by s_waitcnt vmcnt(0) (the payload store + drain of the same-XCD fast path),
//   (c) the same with an sc1 (write-through) store,
// alone and beside an HBM streaming read / a streaming write on XCDs 4-7 or on every XCD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_lat.hip -o tools/micro/xcd_lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }

// ring: per block 16384 words (64 KB), ring[i] = next index (stride 67 words: a new 128-byte line every hop)
__global__ __launch_bounds__(64) void lat_probe(unsigned* rings, long long* out, int n) {
    unsigned* ring = rings + (size_t)blockIdx.x * 16384;
    if (threadIdx.x != 0) return;
    unsigned p = 0;
    for (int i = 0; i < 16384; ++i) p = __hip_atomic_load((gu32*)(ring + p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // warm: the ring now sits in this XCD's L2
    long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) p = __hip_atomic_load((gu32*)(ring + p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    long long* o = out + 16 * blockIdx.x;
    o[0] = xcc_id(); o[1] = c1 - c0; o[2] = r1 - r0;
    // (b) plain store + drain; the stored value keeps the ring intact
    unsigned* scratch = ring + 16384 * 8 * 0;      // same ring region, words we rewrite with their own value
    r0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_readcyclecounter();
    unsigned q = p & 16383;
    for (int i = 0; i < n; ++i) {
        const unsigned v = ring[q];
        __hip_atomic_store((gu32*)(scratch + q), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        q = (q + 67) & 16383;
    }
    c1 = __builtin_readcyclecounter(); r1 = __builtin_amdgcn_s_memrealtime();
    o[3] = c1 - c0; o[4] = r1 - r0;
    r0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        const unsigned v = ring[q];
        __hip_atomic_store((gu32*)(scratch + q), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        q = (q + 67) & 16383;
    }
    c1 = __builtin_readcyclecounter(); r1 = __builtin_amdgcn_s_memrealtime();
    o[5] = c1 - c0; o[6] = r1 - r0; o[7] = p + q;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void mfma_load(const unsigned* seed, int iters, float* out, int xlo, int xn) {
    const int x = (int)xcc_id();
    if (x < xlo || x >= xlo + xn) return;
    unsigned s0 = seed[threadIdx.x & 63] * 2654435761u + 12345u, s1 = seed[64 + (threadIdx.x & 63)] * 40503u + 977u;
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

__global__ __launch_bounds__(256) void hbm_load(const f32x4* src, f32x4* dst, size_t n4, int reps, float* out, int xlo, int xn, int write) {
    const int x = (int)xcc_id();
    if (x < xlo || x >= xlo + xn) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            if (write) dst[i] = acc; else acc += src[i];
        }
    if (acc[0] == 1.2345e30f) out[threadIdx.x] = acc[0];
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("%s, %d CUs; per XCD: sc1 L2-hit load chain | plain store + drain | sc1 store + drain  (shader cycles, ns)\n", pr.name, pr.multiProcessorCount);
    unsigned* rings; long long* out; float* sink; f32x4* big;
    const size_t big_bytes = (size_t)2 << 30;
    hipMalloc(&rings, 8 * 16384 * 4); hipMalloc(&out, 8 * 16 * 8); hipMalloc(&sink, 4096); hipMalloc(&big, big_bytes);
    unsigned* h = (unsigned*)malloc(8 * 16384 * 4);
    for (int b = 0; b < 8; ++b) for (int i = 0; i < 16384; ++i) h[b * 16384 + i] = (i + 67 * 33) & 16383;      // hop 2211 words: new line, new channel
    hipMemcpy(rings, h, 8 * 16384 * 4, hipMemcpyHostToDevice);
    hipMemset(big, 0, big_bytes);
    hipStream_t s_load, s_probe; hipStreamCreate(&s_load); hipStreamCreate(&s_probe);
    const int n = 2000;
    struct Case { const char* name; int kind, xlo, xn, write; } cases[] = {
        {"alone", 0, 0, 0, 0}, {"beside HBM read on XCDs 4-7", 1, 4, 4, 0}, {"beside HBM write on XCDs 4-7", 1, 4, 4, 1},
        {"beside HBM read on ALL XCDs", 1, 0, 8, 0}, {"beside HBM write on ALL XCDs", 1, 0, 8, 1}, {"beside bf16 MFMA loop on XCDs 4-7", 2, 4, 4, 0},
        {"beside bf16 MFMA loop on ALL XCDs", 2, 0, 8, 0}, {"alone (again)", 0, 0, 0, 0}};
    for (auto& c : cases) {
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        if (c.kind == 2) {
            hipEventRecord(e0, s_load);
            hipLaunchKernelGGL(mfma_load, dim3(pr.multiProcessorCount), dim3(512), 0, s_load, (const unsigned*)rings, 60000, sink, c.xlo, c.xn);
            hipEventRecord(e1, s_load);
        }
        if (c.kind == 1) {
            hipEventRecord(e0, s_load);
            hipLaunchKernelGGL(hbm_load, dim3(pr.multiProcessorCount * 8), dim3(256), 0, s_load, big, big, big_bytes / 16, c.xn == 8 ? 12 : 24, sink, c.xlo, c.xn, c.write);
            hipEventRecord(e1, s_load);
        }
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(lat_probe, dim3(8), dim3(64), 0, s_probe, rings, out, n); hipStreamSynchronize(s_probe); }
        long long ho[128];
        hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
        hipDeviceSynchronize();
        float lms = 0.f; if (c.kind) hipEventElapsedTime(&lms, e0, e1);
        const double gbs = c.kind == 1 ? (double)big_bytes * (c.xn == 8 ? 12 : 24) / (lms * 1e-3) / 1e9 : 0.0;
        printf("%-30s load %6.2f ms (%5.0f GB/s) |", c.name, lms, gbs);
        for (int b = 0; b < 8; b += (b == 0 ? 4 : 8)) {      // XCD of blocks 0 and 4
            const long long* o = ho + 16 * b;
            printf("  x%lld: ld %4.0f cyc %4.0f ns | st %4.0f cyc %4.0f ns | st.sc1 %4.0f cyc %4.0f ns", o[0], (double)o[1] / n, o[2] * 10.0 / n,
                   (double)o[3] / n, o[4] * 10.0 / n, (double)o[5] / n, o[6] * 10.0 / n);
        }
        printf("\n");
    }
    return 0;
}
