// What does the chip sustain on v_mfma_f32_32x32x16_bf16 when every SIMD issues them back to back?  (are the
// split-precision GEMMs bound by the schedule or by the package power?)  Pure register loop, no LDS, no memory:
//   waves per SIMD 1 / 2, independent accumulators 2 / 4 / 8, operand data zeros / random bits.
// Prints wall time (events), shader cycles of wave 0 (s_memtime), effective clock, TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x4v __attribute__((ext_vector_type(4)));
// Round 4: the same loop on v_mfma_f32_16x16x32_bf16 (the recurrent sweeps' instruction): one wave issues one every ~32 cycles whatever the
// instruction's size, so the small form moves half the flops per issue slot (profiles/r04_micro_mfma_16x16x32.txt).
template <int NACC>
__global__ __launch_bounds__(1024) void mfma_loop16(const unsigned* seed, int iters, float* out, long long* cyc) {
    unsigned s0 = seed[threadIdx.x & 63], s1 = seed[64 + (threadIdx.x & 63)];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 ua = {s0, s1, s0 * 3u, s1 * 5u}, ub = {s1, s0 * 7u, s1 * 11u, s0};
    for (int i = 0; i < 4; ++i) { ua[i] &= 0xbfffbfffu; ub[i] &= 0xbfffbfffu; }
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    f32x4v acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) acc[n][e] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[n], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) s += acc[n][0];
    if (s == 1.2345e30f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ... and with DISTINCT operand registers per instruction, as a register-resident weight slice has them (24 A operands, 2 B operands)
template <int NACC>
__global__ __launch_bounds__(1024) void mfma_loop16d(const unsigned* seed, int iters, float* out, long long* cyc) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    bf16x8 a[24], b[2];
    for (int i = 0; i < 24; ++i) {
        unsigned s0 = seed[(threadIdx.x + i) & 63], s1 = seed[64 + ((threadIdx.x + 3 * i) & 63)];
        u4 ua = {s0, s1, s0 * 3u + i, s1 * 5u};
        for (int e = 0; e < 4; ++e) ua[e] &= 0xbfffbfffu;
        a[i] = __builtin_bit_cast(bf16x8, ua);
        asm volatile("" : "+v"(a[i]));
    }
    for (int i = 0; i < 2; ++i) {
        unsigned s0 = seed[(threadIdx.x + 7 * i) & 63];
        u4 ub = {s0, s0 * 7u, s0 * 11u + i, s0};
        for (int e = 0; e < 4; ++e) ub[e] &= 0xbfffbfffu;
        b[i] = __builtin_bit_cast(bf16x8, ub);
        asm volatile("" : "+v"(b[i]));
    }
    f32x4v acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) acc[n][e] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 24 / NACC) {      // 24 MFMAs per trip = NACC per `iteration' of the other loops
#pragma unroll
        for (int i = 0; i < 24; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[i & 1], acc[i % NACC], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) s += acc[n][0];
    if (s == 1.2345e30f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(1024) void mfma_loop(const unsigned* seed, int iters, float* out, long long* cyc) {
    unsigned s0 = seed[threadIdx.x & 63], s1 = seed[64 + (threadIdx.x & 63)];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 ua = {s0, s1, s0 * 3u, s1 * 5u}, ub = {s1, s0 * 7u, s1 * 11u, s0};
    // keep exponents sane: clear the top exponent bits of every bf16 half (|x| < 2)
    for (int i = 0; i < 4; ++i) { ua[i] &= 0xbfffbfffu; ub[i] &= 0xbfffbfffu; }
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) s += acc[n][0];
    if (s == 1.2345e30f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
static void run(int waves_per_simd, bool random, int cus, int small = 0) {
    unsigned h[128];
    for (int i = 0; i < 128; ++i) h[i] = random ? (unsigned)rand() * 2654435761u + (unsigned)rand() : 0u;
    unsigned* d; float* out; long long* cyc;
    hipMalloc(&d, sizeof(h)); hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const int threads = 256 * waves_per_simd, iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a, 0);
        if (small == 2) hipLaunchKernelGGL(mfma_loop16d<NACC>, dim3(cus), dim3(threads), 0, 0, d, iters, out, cyc);
        else if (small) hipLaunchKernelGGL(mfma_loop16<NACC>, dim3(cus), dim3(threads), 0, 0, d, iters, out, cyc);
        else hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus), dim3(threads), 0, 0, d, iters, out, cyc);
        hipEventRecord(b, 0); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)cus * 4 * waves_per_simd * iters * NACC;      // MFMAs
    printf("%s CUs %3d acc %d  waves/SIMD %d  data %-6s: %8.3f ms  wave0 %lld cyc (%.1f cyc per MFMA of a wave, %.1f per MFMA and SIMD)  clock %.2f GHz  %7.1f TFLOP/s\n",
           small == 2 ? "16x16x32 distinct operands" : (small ? "16x16x32" : "32x32x16"), cus, NACC, waves_per_simd, random ? "random" : "zeros", ms, c, (double)c / ((double)iters * NACC),
           (double)c / ((double)iters * NACC * waves_per_simd), c / ms / 1e6, nm * (small ? 16384.0 : 32768.0) / ms / 1e9);
    hipFree(d); hipFree(out); hipFree(cyc);
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    printf("%s, %d CUs\n", pr.name, cus);
    if (getenv("MFMA_SMALL")) {         // round 4: instruction size x waves per SIMD, on few CUs (no power throttling) and on all of them
        for (int ncu : {32, cus})
            for (int sm = 0; sm < 3; ++sm)
                for (int wv = 1; wv <= 4; ++wv) { if (sm == 2) run<3>(wv, true, ncu, 2); else run<4>(wv, true, ncu, sm); }
        return 0;
    }
    for (int rnd = 0; rnd < 2; ++rnd) {
        run<2>(1, rnd, cus); run<4>(1, rnd, cus); run<8>(1, rnd, cus);
        run<4>(2, rnd, cus); run<8>(2, rnd, cus);
    }
    return 0;
}
