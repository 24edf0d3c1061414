// Prototype (round 6, session 14): the split-precision TN contraction  C[m][n] = sum_k A[k][m] B[k][n]  (the weight gradients) fed by LDS-DMA.
//   A : the sweeps' PK gate-gradient image (physical row k even = (hi, hi) bf16 pairs of steps k, k+1; row k+1 = their (lo, lo) pairs), M contiguous
//   B : fp32 rows (the layer's input / its own output), N contiguous; split into (hi, lo) by the consumer when it forms its fragments
// 256 x 256 output tile, eight waves (2 x 4, each 128 x 64), K stages of SK rows: every stage row is ONE `buffer_load_dwordx4 ... lds` (1 KiB, no
// staging registers, no conversion pass, no ds_write), NST stages in LDS, ONE barrier per stage, counted vmcnt (never 0 in the loop).
// Per output element the MFMA sequence is the shipped kernel's (gemm_bf16x3.hip: k ascending, (b_hi a_lo), (b_lo a_hi), (b_hi a_hi) per 16 k).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gemm_tn_dma.hip -o tools/micro/gemm_tn_dma && ./tools/micro/gemm_tn_dma [M N K splits]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* ldsp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BM = 256, BN = 256, NTH = 512;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct P {
    const float* A; int lda;      // PK image, (K rows) x lda words
    const float* B; int ldb;      // fp32, (K rows) x ldb
    float* part;                  // [splits][M][N]
    int M, N, K, kchunk, splits, gx, gy;
    const float* B1; float* part1; int pair; int noload;      // pair: a second problem with the SAME A (dW_ih / dW_hh of a GRU layer): slot parity selects it
};

// workgroup -> (problem, tile): an XCD (blockIdx % 8) owns a contiguous share of the tile order, so the tiles that share operand panels run on ONE L2,
// and the two problems of a pair sit in neighbouring slots of that XCD
__device__ __forceinline__ bool map_block(const P& p, int& prob, int& tile) {
    const int x8 = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int np = p.pair ? 2 : 1;
    prob = slot % np;
    const int s2 = slot / np;
    const int ntiles = p.gx * p.gy * p.splits;
    const int q = ntiles / 8, r = ntiles % 8;
    const int lo = x8 < r ? x8 * (q + 1) : r * (q + 1) + (x8 - r) * q;
    const int hi = lo + (x8 < r ? q + 1 : q);
    tile = lo + s2;
    return tile < hi;
}

// SK rows per stage (16: one MFMA k-step), NST stages
template <int SK, int NST>
__global__ __launch_bounds__(NTH) void tn_dma(P p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned smem[];
    constexpr int ROWW = 256;                      // words per stage row (both operands: 256 m / 256 n)
    constexpr int STW = 2 * SK * ROWW;             // words per stage: SK rows of A, SK rows of B
    constexpr int PPW = 2 * SK / 8;                // DMA pieces per wave and stage
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3, half = lane >> 5, l31 = lane & 31;
    int prob, t;
    if (!map_block(p, prob, t)) return;
    const int bx = t % p.gx, by = (t / p.gx) % p.gy, bz = t / (p.gx * p.gy);
    const int m0 = by * BM, n0 = bx * BN, kb = bz * p.kchunk, ke = min(p.K, kb + p.kchunk);
    const int nst = (ke - kb + SK - 1) / SK;

    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0xfffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(prob ? p.B1 : p.B), 0, 0xfffffff0u, 0x00020000);

    // stage `st` (k rows kb + st*SK ..): wave w copies rows w, w+8, .. of the 2*SK stage rows (A rows first, then B rows)
    auto issue = [&](int st) {
        unsigned* base = smem + (st % NST) * STW;
        const int k0 = kb + st * SK;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            constexpr int dummy = 0; (void)dummy;
            const bool isA = q < SK / 8;           // compile-time: rows [0, SK) are A's, [SK, 2 SK) B's; wave w takes rows w, w + 8, ..
            const int r = w + 8 * q;
            const int kr = k0 + (isA ? r : r - SK);       // (every stage is complete: the host rounds the K chunks to 32 rows)
            unsigned* dst = base + r * ROWW;
            if (isA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (ldsp)dst, 16, (unsigned)lane * 16u, ((unsigned)kr * (unsigned)p.lda + (unsigned)m0) * 4u, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (ldsp)dst, 16, (unsigned)lane * 16u, ((unsigned)kr * (unsigned)p.ldb + (unsigned)n0) * 4u, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // prologue: NST - 1 stages in flight
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) if (st < nst) issue(st);

    for (int it = 0; it < nst; ++it) {
        // my pieces of stage `it` have landed (NST - 2 later stages of mine may still be in flight); near the end fewer are outstanding: wait for all
        if (it + NST - 2 < nst) wait_vm<(NST - 2) * PPW>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();              // everybody's pieces of stage `it` landed; everybody finished reading stage it - 1
        if (it + NST - 1 < nst) issue(it + NST - 1);      // into the buffer stage it - 1 occupied
        const unsigned* sa = smem + (it % NST) * STW;
        const unsigned* sb = sa + SK * ROWW;
#pragma unroll
        for (int s = 0; s < SK / 16; ++s) {
            bf16x8 ah[4], al[4], bh[2], bl[2];
            // A fragments: m = wm*128 + i*32 + l31, pairs s*8 + half*4 + {0..3}: hi words in rows 2p, lo words in rows 2p + 1
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned* q0 = sa + (s * 16 + half * 8) * ROWW + wm * 128 + i * 32 + l31;
                u32x4 h = {q0[0], q0[2 * ROWW], q0[4 * ROWW], q0[6 * ROWW]};
                u32x4 l = {q0[ROWW], q0[3 * ROWW], q0[5 * ROWW], q0[7 * ROWW]};
                ah[i] = __builtin_bit_cast(bf16x8, h); al[i] = __builtin_bit_cast(bf16x8, l);
            }
            // B fragments: n = wn*64 + j*32 + l31, k = s*16 + half*8 + {0..7}: eight fp32, split here exactly as the shipped staging pass does
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* q0 = reinterpret_cast<const float*>(sb) + (s * 16 + half * 8) * ROWW + wn * 64 + j * 32 + l31;
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = q0[e * ROWW];
                unsigned hw[4], lw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2v v = {x[2 * e], x[2 * e + 1]};
                    hw[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
                    const f32x2v d = {x[2 * e] - __uint_as_float(hw[e] << 16), x[2 * e + 1] - __uint_as_float(hw[e] & 0xffff0000u)};
                    lw[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2v));
                }
                u32x4 h = {hw[0], hw[1], hw[2], hw[3]}, l = {lw[0], lw[1], lw[2], lw[3]};
                bh[j] = __builtin_bit_cast(bf16x8, h); bl[j] = __builtin_bit_cast(bf16x8, l);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                }
        }
    }
    float* outp = (prob ? p.part1 : p.part) + (size_t)bz * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * half;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<f32x4*>(outp + (size_t)m * p.N + n) = v;
            }
    }
}

// v1: the two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER apart: while one group issues its 24 MFMAs the other reads and splits its
// fragments -- the matrix pipe of a SIMD always has a wave feeding it, the LDS / VALU work of the other hides behind it.  Stages of 16 rows.
template <int NST>
__global__ __launch_bounds__(NTH) void tn_dma_pp(P p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned smem[];
    constexpr int SK = 16, ROWW = 256, STW = 2 * SK * ROWW, PPW = 2 * SK / 8;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3, half = lane >> 5, l31 = lane & 31;
    const int grp = wm;
    int prob, t;
    if (!map_block(p, prob, t)) return;
    const int bx = t % p.gx, by = (t / p.gx) % p.gy, bz = t / (p.gx * p.gy);
    const int m0 = by * BM, n0 = bx * BN, kb = bz * p.kchunk, ke = min(p.K, kb + p.kchunk);
    const int nst = (ke - kb) / SK;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0xfffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(prob ? p.B1 : p.B), 0, 0xfffffff0u, 0x00020000);
    auto issue = [&](int st) {
        unsigned* base = smem + (st % NST) * STW;
        const int k0 = kb + st * SK;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const bool isA = q < SK / 8;
            const int r = w + 8 * q;
            const int kr = k0 + (isA ? r : r - SK);
            unsigned* dst = base + r * ROWW;
            if (isA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (ldsp)dst, 16, (unsigned)lane * 16u, ((unsigned)kr * (unsigned)p.lda + (unsigned)m0) * 4u, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (ldsp)dst, 16, (unsigned)lane * 16u, ((unsigned)kr * (unsigned)p.ldb + (unsigned)n0) * 4u, 0, 0);
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) if (st < nst) issue(st);
    if (NST - 1 <= nst) wait_vm<(NST - 2) * PPW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                  // stage 0 is complete
    if (grp == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one barrier behind group 0
    for (int it = 0; it < nst; ++it) {
        const bool more = it + NST - 1 < nst;
        if (more) issue(it + NST - 1);             // into the buffer of stage it - 1 (both groups finished reading it before the barrier in front of this slot)
        const unsigned* sa = smem + (it % NST) * STW;
        const unsigned* sb = sa + SK * ROWW;
        bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned* q0 = sa + (half * 8) * ROWW + wm * 128 + i * 32 + l31;
            u32x4 h = {q0[0], q0[2 * ROWW], q0[4 * ROWW], q0[6 * ROWW]};
            u32x4 l = {q0[ROWW], q0[3 * ROWW], q0[5 * ROWW], q0[7 * ROWW]};
            ah[i] = __builtin_bit_cast(bf16x8, h); al[i] = __builtin_bit_cast(bf16x8, l);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* q0 = reinterpret_cast<const float*>(sb) + (half * 8) * ROWW + wn * 64 + j * 32 + l31;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = q0[e * ROWW];
            unsigned hw[4], lw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2v v = {x[2 * e], x[2 * e + 1]};
                hw[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
                const f32x2v d = {x[2 * e] - __uint_as_float(hw[e] << 16), x[2 * e + 1] - __uint_as_float(hw[e] & 0xffff0000u)};
                lw[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2v));
            }
            u32x4 h = {hw[0], hw[1], hw[2], hw[3]}, l = {lw[0], lw[1], lw[2], lw[3]};
            bh[j] = __builtin_bit_cast(bf16x8, h); bl[j] = __builtin_bit_cast(bf16x8, l);
        }
        // every wave's pieces of stage it + 1 must have landed before the barrier in front of group 0's next read slot: group 1 is in its read slot then,
        // group 0 in its MFMA slot
        if (grp == 1) { if (more) wait_vm<(NST - 2) * PPW>(); else wait_vm<0>(); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 1 ? bl[j] : bh[j], term == 0 ? al[i] : ah[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) { if (more) wait_vm<(NST - 2) * PPW>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    float* outp = (prob ? p.part1 : p.part) + (size_t)bz * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * half;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<f32x4*>(outp + (size_t)m * p.N + n) = v;
            }
    }
}

// v2: as v1, but a stage's four DMA pieces are issued BETWEEN the MFMAs of the wave's MFMA slot (an LDS-DMA piece costs 100-185 cycles of issue in a slot
// that is busy with ds_reads and VALU, ~60 among bare MFMAs whose execution hides it), not at the head of its read slot.
template <int NST, int HEADQ>
__global__ __launch_bounds__(NTH) void tn_dma_pp2(P p, long long* trace) {
    extern __shared__ __attribute__((aligned(1024))) unsigned smem[];
    constexpr int SK = 16, ROWW = 256, STW = 2 * SK * ROWW, PPW = 2 * SK / 8;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3, half = lane >> 5, l31 = lane & 31;
    const int grp = wm;
    int prob, t;
    if (!map_block(p, prob, t)) return;
    const int bx = t % p.gx, by = (t / p.gx) % p.gy, bz = t / (p.gx * p.gy);
    const int m0 = by * BM, n0 = bx * BN, kb = bz * p.kchunk, ke = min(p.K, kb + p.kchunk);
    const int nst = (ke - kb) / SK;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0xfffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(prob ? p.B1 : p.B), 0, 0xfffffff0u, 0x00020000);
    auto piece = [&](int st, int q) {
        unsigned* base = smem + (st % NST) * STW;
        const int k0 = kb + st * SK;
        const bool isA = q < SK / 8;
        const int r = w + 8 * q;
        const int kr = k0 + (isA ? r : r - SK);
        unsigned* dst = base + r * ROWW;
        if (isA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (ldsp)dst, 16, (unsigned)lane * 16u, ((unsigned)kr * (unsigned)p.lda + (unsigned)m0) * 4u, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (ldsp)dst, 16, (unsigned)lane * 16u, ((unsigned)kr * (unsigned)p.ldb + (unsigned)n0) * 4u, 0, 0);
    };
    auto issue = [&](int st) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) piece(st, q);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) if (st < nst) issue(st);
    if (NST - 1 <= nst) wait_vm<(NST - 2) * PPW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                  // stage 0 is complete
    if (grp == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one barrier behind group 0
    long long ts[5] = {0, 0, 0, 0, 0}; long long acc_t[4] = {0, 0, 0, 0};
#define STAMP(i) do { if (trace) ts[i] = (long long)__builtin_readcyclecounter(); } while (0)
    bf16x8 ah[4], al[4], bh[2], bl[2];
    for (int it = 0; it < nst; ++it) {
        const bool more = it + NST - 1 < nst && p.noload < 2;      // noload: 1 no fragment reads, 2 neither reads nor DMA, 3 reads but no DMA
        STAMP(0);
        const unsigned* sa = smem + (it % NST) * STW;
        const unsigned* sb = sa + SK * ROWW;
        if (!((p.noload == 1 || p.noload == 2) && it > 0)) {
        // Fragment words by hand-placed ds_read2st64_b32: each instruction fetches the two words of ONE operand register pair (rows e, e + 2 of the hi / lo
        // plane; rows e, e + 1 of B), so the results land where the MFMA reads them.  Left to the compiler the loads pair words of DIFFERENT fragments
        // (columns m, m + 32) and ~100 v_mov per stage reassemble the operands: the read slot was VALU-bound (820 cycles) and longer than the MFMA slot.
        typedef unsigned long long u64t;
        u64t ra[4][2][2], rbx[2][4];
        {
            const unsigned abase = (unsigned)(unsigned long long)(ldsp)(sa + (half * 8) * ROWW + wm * 128 + l31);
            const unsigned bbase = (unsigned)(unsigned long long)(ldsp)(sb + (half * 8) * ROWW + wn * 64 + l31);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned ad = abase + i * 128;
                asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:8" : "=v"(ra[i][0][0]) : "v"(ad));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:16 offset1:24" : "=v"(ra[i][0][1]) : "v"(ad));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:4 offset1:12" : "=v"(ra[i][1][0]) : "v"(ad));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:20 offset1:28" : "=v"(ra[i][1][1]) : "v"(ad));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned ad = bbase + j * 128;
                asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:4" : "=v"(rbx[j][0]) : "v"(ad));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:8 offset1:12" : "=v"(rbx[j][1]) : "v"(ad));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:16 offset1:20" : "=v"(rbx[j][2]) : "v"(ad));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:24 offset1:28" : "=v"(rbx[j][3]) : "v"(ad));
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0][0]), "+v"(ra[0][0][1]), "+v"(ra[0][1][0]), "+v"(ra[0][1][1]), "+v"(ra[1][0][0]), "+v"(ra[1][0][1]), "+v"(ra[1][1][0]), "+v"(ra[1][1][1]),
                           "+v"(ra[2][0][0]), "+v"(ra[2][0][1]), "+v"(ra[2][1][0]), "+v"(ra[2][1][1]), "+v"(ra[3][0][0]), "+v"(ra[3][0][1]), "+v"(ra[3][1][0]), "+v"(ra[3][1][1]),
                           "+v"(rbx[0][0]), "+v"(rbx[0][1]), "+v"(rbx[0][2]), "+v"(rbx[0][3]), "+v"(rbx[1][0]), "+v"(rbx[1][1]), "+v"(rbx[1][2]), "+v"(rbx[1][3]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            typedef u64t u64x2 __attribute__((ext_vector_type(2)));
            const u64x2 h = {ra[i][0][0], ra[i][0][1]}, l = {ra[i][1][0], ra[i][1][1]};
            ah[i] = __builtin_bit_cast(bf16x8, h); al[i] = __builtin_bit_cast(bf16x8, l);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned hw[4], lw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = __uint_as_float((unsigned)rbx[j][e]), x1 = __uint_as_float((unsigned)(rbx[j][e] >> 32));
                const f32x2v v = {x0, x1};
                hw[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
                const f32x2v d = {x0 - __uint_as_float(hw[e] << 16), x1 - __uint_as_float(hw[e] & 0xffff0000u)};
                lw[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2v));
            }
            u32x4 h = {hw[0], hw[1], hw[2], hw[3]}, l = {lw[0], lw[1], lw[2], lw[3]};
            bh[j] = __builtin_bit_cast(bf16x8, h); bl[j] = __builtin_bit_cast(bf16x8, l);
        }
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < HEADQ; ++q) piece(it + NST - 1, q);      // the first HEADQ pieces of stage it + 3 BEHIND the fragment reads of the read slot, the rest among the MFMAs
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(1);
        if (grp == 1) { if (it + NST - 2 < nst) wait_vm<(NST - 3) * PPW + HEADQ>(); else wait_vm<0>(); }      // (stage it + 3: HEADQ pieces so far)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        STAMP(2);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 1 ? bl[j] : bh[j], term == 0 ? al[i] : ah[i], acc[i][j], 0, 0, 0);
                if (i == 1 || i == 3) {
                    const int q = HEADQ + term * 2 + (i >> 1);
                    if (q < PPW && more) piece(it + NST - 1, q);
                }
            }
        __builtin_amdgcn_s_setprio(0);
        STAMP(3);
        if (grp == 0) { if (more) wait_vm<(NST - 2) * PPW>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        STAMP(4);
        if (trace && it >= 50 && it < 150) { for (int q = 0; q < 4; ++q) acc_t[q] += ts[q + 1] - ts[q]; }
    }
    if (trace && blockIdx.x == 0 && (tid == 0 || tid == 256)) { for (int q = 0; q < 4; ++q) trace[(tid >> 8) * 4 + q] = acc_t[q]; }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    float* outp = (prob ? p.part1 : p.part) + (size_t)bz * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * half;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<f32x4*>(outp + (size_t)m * p.N + n) = v;
            }
    }
}

template <int NST, int HEADQ>
static float run_pp2(const P& p, int reps) {
    const size_t lds = (size_t)NST * 2 * 16 * 256 * 4;
    CK(hipFuncSetAttribute((const void*)tn_dma_pp2<NST, HEADQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 g((unsigned)((p.gx * p.gy * p.splits + 7) / 8 * 8 * (p.pair ? 2 : 1)));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((tn_dma_pp2<NST, HEADQ>), g, dim3(NTH), lds, 0, p, (long long*)nullptr);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tn_dma_pp2<NST, HEADQ>), g, dim3(NTH), lds, 0, p, (long long*)nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int NST>
static float run_pp(const P& p, int reps) {
    const size_t lds = (size_t)NST * 2 * 16 * 256 * 4;
    CK(hipFuncSetAttribute((const void*)tn_dma_pp<NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 g((unsigned)((p.gx * p.gy * p.splits + 7) / 8 * 8 * (p.pair ? 2 : 1)));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((tn_dma_pp<NST>), g, dim3(NTH), lds, 0, p);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tn_dma_pp<NST>), g, dim3(NTH), lds, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

__global__ void make_pk(const float* x, unsigned* pk, int K, int M) {      // x (K x M) fp32 -> PK image
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)(K / 2) * M) return;
    const int pr = (int)(i / M), m = (int)(i % M);
    const float a = x[(size_t)(2 * pr) * M + m], b = x[(size_t)(2 * pr + 1) * M + m];
    const f32x2v v = {a, b};
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
    const f32x2v d = {a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xffff0000u)};
    const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2v));
    pk[(size_t)(2 * pr) * M + m] = h; pk[(size_t)(2 * pr + 1) * M + m] = l;
}

__global__ void fill_rand(float* x, long n, unsigned seed) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned s = (unsigned)i * 2654435761u + seed; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
    x[i] = ((float)(s & 0xffffff) / 8388608.0f - 1.0f);
}

__global__ void reduce_parts(const float* part, int splits, long MN, float* C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MN) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += part[(size_t)z * MN + i];
    C[i] = s;
}

// reference of a few output elements in double, from the fp32 operands
__global__ void ref_some(const float* A, const float* B, int K, int M, int N, const int* ms, const int* ns, int cnt, double* out) {
    const int i = blockIdx.x;
    if (i >= cnt) return;
    const int m = ms[i], n = ns[i];
    double s = 0.0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) s += (double)A[(size_t)k * M + m] * (double)B[(size_t)k * N + n];
    __shared__ double sh[256];
    sh[threadIdx.x] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[i] = sh[0];
}

template <int SK, int NST>
static float run(const P& p, int reps) {
    const size_t lds = (size_t)NST * 2 * SK * 256 * 4;
    CK(hipFuncSetAttribute((const void*)tn_dma<SK, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 g((unsigned)((p.gx * p.gy * p.splits + 7) / 8 * 8 * (p.pair ? 2 : 1)));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((tn_dma<SK, NST>), g, dim3(NTH), lds, 0, p);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tn_dma<SK, NST>), g, dim3(NTH), lds, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 768, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 153600;
    int splits = argc > 4 ? atoi(argv[4]) : 85;
    P p{};
    p.M = M; p.N = N; p.K = K; p.lda = M; p.ldb = N; p.gx = N / BN; p.gy = M / BM;
    p.kchunk = ((K + splits - 1) / splits + 31) / 32 * 32; p.splits = (K + p.kchunk - 1) / p.kchunk;
    float *Af, *B, *part, *C; unsigned* Apk;
    CK(hipMalloc(&Af, (size_t)K * M * 4)); CK(hipMalloc(&Apk, (size_t)K * M * 4)); CK(hipMalloc(&B, (size_t)K * N * 4));
    CK(hipMalloc(&part, (size_t)p.splits * M * N * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    fill_rand<<<(unsigned)(((long)K * M + 255) / 256), 256>>>(Af, (long)K * M, 1u);
    fill_rand<<<(unsigned)(((long)K * N + 255) / 256), 256>>>(B, (long)K * N, 7u);
    make_pk<<<(unsigned)(((long)(K / 2) * M + 255) / 256), 256>>>(Af, Apk, K, M);
    CK(hipDeviceSynchronize());
    p.A = reinterpret_cast<const float*>(Apk); p.B = B; p.part = part;
    p.pair = argc > 5 ? atoi(argv[5]) : 0;
    const int hot = argc > 6 ? atoi(argv[6]) : 0;      // 1: every stage row is row 0 of its operand (all pieces hit in L2 / TCP: what does the CU's fill path deliver?)  timing only
    if (hot) { p.lda = 0; p.ldb = 0; printf("HOT: row strides 0 -- results are garbage, timing only\n"); }
    if (p.pair) {
        float* B1; CK(hipMalloc(&B1, (size_t)K * N * 4)); CK(hipMalloc(&p.part1, (size_t)p.splits * M * N * 4));
        fill_rand<<<(unsigned)(((long)K * N + 255) / 256), 256>>>(B1, (long)K * N, 11u); CK(hipDeviceSynchronize());
        p.B1 = B1;
    }
    printf("TN %d x %d, K = %d, splits %d (k chunk %d), %d workgroups of 512\n", M, N, K, p.splits, p.kchunk, p.gx * p.gy * p.splits);
    const double flops = 2.0 * M * N * (double)K * (p.pair ? 2 : 1);
    struct { const char* name; float ms; } res[9];
    res[4] = {"ping-pong, 3 stages", run_pp<3>(p, 20)};
    res[5] = {"ping-pong, 5 stages", run_pp<5>(p, 20)};
    res[6] = {"ping-pong, 4 stages", run_pp<4>(p, 20)};
    res[7] = {"ping-pong, 4 stages, 2 pieces at the head", run_pp2<4, 2>(p, 20)};
    res[8] = {"ping-pong, 4 stages, DMA among the MFMAs", run_pp2<4, 0>(p, 20)};
    for (int nl = 0; nl < 4; ++nl) {
        p.noload = nl;
        long long* tr; CK(hipMalloc(&tr, 64)); CK(hipMemset(tr, 0, 64));
        const size_t lds = (size_t)4 * 2 * 16 * 256 * 4;
        const dim3 g((unsigned)((p.gx * p.gy * p.splits + 7) / 8 * 8 * (p.pair ? 2 : 1)));
        hipLaunchKernelGGL((tn_dma_pp2<4, 2>), g, dim3(NTH), lds, 0, p, tr);
        CK(hipDeviceSynchronize());
        long long h[8]; CK(hipMemcpy(h, tr, 64, hipMemcpyDeviceToHost));
        printf("(noload %d)\n", p.noload);
        for (int gq = 0; gq < 2; ++gq)
            printf("trace, group %d wave (ticks per stage, stages 50-149): issue + fragment reads + split %lld | vmcnt + barrier %lld | 24 MFMAs (+ DMA issue) %lld | vmcnt + barrier %lld\n",
                   gq, h[gq * 4] / 100, h[gq * 4 + 1] / 100, h[gq * 4 + 2] / 100, h[gq * 4 + 3] / 100);
    }
    p.noload = 0;
    (void)run_pp2<4, 0>(p, 1);                    // a clean run of the hand-placed-read variant: this is what the check below verifies
    // check this variant
    reduce_parts<<<(unsigned)(((long)M * N + 255) / 256), 256>>>(part, p.splits, (long)M * N, C);
    const int cnt = 64; std::vector<int> hm(cnt), hn(cnt);
    for (int i = 0; i < cnt; ++i) { hm[i] = (i * 37 + 5) % M; hn[i] = (i * 101 + 3) % N; }
    int *dm, *dn; double* dref; CK(hipMalloc(&dm, cnt * 4)); CK(hipMalloc(&dn, cnt * 4)); CK(hipMalloc(&dref, cnt * 8));
    CK(hipMemcpy(dm, hm.data(), cnt * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, hn.data(), cnt * 4, hipMemcpyHostToDevice));
    ref_some<<<cnt, 256>>>(Af, B, K, M, N, dm, dn, cnt, dref);
    std::vector<double> href(cnt); std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(href.data(), dref, cnt * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hC.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < cnt; ++i) { maxerr = fmax(maxerr, fabs(hC[(size_t)hm[i] * N + hn[i]] - href[i])); maxref = fmax(maxref, fabs(href[i])); }
    printf("check (64 elements vs double): max abs err %.3e, max |ref| %.3e, rel %.2e\n", maxerr, maxref, maxerr / maxref);
    res[0] = {"16 rows x 4 stages", run<16, 4>(p, 20)};
    res[1] = {"16 rows x 3 stages", run<16, 3>(p, 20)};
    res[2] = {"32 rows x 2 stages", run<32, 2>(p, 20)};
    res[3] = {"16 rows x 2 stages", run<16, 2>(p, 20)};
    for (auto& r : res) printf("%-42s %.4f ms = %.0f TFLOP/s of products (x3) = %.0f TFLOP/s fp32-equivalent\n", r.name, r.ms, 3 * flops / r.ms / 1e9, flops / r.ms / 1e9);
    return 0;
}
