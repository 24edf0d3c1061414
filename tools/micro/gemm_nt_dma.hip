// Prototype (round 6, session 17): the split-precision NT projection  C[m][n] = sum_k X[m][k] W[n][k] + bias[n]  fed by LDS-DMA.
//   X : fp32 rows (K contiguous), split into (hi, lo) by the ONE wave that owns the row's fragment (wave tile 32 m x 256 n: no redundant conversion)
//   W : pre-split once per call into a stage image (pack_w below): for every 256-column tile and 16-k stage a contiguous 16 KiB block that IS the LDS
//       image -- [plane hi / lo][k octet][n] x 16 bytes -- so its DMA pieces are contiguous 1 KiB reads and its fragment reads conflict-free b128
// 256 x 256 output tile, eight waves (8 x 1), stages of 16 k, four stages in LDS, counted vmcnt, two wave groups one barrier apart (as gemm_tn_dma.hip).
// X's stage rows are 64 bytes: a DMA piece is 16 rows x 64 B; the 16-byte chunk a lane fetches is XOR-swizzled through the SOURCE address
// (chunk ^ (row >> 2) & 3) so that the b128 fragment reads of 32 consecutive rows are conflict-free.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gemm_nt_dma.hip -o tools/micro/gemm_nt_dma && ./tools/micro/gemm_nt_dma [M N K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* ldsp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BM = 256, BN = 256, NTH = 512, SK = 16, NST = 4;
constexpr int STW = 8192;                 // words per stage: X 256 rows x 16 fp32 (16 KiB) + W image block (16 KiB)
constexpr int PPW = 4;                    // DMA pieces per wave and stage: 2 of X, 2 of W

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void split4(f32x4 x, bf16x4& hi, bf16x4& lo) {
    const f32x2v v01 = {x[0], x[1]}, v23 = {x[2], x[3]};
    const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, bf16x2v));
    const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, bf16x2v));
    float d0 = x[0], d1 = x[1], d2 = x[2], d3 = x[3];
    asm("v_dot2c_f32_bf16 %0, %4, %6\n\tv_dot2c_f32_bf16 %1, %5, %6\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %5, %7\n\ts_nop 2"
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
        : "s"(0x0000bf80u), "s"(0xbf800000u), "v"(h0), "v"(h1));
    const f32x2v e01 = {d0, d1}, e23 = {d2, d3};
    const unsigned l0 = __builtin_bit_cast(unsigned, __builtin_convertvector(e01, bf16x2v));
    const unsigned l1 = __builtin_bit_cast(unsigned, __builtin_convertvector(e23, bf16x2v));
    const u32x2v h = {h0, h1}, l = {l0, l1};
    hi = __builtin_bit_cast(bf16x4, h);
    lo = __builtin_bit_cast(bf16x4, l);
}

struct P {
    const float* X; int ldx;      // (M rows) x ldx
    const unsigned* Wimg;         // [N / 256][K / 16][4096 words]
    const float* bias;
    float* C; int ldc;
    int M, N, K, gx, gy, nt;
};

// persistent: a workgroup walks the tiles lo + slot, lo + slot + SL, .. of its XCD's share; (tile, stage) is ONE iteration space, so the first stages of
// the next tile are in flight while this tile's last stages multiply, and a tile's 32 stores per lane drain behind the next tile's MFMAs (the counted
// vmcnt allows for them: 40 instead of 8 in the two waits that follow a write-out)
__global__ __launch_bounds__(NTH) void nt_dma(P p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned smem[];
    float* bias_l = reinterpret_cast<float*>(smem + NST * STW);       // the whole bias vector (N <= 1024)
    const int x8 = blockIdx.x & 7, slot = blockIdx.x >> 3, SL = gridDim.x >> 3;
    int lo, hi;
    {
        const int ntiles = p.gx * p.gy;
        const int q = ntiles / 8, r = ntiles % 8;
        lo = x8 < r ? x8 * (q + 1) : r * (q + 1) + (x8 - r) * q;
        hi = lo + (x8 < r ? q + 1 : q);
    }
    if (lo + slot >= hi) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int grp = w >> 2;
    const int nst = p.K / SK;
    for (int i = tid; i < p.N; i += NTH) bias_l[i] = p.bias ? p.bias[i] : 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, 0xfffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wimg, 0, 0xfffffff0u, 0x00020000);
    // X piece q (0 / 1) of wave w: rows 32 w + 16 q + lane / 4; LDS chunk position lane % 4 holds the row's chunk (lane % 4) ^ ((row >> 2) & 3)
    unsigned xrel[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = 32 * w + 16 * q + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        xrel[q] = ((unsigned)row * (unsigned)p.ldx) * 4u + (unsigned)c * 16u;
    }
    // issue side of the (tile, stage) space
    int ti = lo + slot, si = 0, icount = 0;
    bool iv = true;
    unsigned ixoff = (unsigned)(ti / p.gx) * BM * (unsigned)p.ldx * 4u, iwoff = (unsigned)(ti % p.gx) * (unsigned)nst * 16384u;
    auto issue = [&]() {
        unsigned* base = smem + (icount % NST) * STW;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (ldsp)(base + (32 * w + 16 * q) * 16), 16, xrel[q], ixoff + (unsigned)si * 64u, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (ldsp)(base + 4096 + (w * 2 + q) * 256), 16, (unsigned)lane * 16u, iwoff + (unsigned)si * 16384u + (unsigned)(w * 2 + q) * 1024u, 0, 0);
        ++icount;
        if (++si == nst) {
            si = 0; ti += SL; iv = ti < hi;
            if (iv) { ixoff = (unsigned)(ti / p.gx) * BM * (unsigned)p.ldx * 4u; iwoff = (unsigned)(ti % p.gx) * (unsigned)nst * 16384u; }
        }
    };
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) if (iv) issue();
    wait_vm<(NST - 2) * PPW>();                    // (K >= 48: the first tile has at least three stages)
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    const int xrow = 32 * w + l31;
    const int xs = (xrow >> 2) & 3;
    const int xp0 = xrow * 16 + (((2 * half) ^ xs) << 2), xp1 = xrow * 16 + (((2 * half + 1) ^ xs) << 2);
    int tc = lo + slot, sc = 0, ccount = 0, since = 100;
    while (tc < hi) {
        const bool more = iv;
        if (more) issue();
        ++since;
        const unsigned* sx = smem + (ccount % NST) * STW;
        const unsigned* sw = sx + 4096;
        bf16x8 ah, al, bh[8], bl[8];
        {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(sx + xp0), x1 = *reinterpret_cast<const f32x4*>(sx + xp1);
            bf16x4 h0, l0, h1, l1;
            split4(x0, h0, l0); split4(x1, h1, l1);
            ah = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            al = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bh[j] = *reinterpret_cast<const bf16x8*>(sw + ((0 * 2 + half) * 256 + j * 32 + l31) * 4);
            bl[j] = *reinterpret_cast<const bf16x8*>(sw + ((1 * 2 + half) * 256 + j * 32 + l31) * 4);
        }
        auto wait_next = [&]() {                   // this wave's pieces of the NEXT stage have landed
            if (!more) wait_vm<0>();
            else if (since == 1 || since == 2) wait_vm<(NST - 2) * PPW + 32>();       // the 32 stores of the tile before are younger than those pieces
            else wait_vm<(NST - 2) * PPW>();
        };
        if (grp == 1) wait_next();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 1 ? bl[j] : bh[j], term == 0 ? al : ah, acc[j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) wait_next();
        if (sc == nst - 1) {                       // the tile is complete: write it out, start the next from zero
            const int m = (tc / p.gx) * BM + 32 * w + l31, n0 = (tc % p.gx) * BN;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + j * 32 + 8 * g + 4 * half;
                    f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                    v += *reinterpret_cast<const f32x4*>(bias_l + n);
                    *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
                }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
            since = 0;
        }
        __builtin_amdgcn_s_barrier();
        ++ccount;
        if (++sc == nst) { sc = 0; tc += SL; }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
}

// v2: 128 x 256 tiles, FOUR waves (each 32 m x 256 n), three stages of 24 KiB: two such workgroups share a CU (2 x 72 KiB LDS, 2 x 4 waves of ~216
// registers) and desynchronise by themselves -- one's write-out (256 B per lane, store-issue bound) runs beside the other's MFMAs.  One barrier per stage.
constexpr int NTH2 = 256, NST2 = 3, STW2 = 2048 + 4096, PPW2 = 6;
template <bool SWAP>
__global__ __launch_bounds__(NTH2, 2) void nt_dma2(P p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned smem[];
    float* bias_l = reinterpret_cast<float*>(smem + NST2 * STW2);
    const int x8 = blockIdx.x & 7, slot = blockIdx.x >> 3, SL = gridDim.x >> 3;
    const int gy2 = p.M / 128;
    int lo, hi;
    {
        const int ntiles = p.gx * gy2;
        const int q = ntiles / 8, r = ntiles % 8;
        lo = x8 < r ? x8 * (q + 1) : r * (q + 1) + (x8 - r) * q;
        hi = lo + (x8 < r ? q + 1 : q);
    }
    if (lo + slot >= hi) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int nst = p.K / SK;
    for (int i = tid; i < p.N; i += NTH2) bias_l[i] = p.bias ? p.bias[i] : 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, 0xfffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wimg, 0, 0xfffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, 0xfffffff0u, 0x00020000);
    unsigned xrel[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = 32 * w + 16 * q + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        xrel[q] = ((unsigned)row * (unsigned)p.ldx) * 4u + (unsigned)c * 16u;
    }
    int ti = lo + slot, si = 0, icount = 0;
    bool iv = true;
    unsigned ixoff = (unsigned)(ti / p.gx) * 128u * (unsigned)p.ldx * 4u, iwoff = (unsigned)(ti % p.gx) * (unsigned)nst * 16384u;
    auto issue = [&]() {
        unsigned* base = smem + (icount % NST2) * STW2;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (ldsp)(base + (32 * w + 16 * q) * 16), 16, xrel[q], ixoff + (unsigned)si * 64u, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (ldsp)(base + 2048 + (w * 4 + q) * 256), 16, (unsigned)lane * 16u, iwoff + (unsigned)si * 16384u + (unsigned)(w * 4 + q) * 1024u, 0, 0);
        ++icount;
        if (++si == nst) {
            si = 0; ti += SL; iv = ti < hi;
            if (iv) { ixoff = (unsigned)(ti / p.gx) * 128u * (unsigned)p.ldx * 4u; iwoff = (unsigned)(ti % p.gx) * (unsigned)nst * 16384u; }
        }
    };
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int st = 0; st < NST2 - 1; ++st) if (iv) issue();
    const int xrow = 32 * w + l31;
    const int xs = (xrow >> 2) & 3;
    const int xp0 = xrow * 16 + (((2 * half) ^ xs) << 2), xp1 = xrow * 16 + (((2 * half + 1) ^ xs) << 2);
    int tc = lo + slot, sc = 0, ccount = 0, since = 100;
    while (tc < hi) {
        // this wave's pieces of the stage about to be read have landed (one younger stage may be in flight; after a write-out its 32 stores too)
        ++since;
        if (!iv && ti >= hi) { /* drain */ }
        {
            const bool one_younger = icount > ccount + 1;
            if (!one_younger) wait_vm<0>();
            else if (since == 1) wait_vm<PPW2 + (SWAP ? 57 : 32)>();
            else wait_vm<PPW2>();
        }
        __builtin_amdgcn_s_barrier();              // everybody's pieces landed; everybody finished reading the stage before
        if (iv) issue();                           // into the buffer of the stage before
        const unsigned* sx = smem + (ccount % NST2) * STW2;
        const unsigned* sw = sx + 2048;
        bf16x8 ah, al, bh[8], bl[8];
        {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(sx + xp0), x1 = *reinterpret_cast<const f32x4*>(sx + xp1);
            bf16x4 h0, l0, h1, l1;
            split4(x0, h0, l0); split4(x1, h1, l1);
            ah = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            al = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bh[j] = *reinterpret_cast<const bf16x8*>(sw + ((0 * 2 + half) * 256 + j * 32 + l31) * 4);
            bl[j] = *reinterpret_cast<const bf16x8*>(sw + ((1 * 2 + half) * 256 + j * 32 + l31) * 4);
        }
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 0 ? al : ah, term == 1 ? bl[j] : bh[j], acc[j], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 1 ? bl[j] : bh[j], term == 0 ? al : ah, acc[j], 0, 0, 0);
        if (sc == nst - 1) {
            const int n0 = (tc % p.gx) * BN;
            if constexpr (SWAP) {
                // lane owns column n = j*32 + l31; register 4 g + e holds row 8 g + 4 half + e: a store instruction writes 2 rows x 128 contiguous bytes
                const int mb = (tc / p.gx) * 128 + 32 * w;
                if (p.nt == 3) {
                    // TIMING ONLY (wrong placement): what a transposed write-out would cost -- 16-byte stores, a store instruction = 8 rows x 128 contiguous bytes
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) {
                            const int m = mb + 8 * k4 + (lane >> 3), n = n0 + j * 32 + (lane & 7) * 4;
                            const u32x4 v = {__float_as_uint(acc[j][4 * k4]), __float_as_uint(acc[j][4 * k4 + 1]), __float_as_uint(acc[j][4 * k4 + 2]), __float_as_uint(acc[j][4 * k4 + 3])};
                            __builtin_amdgcn_raw_buffer_store_b128(v, rc, ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 4u, 0, 2);
                        }
                } else
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = n0 + j * 32 + l31;
                    const float bv = bias_l[n];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mb + 8 * (r >> 2) + 4 * half + (r & 3);
                        if (p.nt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][r] + bv), rc, ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 4u, 0, 2 /* nt */);
                        else p.C[(size_t)m * p.ldc + n] = acc[j][r] + bv;
                    }
                }
            } else {
            const int m = (tc / p.gx) * 128 + 32 * w + l31;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + j * 32 + 8 * g + 4 * half;
                    f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                    v += *reinterpret_cast<const f32x4*>(bias_l + n);
                    *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
            since = 0;
        }
        ++ccount;
        if (++sc == nst) { sc = 0; tc += SL; }
    }
}

// W (N x K fp32) -> the stage image: block (n tile, stage) = [plane][k octet][256 n] x 8 bf16
__global__ void pack_w(const float* W, int ldw, int N, int K, unsigned* img) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // one (n, k octet)
    const int oct = K / 8;
    if (i >= (long)N * oct) return;
    const int n = (int)(i / oct), o = (int)(i % oct);
    const float* src = W + (size_t)n * ldw + o * 8;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
    bf16x4 h0, l0, h1, l1;
    split4(x0, h0, l0); split4(x1, h1, l1);
    const int ntile = n >> 8, nn = n & 255, stage = o >> 1, half = o & 1;
    unsigned* blk = img + ((size_t)ntile * (K / 16) + stage) * 4096;
    const u32x2v a = __builtin_bit_cast(u32x2v, h0), b = __builtin_bit_cast(u32x2v, h1), c = __builtin_bit_cast(u32x2v, l0), d = __builtin_bit_cast(u32x2v, l1);
    u32x4 hv = {a[0], a[1], b[0], b[1]}, lv = {c[0], c[1], d[0], d[1]};
    *reinterpret_cast<u32x4*>(blk + ((0 * 2 + half) * 256 + nn) * 4) = hv;
    *reinterpret_cast<u32x4*>(blk + ((1 * 2 + half) * 256 + nn) * 4) = lv;
}

__global__ void fill_rand(float* x, long n, unsigned seed, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned s = (unsigned)i * 2654435761u + seed; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
    x[i] = ((float)(s & 0xffffff) / 8388608.0f - 1.0f) * scale;
}

__global__ void ref_some(const float* X, const float* W, const float* bias, int K, const int* ms, const int* ns, int cnt, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    double s = bias ? bias[ns[i]] : 0.0;
    for (int k = 0; k < K; ++k) s += (double)X[(size_t)ms[i] * K + k] * (double)W[(size_t)ns[i] * K + k];
    out[i] = s;
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 153600, N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 256;
    P p{};
    p.M = M; p.N = N; p.K = K; p.ldx = K; p.ldc = N; p.gx = N / BN; p.gy = M / BM;
    float *X, *W, *bias, *C; unsigned* img;
    CK(hipMalloc(&X, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&img, (size_t)N * K * 4)); CK(hipMalloc(&bias, N * 4));
    CK(hipMalloc(&C, (size_t)M * N * 4));
    fill_rand<<<(unsigned)(((long)M * K + 255) / 256), 256>>>(X, (long)M * K, 1u, 1.f);
    fill_rand<<<(unsigned)(((long)N * K + 255) / 256), 256>>>(W, (long)N * K, 7u, 0.0625f);
    fill_rand<<<(N + 255) / 256, 256>>>(bias, N, 3u, 1.f);
    CK(hipDeviceSynchronize());
    p.X = X; p.Wimg = img; p.bias = bias; p.C = C;
    const size_t lds = (size_t)NST * STW * 4 + 4096;
    CK(hipFuncSetAttribute((const void*)nt_dma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_xcd = (p.gx * p.gy + 7) / 8;
    const dim3 g((unsigned)((per_xcd < 32 ? per_xcd : 32) * 8));
    auto once = [&]() {
        pack_w<<<(unsigned)(((long)N * (K / 8) + 255) / 256), 256>>>(W, K, N, K, img);
        hipLaunchKernelGGL(nt_dma, g, dim3(NTH), lds, 0, p);
    };
    printf("NT %d x %d, K = %d: %d workgroups of 512\n", M, N, K, p.gx * p.gy);
    for (int i = 0; i < 2; ++i) once();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) once();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    const double flops = 2.0 * M * N * (double)K;
    printf("pack + projection %.4f ms = %.0f TFLOP/s of products (x3) = %.0f TFLOP/s fp32-equivalent\n", ms, 3 * flops / ms / 1e9, flops / ms / 1e9);
    for (int sw = 3; sw >= 0; --sw) {
        p.nt = sw == 2 ? 1 : (sw == 3 ? 3 : 0);
        const size_t lds2 = (size_t)NST2 * STW2 * 4 + 4096;
        CK(hipFuncSetAttribute((const void*)nt_dma2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        CK(hipFuncSetAttribute((const void*)nt_dma2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        const int per2 = (p.gx * (M / 128) + 7) / 8;
        const dim3 g2((unsigned)((per2 < 64 ? per2 : 64) * 8));
        auto once2 = [&]() {
            pack_w<<<(unsigned)(((long)N * (K / 8) + 255) / 256), 256>>>(W, K, N, K, img);
            if (sw) hipLaunchKernelGGL(nt_dma2<true>, g2, dim3(NTH2), lds2, 0, p); else hipLaunchKernelGGL(nt_dma2<false>, g2, dim3(NTH2), lds2, 0, p);
        };
        for (int i = 0; i < 2; ++i) once2();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) once2();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms2; CK(hipEventElapsedTime(&ms2, e0, e1)); ms2 /= 20;
        printf("v2 (128 x 256 tiles, two workgroups per CU%s): pack + projection %.4f ms = %.0f TFLOP/s of products\n", sw == 3 ? ", TIMING ONLY: 16-byte full-line nt stores" : sw == 2 ? ", roles swapped, nt stores" : (sw ? ", roles swapped: dword stores of full lines" : ""), ms2, 3 * flops / ms2 / 1e9);
    }
    const int cnt = 256; std::vector<int> hm(cnt), hn(cnt);
    for (int i = 0; i < cnt; ++i) { hm[i] = (int)(((long)i * 7919 + 13) % M); hn[i] = (i * 101 + 3) % N; }
    int *dm, *dn; double* dref; CK(hipMalloc(&dm, cnt * 4)); CK(hipMalloc(&dn, cnt * 4)); CK(hipMalloc(&dref, cnt * 8));
    CK(hipMemcpy(dm, hm.data(), cnt * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, hn.data(), cnt * 4, hipMemcpyHostToDevice));
    ref_some<<<(cnt + 63) / 64, 64>>>(X, W, bias, K, dm, dn, cnt, dref);
    std::vector<double> href(cnt); std::vector<float> hc(cnt);
    CK(hipMemcpy(href.data(), dref, cnt * 8, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < cnt; ++i) {
        float v; CK(hipMemcpy(&v, C + (size_t)hm[i] * N + hn[i], 4, hipMemcpyDeviceToHost));
        maxerr = fmax(maxerr, fabs(v - href[i])); maxref = fmax(maxref, fabs(href[i]));
    }
    printf("check (256 elements vs double): max abs err %.3e, max |ref| %.3e, rel %.2e\n", maxerr, maxref, maxerr / maxref);
    return 0;
}
