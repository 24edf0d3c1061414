// Split-precision GEMM for the large time-parallel contractions: fp32 operands in HBM, fp32 accumulate,
// products formed on the bf16 matrix cores with the 3-term split
//       a*b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi ,   x_hi = bf16(x), x_lo = bf16(x - x_hi)
// (|x - x_hi - x_lo| <= 2^-18 |x|, dropped a_lo*b_lo <= 2^-18 |ab|: relative error per product ~1e-5 worst case,
// ~4e-6 typical -- inside the path's 1e-4 parity budget, checked by the same tests as the exact-f32 kernel).
// v_mfma_f32_32x32x16_bf16 does 16 k per 32 cycles vs 2 k per 64 cycles for v_mfma_f32_32x32x2_f32, so three of them
// cost 6 cycles/k against 32: the contraction stops being MFMA-bound and runs at the rate fp32 operands can be fed.
// The split happens once per element while staging global -> LDS (v_cvt_pk_bf16_f32), never in HBM.
//
//   C[M,N] = opA(A)[M,K] * opB(B)[K,N] + bias[N] + beta*C        same operand forms / split-K / row shift as gemm.hip
#include "dep_common.h"

namespace {

constexpr int BN = 128, BK = 32, NT = 256;
constexpr int LDK = 40;      // bf16 per LDS row: 80-byte rows -> conflict-free ds_read_b128 fragments and ds_write_b64 staging

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct GemmP {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias; float beta;
    int seqT, shiftB;
    int kchunk, splits;
    float* part;
    int gx, gy;             // tile grid (x: N tiles, y: M tiles); the launch is 1-D, see dep_xcd_tile
    int ablate;             // debug (DEP_GEMM_ABLATE): 1 no epilogue stores, 2 no MFMA, 4 no tile reloads, 8 no LDS staging
};

// Operand tile of ROWS (128 or 256) rows x 32 k, 256 threads: a = tid&7, bq = tid>>3.
//   !TR (K-contiguous rows): r[i]      = row (mn0 + bq + 32 i),            k  = k0 + a*4 + 0..3     i < ROWS/32
//    TR (MN-contiguous)    : r[4jj + i] = k row (k0 + a*4 + i),           mn = mn0 + (bq + 32 jj)*4 + 0..3
template <bool TR, bool VEC, int ROWS>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int mn0, int MN, int k0, int Kend,
                                          int tid, float (&r)[ROWS / 32][4], int seqT, int shift) {
    const int a = tid & 7, bq = tid >> 3;
    if (!TR) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int mn = mn0 + bq + 32 * i, k = k0 + a * 4;
            const float* src = P + (size_t)mn * ld + k;
            if (VEC) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (mn < MN && k < Kend) v = *reinterpret_cast<const f32x4*>(src);
                r[i][0] = v[0]; r[i][1] = v[1]; r[i][2] = v[2]; r[i][3] = v[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[i][e] = (mn < MN && k + e < Kend) ? src[e] : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < ROWS / 128; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + a * 4 + i, mn = mn0 + (bq + 32 * jj) * 4;
                bool ok = k < Kend;
                if (seqT > 0) { const int tt = k % seqT + shift; ok = ok && tt >= 0 && tt < seqT; }
                const float* src = P + ((long)k + shift) * ld + mn;
                float (&rr)[4] = r[jj * 4 + i];
                if (VEC) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (ok && mn < MN) v = *reinterpret_cast<const f32x4*>(src);
                    rr[0] = v[0]; rr[1] = v[1]; rr[2] = v[2]; rr[3] = v[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rr[e] = (ok && mn + e < MN) ? src[e] : 0.f;
                }
            }
    }
}

// v_cvt_pk_bf16_f32 is a quarter-rate instruction, so each PAIR of values costs exactly two of them: hi pair =
// cvt(x0, x1); the fp32 images of the two hi halves come back with a shift / mask of the packed word (hipcc would
// otherwise re-convert every element on its own); lo pair = cvt(x0 - hi0, x1 - hi1).
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2v v = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
    const f32x2v d = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2v));
}
__device__ __forceinline__ void split4(f32x4 x, bf16x4& hi, bf16x4& lo) {
    unsigned h0, h1, l0, l1;
    split2(x[0], x[1], h0, l0);
    split2(x[2], x[3], h1, l1);
    const u32x2v h = {h0, h1}, l = {l0, l1};
    hi = __builtin_bit_cast(bf16x4, h);
    lo = __builtin_bit_cast(bf16x4, l);
}

// LDS images Sh/Sl: [ROWS rows (m or n)][LDK] bf16, k contiguous
template <bool TR, int ROWS>
__device__ __forceinline__ void store_tile(__bf16* Sh, __bf16* Sl, int tid, const float (&r)[ROWS / 32][4]) {
    const int a = tid & 7, bq = tid >> 3;
    if (!TR) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            f32x4 x = {r[i][0], r[i][1], r[i][2], r[i][3]};
            bf16x4 hi, lo; split4(x, hi, lo);
            const int o = (bq + 32 * i) * LDK + a * 4;
            *reinterpret_cast<bf16x4*>(Sh + o) = hi; *reinterpret_cast<bf16x4*>(Sl + o) = lo;
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < ROWS / 128; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 x = {r[jj * 4 + 0][e], r[jj * 4 + 1][e], r[jj * 4 + 2][e], r[jj * 4 + 3][e]};
                bf16x4 hi, lo; split4(x, hi, lo);
                const int o = ((bq + 32 * jj) * 4 + e) * LDK + a * 4;
                *reinterpret_cast<bf16x4*>(Sh + o) = hi; *reinterpret_cast<bf16x4*>(Sl + o) = lo;
            }
    }
}

// Persistent, cross-tile pipelined: a workgroup walks a strided list of output tiles taken from ITS XCD's contiguous
// share of the tile order (so tiles processed together on an XCD share operand panels in that L2) and treats
// (tile, k-tile) as one iteration space: the register prefetch issued in the last k-iteration of a tile already
// belongs to the next tile, and a tile's epilogue stores drain while the next tile's loads are in flight.
// Tile = BMT x 128 (BMT = 256 for the big contractions: these kernels are bound by the L2 -> CU operand feed, not by
// the matrix pipes, and a 256-row tile moves 25 % fewer operand bytes per flop than 128 x 128), 4 waves as 2 x 2,
// each (BMT/2) x 64 = (BMT/64) x 2 MFMA tiles of 32x32.
template <bool TA, bool TB, bool VEC, int BMT>
__global__ __launch_bounds__(NT, (BMT == 256 ? 2 : 3)) void gemm_bf16x3(GemmP p) {
    constexpr bool A_TR = TA, B_TR = !TB;
    constexpr int MI = BMT / 64;
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * (BMT + BN) * LDK];
    __bf16* Ah = smem; __bf16* Al = smem + BMT * LDK; __bf16* Bh = smem + 2 * BMT * LDK; __bf16* Bl = Bh + BN * LDK;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int ntiles = p.gx * p.gy * p.splits;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int q = ntiles / 8, r = ntiles % 8;
    const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int hi = lo + (xcd < r ? q + 1 : q);
    int tile = lo + slot;
    if (tile >= hi) return;

    auto coords = [&](int t, int& m0, int& n0, int& kb, int& ke, int& bz) {
        const int bx = t % p.gx, by = (t / p.gx) % p.gy; bz = t / (p.gx * p.gy);
        m0 = by * BMT; n0 = bx * BN; kb = bz * p.kchunk; ke = min(p.K, kb + p.kchunk);
    };
    int m0, n0, kbeg, kend, bz;
    coords(tile, m0, n0, kbeg, kend, bz);

    f32x16 acc[MI][2];
    float ra[BMT / 32][4], rb[BN / 32][4];
    load_tile<A_TR, VEC, BMT>(p.A, p.lda, m0, p.M, kbeg, kend, tid, ra, 0, 0);
    load_tile<B_TR, VEC, BN>(p.B, p.ldb, n0, p.N, kbeg, kend, tid, rb, p.seqT, p.shiftB);

    while (true) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int next = tile + slots;
        const bool has_next = next < hi;
        int nm0 = 0, nn0 = 0, nkb = 0, nke = 0, nbz = 0;
        if (has_next) coords(next, nm0, nn0, nkb, nke, nbz);

        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            if (!(p.ablate & 8) || k0 == kbeg) {
                store_tile<A_TR, BMT>(Ah, Al, tid, ra);
                store_tile<B_TR, BN>(Bh, Bl, tid, rb);
            }
            __syncthreads();
            if (!(p.ablate & 4)) {
                if (k0 + BK < kend) {
                    load_tile<A_TR, VEC, BMT>(p.A, p.lda, m0, p.M, k0 + BK, kend, tid, ra, 0, 0);
                    load_tile<B_TR, VEC, BN>(p.B, p.ldb, n0, p.N, k0 + BK, kend, tid, rb, p.seqT, p.shiftB);
                } else if (has_next) {       // first k-tile of the NEXT output tile
                    load_tile<A_TR, VEC, BMT>(p.A, p.lda, nm0, p.M, nkb, nke, tid, ra, 0, 0);
                    load_tile<B_TR, VEC, BN>(p.B, p.ldb, nn0, p.N, nkb, nke, tid, rb, p.seqT, p.shiftB);
                }
            }
            if (!(p.ablate & 2))
#pragma unroll
            for (int s = 0; s < BK / 16; ++s) {
                const int ko = s * 16 + half * 8;
                bf16x8 ah[MI], al[MI], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int ro = (wm * (BMT / 2) + i * 32 + l31) * LDK + ko;
                    ah[i] = *reinterpret_cast<const bf16x8*>(Ah + ro); al[i] = *reinterpret_cast<const bf16x8*>(Al + ro);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ro = (wn * 64 + j * 32 + l31) * LDK + ko;
                    bh[j] = *reinterpret_cast<const bf16x8*>(Bh + ro); bl[j] = *reinterpret_cast<const bf16x8*>(Bl + ro);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        // operands swapped (D = B-rows x A-rows^T): a lane then owns ONE output row m = l31 and, per register
                        // quad, 4 consecutive n -> the epilogue stores 16 bytes per lane instead of 4
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                    }
            }
            __syncthreads();
        }

        const bool split = p.part != nullptr;
        float* outp = split ? p.part + (size_t)bz * p.M * p.N : p.C;
        const int ldo = split ? p.N : p.ldc;
        const bool v4 = ((ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(outp) & 15) == 0);
        const bool addb = !split && p.bias;
        const bool rmw = !split && p.beta != 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * (BMT / 2) + i * 32 + l31;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * half;
                    if (n >= p.N) continue;
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    float* dst = outp + (size_t)m * ldo + n;
                    if (v4 && n + 3 < p.N) {
                        if (addb) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if (rmw) v += p.beta * *reinterpret_cast<const f32x4*>(dst);
                        if (!(p.ablate & 1) || v[0] == 1.2345e30f) *reinterpret_cast<f32x4*>(dst) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) {
                                float x = v[e] + (addb ? p.bias[n + e] : 0.f);
                                if (rmw) x += p.beta * dst[e];
                                dst[e] = x;
                            }
                    }
                }
        }
        if (!has_next) break;
        tile = next; m0 = nm0; n0 = nn0; kbeg = nkb; kend = nke; bz = nbz;
    }
}

__global__ void splitk_reduce2(const float* __restrict__ part, int splits, int M, int N, float* C, int ldc,
                               const float* bias, float beta) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    // four independent partial sums: the loads of consecutive splits overlap instead of forming one dependent chain
    const size_t MN = (size_t)M * N;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 3 < splits; z += 4) {
        s0 += part[(size_t)z * MN + idx]; s1 += part[(size_t)(z + 1) * MN + idx];
        s2 += part[(size_t)(z + 2) * MN + idx]; s3 += part[(size_t)(z + 3) * MN + idx];
    }
    for (; z < splits; ++z) s0 += part[(size_t)z * MN + idx];
    float s = (s0 + s1) + (s2 + s3);
    if (bias) s += bias[n];
    float* dst = C + (size_t)m * ldc + n;
    if (beta != 0.f) s += beta * *dst;
    *dst = s;
}

}  // namespace

// Same contract as dep_gemm_internal (gemm.hip); `splits` is decided by the caller's shared heuristic.
int dep_gemm_bf16x3_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                           int ldb, float* C, int ldc, const float* bias, float beta, int seq_T, int shiftB,
                           int splits, int kchunk, float* part, bool vec, hipStream_t s) {
    static int abl = -1, persist = -1, bm256 = -1;
    if (abl < 0) { const char* e = getenv("DEP_GEMM_ABLATE"); abl = e ? atoi(e) : 0; }
    if (persist < 0) { const char* e = getenv("DEP_GEMM_PERSIST"); persist = e ? atoi(e) : 768; if (persist < 8) persist = 8; persist = persist / 8 * 8; }
    if (bm256 < 0) { const char* e = getenv("DEP_GEMM_BM"); bm256 = (e && atoi(e) == 128) ? 0 : 1; }
    // measured at cfg2: 256-row tiles win 8-10 % on the NN (dX) and TN (dW) forms, lose 6 % on the short-K NT projection
    const bool big = bm256 && M >= 512 && !(!transA && transB);
    const int BMT = big ? 256 : 128;
    GemmP p{M, N, K, A, lda, B, ldb, C, ldc, bias, beta, seq_T, shiftB, kchunk, splits, part, dep_cdiv(N, BN), dep_cdiv(M, BMT), abl};
    // persistent launch: at most `persist` workgroups (a multiple of 8: one share per XCD), each walks a list of tiles
    const int ntiles = p.gx * p.gy * splits;
    const int cap = big ? persist * 2 / 3 : persist;              // 2 resident workgroups per CU with 256-row tiles, 3 otherwise
    dim3 g(ntiles < cap ? (ntiles + 7) / 8 * 8 : cap / 8 * 8);
#define LAUNCH(TA, TB)                                                                               \
    do {                                                                                             \
        if (big) { if (vec) hipLaunchKernelGGL((gemm_bf16x3<TA, TB, true, 256>), g, dim3(NT), 0, s, p);      \
                   else     hipLaunchKernelGGL((gemm_bf16x3<TA, TB, false, 256>), g, dim3(NT), 0, s, p); }   \
        else     { if (vec) hipLaunchKernelGGL((gemm_bf16x3<TA, TB, true, 128>), g, dim3(NT), 0, s, p);      \
                   else     hipLaunchKernelGGL((gemm_bf16x3<TA, TB, false, 128>), g, dim3(NT), 0, s, p); }   \
    } while (0)
    if (!transA && transB) LAUNCH(false, true);
    else if (!transA && !transB) LAUNCH(false, false);
    else LAUNCH(true, false);
#undef LAUNCH
    DEP_CHECK_LAUNCH();
    if (splits > 1) {
        const long n = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce2, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, part, splits, M, N, C, ldc, bias, beta);
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}
