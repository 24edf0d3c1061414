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
#include <type_traits>
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
    const unsigned* only_if;    // run only if this device word is non-zero (dep_gemm_set_predicate), or nullptr
    int xcd_lo, xcd_n;          // experiment (dep_gemm_set_xcds): only the workgroups of XCDs [xcd_lo, xcd_lo + xcd_n) work; 0, 8 = all
    int skip_at, skip_by;       // A stored MN-contiguous (transA): logical column m lives at m + (m >= skip_at ? skip_by : 0); 0, 0 = off
};

// Operand storage formats (round 4).  FMT_F32: fp32 values, split while staging (above).  FMT_PK: the producer already stored the
// (hi, lo) bf16 planes the staging would form, in place of the fp32 array (same 4 bytes per element, same row stride): physical
// rows 2j / 2j+1 of the (rows x cols) array hold, per column c,
//     row 2j   : bf16hi(x[2j][c]) | bf16hi(x[2j+1][c]) << 16          row 2j+1 : bf16lo(x[2j][c]) | bf16lo(x[2j+1][c]) << 16
// i.e. k-pairs packed the way the LDS planes want them when the ROW index is the contraction index (TN forms: zero VALU per
// element, the four loaded rows of a thread ARE hi01, lo01, hi23, lo23); when the row index is M (NN / NT A operand) a thread
// loads both rows of a pair and separates the halves with one v_perm_b32 per element.  Bit-identical to the on-the-fly split.
// bf16-STORAGE mode (dep_set_gemm_mode(3), single products only, never the parity path):
//   FMT_PKH : the PK image with ONLY its hi rows written (rows 2j: bf16 pairs of logical rows 2j, 2j+1; rows 2j+1 unused) -- the gate
//             gradients as bf16 with the k-pairs already packed: half the operand bytes, nothing to convert
//   FMT_BF16: a plain row-major bf16 array (ld counted in bf16 elements) -- the hidden sequences y / dropout(y) as the B operand of
//             the TN contractions (rows may be shifted: dW_hh)
enum { FMT_F32 = 0, FMT_PK = 1, FMT_PKH = 2, FMT_BF16 = 3 };


// Operand tile of ROWS (128 or 256) rows x 32 k, 256 threads: a = tid&7, bq = tid>>3.
//   !TR (K-contiguous rows): r[i]      = row (mn0 + bq + 32 i),            k  = k0 + a*4 + 0..3     i < ROWS/32
//    TR (MN-contiguous)    : r[4jj + i] = k row (k0 + a*4 + i),           mn = mn0 + (bq + 32 jj)*4 + 0..3
template <bool TR, bool VEC, int ROWS, int FMT = FMT_F32>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int mn0, int MN, int k0, int Kend,
                                          int tid, float (&r)[ROWS / 32][4], int seqT, int shift, int skip_at = 0, int skip_by = 0) {
    const int a = tid & 7, bq = tid >> 3;
    if (!TR) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            // FMT_PK: slots 2j / 2j+1 are the hi-pair / lo-pair rows of logical rows (R, R+1), R = mn0 + 2 bq + 64 j  (FMT_PKH: the hi row only)
            if (FMT == FMT_PKH && (i & 1)) continue;
            const int mn = (FMT == FMT_PK || FMT == FMT_PKH) ? mn0 + 2 * bq + 64 * (i >> 1) + (i & 1) : mn0 + bq + 32 * i, k = k0 + a * 4;
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
                if (FMT == FMT_PKH && (i & 1)) continue;      // the lo rows are not there
                const int k = k0 + a * 4 + i, mn = mn0 + (bq + 32 * jj) * 4;
                bool ok = k < Kend;
                if (seqT > 0) { const int tt = k % seqT + shift; ok = ok && tt >= 0 && tt < seqT; }
                // column skip: skip_at is a multiple of the tile's 4-column pieces, so a piece never straddles it
                const float* src = P + ((long)k + shift) * ld + mn + ((skip_by && mn >= skip_at) ? skip_by : 0);
                float (&rr)[4] = r[jj * 4 + i];
                if (FMT == FMT_BF16) {                        // four bf16 of row k: 8 bytes -> rr[0], rr[1]
                    float2 v = {0.f, 0.f};
                    if (ok && mn < MN) v = *reinterpret_cast<const float2*>(reinterpret_cast<const unsigned short*>(P) + ((long)k + shift) * ld + mn);
                    rr[0] = v.x; rr[1] = v.y;
                } else if (VEC) {
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
// x - bf16(x) for both halves of a packed pair in ONE instruction each: v_dot2c_f32_bf16 d, a, b does d += a.lo*b.lo + a.hi*b.hi
// on packed bf16 operands, so with b = {-1, 0} / {0, -1} and d = x it leaves x - hi exactly (the difference is representable:
// hi is x rounded to 8 significant bits; checked against integer arithmetic on the GPU, tools/micro/t_dot2c).  That replaces
// the shift / mask / two subtractions per pair: 4 VALU per pair instead of 6 -- these kernels are VALU-issue bound (the
// conversions share the SIMD's issue port with the MFMAs), not matrix-pipe bound.
// The instruction is issued from inline asm, so hipcc's hazard recognizer cannot see it: on gfx90a+ a DOT instruction's VGPR
// result needs 3 wait states before a DIFFERENT VALU opcode reads or overwrites it (LLVM GCNHazardRecognizer:
// DotWriteDifferentVALURead / ...Write = 3; without them the conversions below read stale registers -- seen as O(1) errors).
// All four residuals of a quad are formed in one block that ends with the wait states.
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

// hi plane only (single-product mode, see gemm_bf16x3's TERMS): two conversions per quad, nothing else
__device__ __forceinline__ void hi4(f32x4 x, bf16x4& hi) {
    const f32x2v v01 = {x[0], x[1]}, v23 = {x[2], x[3]};
    const u32x2v h = {__builtin_bit_cast(unsigned, __builtin_convertvector(v01, bf16x2v)),
                      __builtin_bit_cast(unsigned, __builtin_convertvector(v23, bf16x2v))};
    hi = __builtin_bit_cast(bf16x4, h);
}

// LDS images Sh/Sl: [ROWS rows (m or n)][LDK] bf16, k contiguous
template <bool TR, int ROWS, int TERMS = 3, int FMT = FMT_F32>
__device__ __forceinline__ void store_tile(__bf16* Sh, __bf16* Sl, int tid, const float (&r)[ROWS / 32][4]) {
    const int a = tid & 7, bq = tid >> 3;
    if constexpr (FMT == FMT_BF16) {
        static_assert(TR && TERMS == 1, "FMT_BF16: MN-contiguous operand of the single-product kernel");
        // rows k0 + 4a + i (i < 4) of four columns, two bf16 per word: column e sits in half e & 1 of word e >> 1
#pragma unroll
        for (int jj = 0; jj < ROWS / 128; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
                const unsigned w0 = __float_as_uint(r[jj * 4 + 0][e >> 1]), w1 = __float_as_uint(r[jj * 4 + 1][e >> 1]);
                const unsigned w2 = __float_as_uint(r[jj * 4 + 2][e >> 1]), w3 = __float_as_uint(r[jj * 4 + 3][e >> 1]);
                const u32x2v h = {__builtin_amdgcn_perm(w1, w0, sel), __builtin_amdgcn_perm(w3, w2, sel)};
                *reinterpret_cast<u32x2v*>(Sh + ((bq + 32 * jj) * 4 + e) * LDK + a * 4) = h;
            }
    } else if constexpr ((FMT == FMT_PK || FMT == FMT_PKH) && TR) {
        static_assert(FMT != FMT_PKH || TERMS == 1, "FMT_PKH carries no lo planes");
        // rows k0 + 4a + {0,1,2,3} of a thread = hi(k, k+1), lo(k, k+1), hi(k+2, k+3), lo(k+2, k+3) of its four columns: nothing to compute
#pragma unroll
        for (int jj = 0; jj < ROWS / 128; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = ((bq + 32 * jj) * 4 + e) * LDK + a * 4;
                const u32x2v h = {__float_as_uint(r[jj * 4 + 0][e]), __float_as_uint(r[jj * 4 + 2][e])};
                *reinterpret_cast<u32x2v*>(Sh + o) = h;
                if constexpr (TERMS == 3) {
                    const u32x2v l = {__float_as_uint(r[jj * 4 + 1][e]), __float_as_uint(r[jj * 4 + 3][e])};
                    *reinterpret_cast<u32x2v*>(Sl + o) = l;
                }
            }
    } else if constexpr (FMT == FMT_PK || FMT == FMT_PKH) {
        static_assert(FMT != FMT_PKH || TERMS == 1, "FMT_PKH carries no lo planes");
        // slots 2j / 2j+1 = hi-pair / lo-pair rows of logical rows (R, R+1): low halves belong to R, high halves to R+1
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j) {
            const unsigned h0 = __float_as_uint(r[2 * j][0]), h1 = __float_as_uint(r[2 * j][1]), h2 = __float_as_uint(r[2 * j][2]), h3 = __float_as_uint(r[2 * j][3]);
            const int o = (2 * bq + 64 * j) * LDK + a * 4;
            const u32x2v he = {__builtin_amdgcn_perm(h1, h0, 0x05040100u), __builtin_amdgcn_perm(h3, h2, 0x05040100u)};
            const u32x2v ho = {__builtin_amdgcn_perm(h1, h0, 0x07060302u), __builtin_amdgcn_perm(h3, h2, 0x07060302u)};
            *reinterpret_cast<u32x2v*>(Sh + o) = he; *reinterpret_cast<u32x2v*>(Sh + o + LDK) = ho;
            if constexpr (TERMS == 3) {
                const unsigned l0 = __float_as_uint(r[2 * j + 1][0]), l1 = __float_as_uint(r[2 * j + 1][1]), l2 = __float_as_uint(r[2 * j + 1][2]), l3 = __float_as_uint(r[2 * j + 1][3]);
                const u32x2v le = {__builtin_amdgcn_perm(l1, l0, 0x05040100u), __builtin_amdgcn_perm(l3, l2, 0x05040100u)};
                const u32x2v lo = {__builtin_amdgcn_perm(l1, l0, 0x07060302u), __builtin_amdgcn_perm(l3, l2, 0x07060302u)};
                *reinterpret_cast<u32x2v*>(Sl + o) = le; *reinterpret_cast<u32x2v*>(Sl + o + LDK) = lo;
            }
        }
    } else if (!TR) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            f32x4 x = {r[i][0], r[i][1], r[i][2], r[i][3]};
            const int o = (bq + 32 * i) * LDK + a * 4;
            bf16x4 hi, lo;
            if constexpr (TERMS == 3) { split4(x, hi, lo); *reinterpret_cast<bf16x4*>(Sl + o) = lo; } else hi4(x, hi);
            *reinterpret_cast<bf16x4*>(Sh + o) = hi;
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < ROWS / 128; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 x = {r[jj * 4 + 0][e], r[jj * 4 + 1][e], r[jj * 4 + 2][e], r[jj * 4 + 3][e]};
                const int o = ((bq + 32 * jj) * 4 + e) * LDK + a * 4;
                bf16x4 hi, lo;
                if constexpr (TERMS == 3) { split4(x, hi, lo); *reinterpret_cast<bf16x4*>(Sl + o) = lo; } else hi4(x, hi);
                *reinterpret_cast<bf16x4*>(Sh + o) = hi;
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
// TERMS = 3: the split-precision product (default).  TERMS = 1: a_hi * b_hi only -- plain bf16 products with fp32 accumulation
// (dep_set_gemm_mode(2), the "bf16" throughput mode of BASELINE configs[1]: a third of the MFMAs, no lo planes; relative error
// per product ~4e-3, so it is a separately labelled mode with its own tolerance, never the parity path).
// (the kernel's whole body as a device function of (parameters, block id, block count): gemm_bf16x3_pair below runs TWO contractions
// in one launch by handing alternate workgroup slots to either)
template <bool TA, bool TB, bool VEC, int BMT, int TERMS = 3, int FA = FMT_F32, int FB = FMT_F32>
__device__ __forceinline__ void gemm_bf16x3_walk(const GemmP& p, const int block_id, const int block_count) {
    if (p.only_if && *p.only_if == 0) return;
    constexpr bool A_TR = TA, B_TR = !TB;
    constexpr int MI = BMT / 64;
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * (BMT + BN) * LDK];
    __bf16* Ah = smem; __bf16* Al = smem + BMT * LDK; __bf16* Bh = smem + 2 * BMT * LDK; __bf16* Bl = Bh + BN * LDK;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int ntiles = p.gx * p.gy * p.splits;
    const int x8 = block_id & 7, slot = block_id >> 3, slots = block_count >> 3;
    if (x8 < p.xcd_lo || x8 >= p.xcd_lo + p.xcd_n) return;
    const int xcd = x8 - p.xcd_lo;
    const int q = ntiles / p.xcd_n, r = ntiles % p.xcd_n;
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
    load_tile<A_TR, VEC, BMT, FA>(p.A, p.lda, m0, p.M, kbeg, kend, tid, ra, 0, 0, p.skip_at, p.skip_by);
    load_tile<B_TR, VEC, BN, FB>(p.B, p.ldb, n0, p.N, kbeg, kend, tid, rb, p.seqT, p.shiftB);

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
                store_tile<A_TR, BMT, TERMS, FA>(Ah, Al, tid, ra);
                store_tile<B_TR, BN, TERMS, FB>(Bh, Bl, tid, rb);
            }
            __syncthreads();
            if (!(p.ablate & 4)) {
                if (k0 + BK < kend) {
                    load_tile<A_TR, VEC, BMT, FA>(p.A, p.lda, m0, p.M, k0 + BK, kend, tid, ra, 0, 0, p.skip_at, p.skip_by);
                    load_tile<B_TR, VEC, BN, FB>(p.B, p.ldb, n0, p.N, k0 + BK, kend, tid, rb, p.seqT, p.shiftB);
                } else if (has_next) {       // first k-tile of the NEXT output tile
                    load_tile<A_TR, VEC, BMT, FA>(p.A, p.lda, nm0, p.M, nkb, nke, tid, ra, 0, 0, p.skip_at, p.skip_by);
                    load_tile<B_TR, VEC, BN, FB>(p.B, p.ldb, nn0, p.N, nkb, nke, tid, rb, p.seqT, p.shiftB);
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
                    ah[i] = *reinterpret_cast<const bf16x8*>(Ah + ro);
                    if constexpr (TERMS == 3) al[i] = *reinterpret_cast<const bf16x8*>(Al + ro);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ro = (wn * 64 + j * 32 + l31) * LDK + ko;
                    bh[j] = *reinterpret_cast<const bf16x8*>(Bh + ro);
                    if constexpr (TERMS == 3) bl[j] = *reinterpret_cast<const bf16x8*>(Bl + ro);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        // operands swapped (D = B-rows x A-rows^T): a lane then owns ONE output row m = l31 and, per register
                        // quad, 4 consecutive n -> the epilogue stores 16 bytes per lane instead of 4
                        if constexpr (TERMS == 3) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                        }
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

template <bool TA, bool TB, bool VEC, int BMT, int TERMS = 3, int FA = FMT_F32, int FB = FMT_F32>
__global__ __launch_bounds__(NT, (BMT == 256 ? 2 : 3)) void gemm_bf16x3(GemmP p) {
    gemm_bf16x3_walk<TA, TB, VEC, BMT, TERMS, FA, FB>(p, blockIdx.x, gridDim.x);
}

// Round 5: dW_ih and dW_hh of a GRU layer as ONE launch.  Both are TN contractions over K = B T whose A operand is the
// sweep's 4H-wide PK gate-gradient image -- columns [dr | dz] are the SAME bytes for both, only the third block differs (dn / dn*r) -- so two
// launches fetched two thirds of A twice (PMC: 0.63 GB per launch, four launches per step).  Here the launch has twice the workgroups of one
// contraction and slot 2i / 2i+1 of an XCD walk tile list i of problem 0 / problem 1: the two workgroups that need one (K chunk, M tile)
// panel of A run next to each other in time and on one XCD, and the second one finds it in that L2.  No per-tile operand selection (round 3's
// merged-tile attempt lost more to its uniform selects than the shared fetch saved): a workgroup belongs to ONE problem for its whole life.
// Tile decomposition, K chunks and split-K order per problem are exactly those of the single launches: the results are bit-identical.
template <bool VEC, int BMT, int FA, int FB>
__global__ __launch_bounds__(NT, (BMT == 256 ? 2 : 3)) void gemm_bf16x3_tn_pair(GemmP p0, GemmP p1) {
    const int x8 = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bid = ((slot >> 1) << 3) | x8, nblk = gridDim.x >> 1;
    if (slot & 1) gemm_bf16x3_walk<true, false, VEC, BMT, 3, FA, FB>(p1, bid, nblk);
    else gemm_bf16x3_walk<true, false, VEC, BMT, 3, FA, FB>(p0, bid, nblk);
}

// both problems' split-K partials in one launch (blockIdx.y = problem)
__global__ void splitk_reduce2_pair(const float* __restrict__ part0, const float* __restrict__ part1, int splits, int M, int N,
                                    float* C0, int ldc0, float* C1, int ldc1) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * N) return;
    const float* part = blockIdx.y ? part1 : part0;
    const int m = (int)(idx / N), n = (int)(idx % N);
    const size_t MN = (size_t)M * N;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // the same four interleaved sums as splitk_reduce2: bit-identical
    int z = 0;
    for (; z + 15 < splits; z += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = part[(size_t)(z + k) * MN + idx];
#pragma unroll
        for (int k = 0; k < 16; k += 4) { s0 += v[k]; s1 += v[k + 1]; s2 += v[k + 2]; s3 += v[k + 3]; }
    }
    for (; z + 3 < splits; z += 4) {
        s0 += part[(size_t)z * MN + idx]; s1 += part[(size_t)(z + 1) * MN + idx];
        s2 += part[(size_t)(z + 2) * MN + idx]; s3 += part[(size_t)(z + 3) * MN + idx];
    }
    for (; z < splits; ++z) s0 += part[(size_t)z * MN + idx];
    const float s = (s0 + s1) + (s2 + s3);
    if (blockIdx.y) C1[(size_t)m * ldc1 + n] = s; else C0[(size_t)m * ldc0 + n] = s;
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant (round 3).  What held the kernel above at 5.5-6.3 TB/s of L2 -> CU traffic was not the feed
// (tools/micro/l2bw.hip: 30 TB/s of L2 hits, 6.3-7.0 TB/s from HBM with the same 16-byte loads) but its prefetch distance:
// a k-tile's loads are issued one MFMA phase (~1 us) before the conversion that needs them, while a loaded HBM round trip is
// 1.5-3 us -- every iteration stalled on them (3.9 us per 256x128x32 step for 0.77 us of MFMAs).
//
// One workgroup of EIGHT waves per CU, tile 256 x 128 x 32:
//   * waves 4-7, the PRODUCERS: global loads D k-tiles ahead into D register sets (16-byte loads, 48 KB per set and
//     workgroup -> up to D x 48 KB in flight per CU), fp32 -> (hi, lo) bf16 split, ds_write into the LDS stage the consumers
//     will read NEXT step;
//   * waves 0-3, the CONSUMERS: ds_read_b128 fragments + 48 v_mfma_f32_32x32x16_bf16 per step and wave (a 128 x 64 sub-tile, 128
//     accumulator registers), epilogue stores at a tile's end.
//   One LDS-only barrier per step (raw s_barrier: __syncthreads() would drain vmcnt, i.e. the producers' prefetches); the
//   two LDS stages alternate.  Each SIMD carries one consumer and one producer wave, so the conversions' VALU work and the
//   loads' address arithmetic fill issue slots beside the other wave's MFMAs instead of in front of them.
constexpr int WS_BM = 256, WS_NT = 512;
constexpr int WS_PLANE_A = WS_BM * LDK, WS_PLANE_B = BN * LDK;
constexpr int WS_STAGE = 2 * (WS_PLANE_A + WS_PLANE_B);               // bf16 elements per stage: Ah, Al, Bh, Bl
constexpr size_t WS_LDS_BYTES = (size_t)2 * WS_STAGE * sizeof(__bf16);    // 122,880 B

__device__ __forceinline__ void ws_bar() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

struct WsWalk {
    int tile, hi, slots, gx, gy, kchunk, K, seqT;
    int m0, n0, kbeg, kend, bz, k0, t0;      // t0 = k0 % seqT (position inside the sequence, for the row-shifted operand)
    bool valid;
    __device__ __forceinline__ void coords() {
        const int bx = tile % gx, by = (tile / gx) % gy; bz = tile / (gx * gy);
        m0 = by * WS_BM; n0 = bx * BN; kbeg = bz * kchunk; kend = min(K, kbeg + kchunk); k0 = kbeg;
        t0 = seqT > 0 ? kbeg % seqT : 0;
    }
    __device__ __forceinline__ void init(const GemmP& p, int first, int hi_, int slots_) {
        tile = first; hi = hi_; slots = slots_; gx = p.gx; gy = p.gy; kchunk = p.kchunk; K = p.K; seqT = p.seqT;
        valid = tile < hi;
        if (valid) coords(); else { m0 = n0 = kbeg = kend = bz = k0 = t0 = 0; }
    }
    __device__ __forceinline__ bool last_k() const { return k0 + BK >= kend; }
    __device__ __forceinline__ void advance() {
        k0 += BK;
        if (seqT > 0) { t0 += BK; while (t0 >= seqT) t0 -= seqT; }
        if (k0 >= kend) { tile += slots; valid = tile < hi; if (valid) coords(); }
    }
};

// Producer-side tile movers.  The loads are UNCONDITIONAL (addresses clamped into the operand, nothing predicated): a
// predicated load compiles to a branch plus a copy of the loaded value into the rotating register set, i.e. an
// s_waitcnt vmcnt(0) behind every single load (seen in the first version of this kernel).  Elements outside the tile's
// valid range are zeroed when the set is converted, from the coordinates remembered with the set.
struct WsTag { int mn0a, mn0b, k0, kend; bool fast; };

template <bool TR, bool VEC, int ROWS>
__device__ __forceinline__ void ws_load(const float* __restrict__ P, int ld, int mn0, int MN, int k0, int Ktot, int tid,
                                        float (&r)[ROWS / 32][4], int shift) {
    const int a = tid & 7, bq = tid >> 3;
    if (!TR) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int mn = min(mn0 + bq + 32 * i, MN - 1);
            if (VEC) {
                const int k = min(k0 + a * 4, Ktot - 4);
                const f32x4 v = *reinterpret_cast<const f32x4*>(P + (size_t)mn * ld + k);
                r[i][0] = v[0]; r[i][1] = v[1]; r[i][2] = v[2]; r[i][3] = v[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[i][e] = P[(size_t)mn * ld + min(k0 + a * 4 + e, Ktot - 1)];
            }
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < ROWS / 128; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = min(max(k0 + a * 4 + i + shift, 0), Ktot - 1);
                float (&rr)[4] = r[jj * 4 + i];
                if (VEC) {
                    const int mn = min(mn0 + (bq + 32 * jj) * 4, MN - 4);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(P + (size_t)k * ld + mn);
                    rr[0] = v[0]; rr[1] = v[1]; rr[2] = v[2]; rr[3] = v[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rr[e] = P[(size_t)k * ld + min(mn0 + (bq + 32 * jj) * 4 + e, MN - 1)];
                }
            }
    }
}

// zero what lies outside [mn0, MN) x [k0, kend) (and, for the row-shifted operand, outside its sequence), then split + store
template <bool TR, int ROWS>
__device__ __forceinline__ void ws_store(__bf16* Sh, __bf16* Sl, int tid, float (&r)[ROWS / 32][4], int mn0, int MN, int k0,
                                         int kend, int seqT, int shift) {
    const int a = tid & 7, bq = tid >> 3;
    if (!TR) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const bool okr = mn0 + bq + 32 * i < MN;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!(okr && k0 + a * 4 + e < kend)) r[i][e] = 0.f;
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < ROWS / 128; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + a * 4 + i;
                bool ok = k < kend;
                if (seqT > 0) { const int tt = k % seqT + shift; ok = ok && tt >= 0 && tt < seqT; }
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!(ok && mn0 + (bq + 32 * jj) * 4 + e < MN)) r[jj * 4 + i][e] = 0.f;
            }
    }
    store_tile<TR, ROWS>(Sh, Sl, tid, r);
}

template <bool TA, bool TB, bool VEC, int D>
__global__ __launch_bounds__(WS_NT, 1) void gemm_bf16x3_ws(GemmP p) {
    if (p.only_if && *p.only_if == 0) return;
    constexpr bool A_TR = TA, B_TR = !TB;
    extern __shared__ __attribute__((aligned(16))) __bf16 ws_smem[];

    const int ntiles = p.gx * p.gy * p.splits;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int q = ntiles / 8, r = ntiles % 8;
    const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int hi = lo + (xcd < r ? q + 1 : q);
    if (lo + slot >= hi) return;
    // number of (tile, k-tile) steps of this workgroup: identical in both roles (they meet at one barrier per step);
    // rounded up to a multiple of D so that the producers' loop over the rotating register sets has no partial round
    int G = 0;
    for (int t = lo + slot; t < hi; t += slots) {
        const int bz = t / (p.gx * p.gy), kb = bz * p.kchunk, ke = min(p.K, kb + p.kchunk);
        G += (ke - kb + BK - 1) / BK;
    }
    const int Gr = (G + D - 1) / D * D;

    if (threadIdx.x >= 256) {
        // ------------------------------------------------------------------ producers
        const int tid = threadIdx.x - 256;
        WsWalk wl; wl.init(p, lo + slot, hi, slots);
        float ra[D][WS_BM / 32][4], rb[D][BN / 32][4];
        WsTag tag[D];
        // per-thread byte offsets inside an operand tile: loop invariants, so an interior step's loads are
        // global_load_dwordx4 v, v_off, s[base] with a scalar base that moves per step -- no VALU per load
        const int la = tid & 7, lbq = tid >> 3;
        unsigned offA[WS_BM / 32], offB[BN / 32];
#pragma unroll
        for (int i = 0; i < WS_BM / 32; ++i)
            offA[i] = A_TR ? (unsigned)(((la * 4 + (i & 3)) * p.lda + (lbq + 32 * (i >> 2)) * 4) * 4)
                           : (unsigned)(((lbq + 32 * i) * p.lda + la * 4) * 4);
#pragma unroll
        for (int i = 0; i < BN / 32; ++i)
            offB[i] = B_TR ? (unsigned)(((la * 4 + (i & 3)) * p.ldb + (lbq + 32 * (i >> 2)) * 4) * 4)
                           : (unsigned)(((lbq + 32 * i) * p.ldb + la * 4) * 4);
        typedef const char __attribute__((address_space(1)))* gcp;          // global address space: global_load, not flat_load
        typedef const f32x4 __attribute__((address_space(1)))* gv4p;
        auto sgpr_ptr = [](const char* q) {          // pin a uniform pointer into SGPRs: the loads below then take the saddr form
            const unsigned long long u = reinterpret_cast<unsigned long long>(q);
            const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)u), h = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            return (gcp)(((unsigned long long)h << 32) | l);
        };
        auto issue = [&](auto SET) {
            constexpr int S = decltype(SET)::value;
            // interior step: whole tile inside the operands, a full k-tile, no sequence boundary of the row-shifted operand
            bool fast = VEC && wl.valid && wl.m0 + WS_BM <= p.M && wl.n0 + BN <= p.N && wl.k0 + BK <= wl.kend;
            if (p.seqT > 0) fast = fast && wl.t0 + min(p.shiftB, 0) >= 0 && wl.t0 + BK - 1 + max(p.shiftB, 0) < p.seqT;
            // past the last step the walker stays on its last coordinates: the loads stay valid, nobody consumes them
            tag[S] = WsTag{wl.m0, wl.n0, wl.k0, wl.valid ? wl.kend : 0, fast};
            if constexpr (VEC) {
                // ONE load stream for both kinds of step (scalar base + 32-bit lane offset); only the offsets differ.  Two
                // separate streams would meet in copies of the loaded registers, i.e. in waits for the loads just issued.
                const char* ba; const char* bb;
                unsigned va[WS_BM / 32], vb[BN / 32];
                if (fast) {
                    ba = reinterpret_cast<const char*>(p.A) + (A_TR ? ((size_t)wl.k0 * p.lda + wl.m0) : ((size_t)wl.m0 * p.lda + wl.k0)) * 4;
                    bb = reinterpret_cast<const char*>(p.B) + (B_TR ? ((size_t)(wl.k0 + p.shiftB) * p.ldb + wl.n0) : ((size_t)wl.n0 * p.ldb + wl.k0)) * 4;
#pragma unroll
                    for (int i = 0; i < WS_BM / 32; ++i) va[i] = offA[i];
#pragma unroll
                    for (int i = 0; i < BN / 32; ++i) vb[i] = offB[i];
                } else {
                    ba = reinterpret_cast<const char*>(p.A); bb = reinterpret_cast<const char*>(p.B);
#pragma unroll
                    for (int i = 0; i < WS_BM / 32; ++i) {
                        if (A_TR) {
                            const int k = min(max(wl.k0 + la * 4 + (i & 3), 0), p.K - 1), mn = min(wl.m0 + (lbq + 32 * (i >> 2)) * 4, p.M - 4);
                            va[i] = (unsigned)(k * p.lda + mn) * 4u;
                        } else {
                            const int mn = min(wl.m0 + lbq + 32 * i, p.M - 1), k = min(wl.k0 + la * 4, p.K - 4);
                            va[i] = (unsigned)(mn * p.lda + k) * 4u;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < BN / 32; ++i) {
                        if (B_TR) {
                            const int k = min(max(wl.k0 + la * 4 + (i & 3) + p.shiftB, 0), p.K - 1), mn = min(wl.n0 + (lbq + 32 * (i >> 2)) * 4, p.N - 4);
                            vb[i] = (unsigned)(k * p.ldb + mn) * 4u;
                        } else {
                            const int mn = min(wl.n0 + lbq + 32 * i, p.N - 1), k = min(wl.k0 + la * 4, p.K - 4);
                            vb[i] = (unsigned)(mn * p.ldb + k) * 4u;
                        }
                    }
                }
                const gcp ga = sgpr_ptr(ba), gb = sgpr_ptr(bb);
#pragma unroll
                for (int i = 0; i < WS_BM / 32; ++i) {
                    const f32x4 v = *(gv4p)(ga + va[i]);
                    ra[S][i][0] = v[0]; ra[S][i][1] = v[1]; ra[S][i][2] = v[2]; ra[S][i][3] = v[3];
                }
#pragma unroll
                for (int i = 0; i < BN / 32; ++i) {
                    const f32x4 v = *(gv4p)(gb + vb[i]);
                    rb[S][i][0] = v[0]; rb[S][i][1] = v[1]; rb[S][i][2] = v[2]; rb[S][i][3] = v[3];
                }
            } else {
                ws_load<A_TR, false, WS_BM>(p.A, p.lda, wl.m0, p.M, wl.k0, p.K, tid, ra[S], 0);
                ws_load<B_TR, false, BN>(p.B, p.ldb, wl.n0, p.N, wl.k0, p.K, tid, rb[S], p.shiftB);
            }
            if (wl.valid) wl.advance();
        };
        auto stage_out = [&](auto SET, int stage) {
            constexpr int S = decltype(SET)::value;
            __bf16* base = ws_smem + stage * WS_STAGE;
            if (tag[S].fast) {
                store_tile<A_TR, WS_BM>(base, base + WS_PLANE_A, tid, ra[S]);
                store_tile<B_TR, BN>(base + 2 * WS_PLANE_A, base + 2 * WS_PLANE_A + WS_PLANE_B, tid, rb[S]);
            } else {
                ws_store<A_TR, WS_BM>(base, base + WS_PLANE_A, tid, ra[S], tag[S].mn0a, p.M, tag[S].k0, tag[S].kend, 0, 0);
                ws_store<B_TR, BN>(base + 2 * WS_PLANE_A, base + 2 * WS_PLANE_A + WS_PLANE_B, tid, rb[S], tag[S].mn0b, p.N,
                                   tag[S].k0, tag[S].kend, p.seqT, p.shiftB);
            }
        };
#define WS_ISSUE(S) issue(std::integral_constant<int, S>{})
        WS_ISSUE(0);
        if constexpr (D > 1) WS_ISSUE(1);
        if constexpr (D > 2) WS_ISSUE(2);
        if constexpr (D > 3) WS_ISSUE(3);                        // steps 0 .. D-1 in flight
        stage_out(std::integral_constant<int, 0>{}, 0);          // step 0 -> stage 0
        WS_ISSUE(0);                                             // step D
        ws_bar();
        // step g: the consumers multiply stage g & 1; convert step g+1 (register set (g+1) % D) into the other stage and
        // refill that set with step g+1+D.  (The last step converts a set nobody reads: cheaper than a conditional that
        // would make hipcc drain the load queue at the loop head.)
#define WS_STEP(S, GG) { if (!(p.ablate & 8)) stage_out(std::integral_constant<int, S>{}, ((GG) + 1) & 1); if (!(p.ablate & 4)) WS_ISSUE(S); ws_bar(); }
        for (int g = 0; g < Gr; g += D) {
            if constexpr (D == 1) { WS_STEP(0, g) }
            if constexpr (D == 2) { WS_STEP(1, g) WS_STEP(0, g + 1) }
            if constexpr (D == 3) { WS_STEP(1, g) WS_STEP(2, g + 1) WS_STEP(0, g + 2) }
            if constexpr (D == 4) { WS_STEP(1, g) WS_STEP(2, g + 1) WS_STEP(3, g + 2) WS_STEP(0, g + 3) }
        }
#undef WS_STEP
#undef WS_ISSUE
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int MI = WS_BM / 64;
    WsWalk wc; wc.init(p, lo + slot, hi, slots);
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    ws_bar();                                                    // stage 0 is ready
    for (int g = 0; g < Gr; ++g) {
        if (g >= G) { ws_bar(); continue; }                      // padding steps of the producers' last round
        const __bf16* Ah = ws_smem + (g & 1) * WS_STAGE; const __bf16* Al = Ah + WS_PLANE_A;
        const __bf16* Bh = Ah + 2 * WS_PLANE_A; const __bf16* Bl = Bh + WS_PLANE_B;
        if (!(p.ablate & 2))
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int ko = s * 16 + half * 8;
            bf16x8 ah[MI], al[MI], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int ro = (wm * (WS_BM / 2) + i * 32 + l31) * LDK + ko;
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                }
        }
        const bool fin = wc.last_k();
        const int m0 = wc.m0, n0 = wc.n0, bz = wc.bz;
        wc.advance();
        ws_bar();                                                // the producers go on with the next stage while a finished tile is stored
        if (fin) {
            const bool split = p.part != nullptr;
            float* outp = split ? p.part + (size_t)bz * p.M * p.N : p.C;
            const int ldo = split ? p.N : p.ldc;
            const bool v4 = ((ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(outp) & 15) == 0);
            const bool addb = !split && p.bias;
            const bool rmw = !split && p.beta != 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * (WS_BM / 2) + i * 32 + l31;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int n = n0 + wn * 64 + j * 32 + 8 * gq + 4 * half;
                        f32x4 v = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
                        acc[i][j][4 * gq] = 0.f; acc[i][j][4 * gq + 1] = 0.f; acc[i][j][4 * gq + 2] = 0.f; acc[i][j][4 * gq + 3] = 0.f;
                        if (m >= p.M || n >= p.N) continue;
                        float* dst = outp + (size_t)m * ldo + n;
                        if (v4 && n + 3 < p.N) {
                            if (addb) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                            if (rmw) v += p.beta * *reinterpret_cast<const f32x4*>(dst);
                            *reinterpret_cast<f32x4*>(dst) = v;
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
        }
    }
}

__global__ void splitk_reduce2(const unsigned* only_if, const float* __restrict__ part, int splits, int M, int N, float* C, int ldc,
                               const float* bias, float beta) {
    if (only_if && *only_if == 0) return;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    // four independent partial sums: the loads of consecutive splits overlap instead of forming one dependent chain
    const size_t MN = (size_t)M * N;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 15 < splits; z += 16) {                      // sixteen loads in flight; the same four interleaved sums as below
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = part[(size_t)(z + k) * MN + idx];
#pragma unroll
        for (int k = 0; k < 16; k += 4) { s0 += v[k]; s1 += v[k + 1]; s2 += v[k + 2]; s3 += v[k + 3]; }
    }
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

// experiment hook: confine the persistent kernel's working workgroups to a range of XCDs (per calling thread)
static thread_local int g_xcd_lo = 0, g_xcd_n = 8;
static thread_local int g_skip_at = 0, g_skip_by = 0;
void dep_gemm_set_a_colskip(int at, int by) { g_skip_at = at; g_skip_by = by; }
static thread_local int g_fmt_a = FMT_F32, g_fmt_b = FMT_F32;
void dep_gemm_set_operand_formats(int fmt_a, int fmt_b) { g_fmt_a = fmt_a; g_fmt_b = fmt_b; }
bool dep_gemm_pk_pending() { return g_fmt_a != FMT_F32 || g_fmt_b != FMT_F32; }
extern "C" int dep_gemm_set_xcds(int lo, int n) { if (lo < 0 || n < 1 || lo + n > 8) return DEP_ERR_ARG; g_xcd_lo = lo; g_xcd_n = n; return DEP_OK; }

// Same contract as dep_gemm_internal (gemm.hip); `splits` is decided by the caller's shared heuristic.
int dep_gemm_bf16x3_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                           int ldb, float* C, int ldc, const float* bias, float beta, int seq_T, int shiftB,
                           int splits, int kchunk, float* part, bool vec, hipStream_t s, int terms) {
    static int abl = -1, persist = -1, bm256 = -1, wsd = -2;
    if (abl < 0) { const char* e = getenv("DEP_GEMM_ABLATE"); abl = e ? atoi(e) : 0; }
    if (wsd == -2) { const char* e = getenv("DEP_GEMM_WS"); wsd = e ? atoi(e) : 0; if (wsd < 0 || wsd > 4) wsd = 0; }
    // (32-bit lane offsets: every operand must span less than 4 GB)
    const size_t spanA = (size_t)(transA ? K : M) * lda * 4, spanB = (size_t)(transB ? N : K) * ldb * 4;
    const int fa = g_fmt_a, fb = g_fmt_b;
    if (fa != FMT_F32 || fb != FMT_F32) {
        // pre-split operands: vector loads, three-term products; a PK operand pairs ROWS, so its row count must be even, the
        // contraction index (TN) must start on even rows (k-chunks are multiples of 32) and a row-shifted operand cannot be PK
        if (fa == FMT_PKH || fb == FMT_BF16) {
            // bf16-storage mode: single products; A = PKH gate gradients (TN / NN), B = fp32 or (TN only) a row-major bf16 sequence
            DEP_CHECK_ARG(vec && terms == 1 && fa == FMT_PKH && !(!transA && transB) && (fb == FMT_F32 || (fb == FMT_BF16 && transA)));
        } else {
            DEP_CHECK_ARG(vec && terms == 3 && fa == FMT_PK && (fb == FMT_F32 || (fb == FMT_PK && transA && !transB && shiftB == 0)));
        }
        DEP_CHECK_ARG(transA ? (K % 2 == 0 && kchunk % 2 == 0) : (M % 2 == 0));
    }
    if (wsd > 0 && fa == FMT_F32 && fb == FMT_F32 && terms == 3 && M >= 256 && !(transA && g_skip_by) && spanA < (1ull << 32) && spanB < (1ull << 32)) {
        // wave-specialised kernel: one 8-wave workgroup per CU, 256 x 128 tiles, `wsd` register sets of prefetch
        GemmP p{M, N, K, A, lda, B, ldb, C, ldc, bias, beta, seq_T, shiftB, kchunk, splits, part, dep_cdiv(N, BN), dep_cdiv(M, WS_BM), abl, dep_gemm_predicate(), 0, 8, 0, 0};
        const int ntiles = p.gx * p.gy * splits;
        int ncu = 256;
        { static int cus = -1; if (cus < 0) { hipDeviceProp_t pr; int dv = 0; cus = (hipGetDevice(&dv) == hipSuccess && hipGetDeviceProperties(&pr, dv) == hipSuccess) ? pr.multiProcessorCount : 256; } ncu = cus / 8 * 8; if (ncu < 8) ncu = 8; }
        dim3 g(ntiles < ncu ? (ntiles + 7) / 8 * 8 : ncu);
        static bool attr = false;
#define WS_ATTR1(TA, TB, V, DD) (void)hipFuncSetAttribute((const void*)gemm_bf16x3_ws<TA, TB, V, DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WS_LDS_BYTES)
#define WS_ATTR(DD) WS_ATTR1(false, true, true, DD); WS_ATTR1(false, true, false, DD); WS_ATTR1(false, false, true, DD); \
                    WS_ATTR1(false, false, false, DD); WS_ATTR1(true, false, true, DD); WS_ATTR1(true, false, false, DD)
        if (!attr) { WS_ATTR(2); WS_ATTR(3); WS_ATTR(4); attr = true; }
#undef WS_ATTR
#undef WS_ATTR1
#define WS_L1(TA, TB, DD) do { if (vec) DEP_LAUNCH((gemm_bf16x3_ws<TA, TB, true, DD>), g, dim3(WS_NT), WS_LDS_BYTES, s, p); \
                               else     DEP_LAUNCH((gemm_bf16x3_ws<TA, TB, false, DD>), g, dim3(WS_NT), WS_LDS_BYTES, s, p); } while (0)
#define WS_L(TA, TB) do { if (wsd == 2) WS_L1(TA, TB, 2); else if (wsd == 4) WS_L1(TA, TB, 4); else WS_L1(TA, TB, 3); } while (0)
        if (!transA && transB) WS_L(false, true);
        else if (!transA && !transB) WS_L(false, false);
        else WS_L(true, false);
#undef WS_L
#undef WS_L1
        DEP_CHECK_LAUNCH();
        if (splits > 1) {
            const long n = (long)M * N;
            DEP_LAUNCH(splitk_reduce2, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, dep_gemm_predicate(), part, splits, M, N, C, ldc, bias, beta);
            DEP_CHECK_LAUNCH();
        }
        return DEP_OK;
    }
    if (abl < 0) { const char* e = getenv("DEP_GEMM_ABLATE"); abl = e ? atoi(e) : 0; }
    if (persist < 0) { const char* e = getenv("DEP_GEMM_PERSIST"); persist = e ? atoi(e) : 768; if (persist < 8) persist = 8; persist = persist / 8 * 8; }
    if (bm256 < 0) { const char* e = getenv("DEP_GEMM_BM"); bm256 = (e && atoi(e) == 128) ? 0 : 1; }
    // measured at cfg2: 256-row tiles win 8-10 % on the NN (dX) and TN (dW) forms, lose 6 % on the short-K NT projection
    // (round 4: long-K projections -- cfg3's F = 1024 layer -- take the 256-row tile too; DEP_GEMM_NT256=0 restores 128 rows for every NT call)
    static int nt256 = -1;
    if (nt256 < 0) { const char* e = getenv("DEP_GEMM_NT256"); nt256 = (e && e[0] == '0') ? 0 : 1; }
    // (a pre-split A operand in the NT form has only the 128-row instantiation: the tile grid must follow -- ADVICE r4)
    const bool big = bm256 && M >= 512 && (!(!transA && transB) || (nt256 && K >= 512)) && !(fa != FMT_F32 && !transA && transB);
    const int BMT = big ? 256 : 128;
    GemmP p{M, N, K, A, lda, B, ldb, C, ldc, bias, beta, seq_T, shiftB, kchunk, splits, part, dep_cdiv(N, BN), dep_cdiv(M, BMT), abl, dep_gemm_predicate(), g_xcd_lo, g_xcd_n, transA ? g_skip_at : 0, transA ? g_skip_by : 0};
    // persistent launch: at most `persist` workgroups (a multiple of 8: one share per XCD), each walks a list of tiles
    const int ntiles = p.gx * p.gy * splits;
    const int cap = big ? persist * 2 / 3 : persist;              // 2 resident workgroups per CU with 256-row tiles, 3 otherwise
    const int per_xcd = (ntiles + g_xcd_n - 1) / g_xcd_n;
    dim3 g((per_xcd < cap / 8 ? per_xcd : cap / 8) * 8);
#define LAUNCH1(TA, TB, TERMS)                                                                       \
    do {                                                                                             \
        if (big) { if (vec) DEP_LAUNCH((gemm_bf16x3<TA, TB, true, 256, TERMS>), g, dim3(NT), 0, s, p);      \
                   else     DEP_LAUNCH((gemm_bf16x3<TA, TB, false, 256, TERMS>), g, dim3(NT), 0, s, p); }   \
        else     { if (vec) DEP_LAUNCH((gemm_bf16x3<TA, TB, true, 128, TERMS>), g, dim3(NT), 0, s, p);      \
                   else     DEP_LAUNCH((gemm_bf16x3<TA, TB, false, 128, TERMS>), g, dim3(NT), 0, s, p); }   \
    } while (0)
#define LAUNCH(TA, TB) do { if (terms == 1) LAUNCH1(TA, TB, 1); else LAUNCH1(TA, TB, 3); } while (0)
#define LAUNCH_PK(TA, TB, BM, FA_, FB_) DEP_LAUNCH((gemm_bf16x3<TA, TB, true, BM, 3, FA_, FB_>), g, dim3(NT), 0, s, p)
#define LAUNCH_H(TA, TB, BM, FB_) DEP_LAUNCH((gemm_bf16x3<TA, TB, true, BM, 1, FMT_PKH, FB_>), g, dim3(NT), 0, s, p)
    if (fa == FMT_PKH) {
        if (transA) {
            if (big) { if (fb == FMT_BF16) LAUNCH_H(true, false, 256, FMT_BF16); else LAUNCH_H(true, false, 256, FMT_F32); }
            else     { if (fb == FMT_BF16) LAUNCH_H(true, false, 128, FMT_BF16); else LAUNCH_H(true, false, 128, FMT_F32); }
        } else {
            if (big) LAUNCH_H(false, false, 256, FMT_F32); else LAUNCH_H(false, false, 128, FMT_F32);
        }
    }
    else if (fa == FMT_PK) {
        if (transA) {
            if (big) { if (fb == FMT_PK) LAUNCH_PK(true, false, 256, FMT_PK, FMT_PK); else LAUNCH_PK(true, false, 256, FMT_PK, FMT_F32); }
            else     { if (fb == FMT_PK) LAUNCH_PK(true, false, 128, FMT_PK, FMT_PK); else LAUNCH_PK(true, false, 128, FMT_PK, FMT_F32); }
        } else if (!transB) {
            if (big) LAUNCH_PK(false, false, 256, FMT_PK, FMT_F32); else LAUNCH_PK(false, false, 128, FMT_PK, FMT_F32);
        } else {
            LAUNCH_PK(false, true, 128, FMT_PK, FMT_F32);
        }
    }
    else if (!transA && transB) LAUNCH(false, true);
    else if (!transA && !transB) LAUNCH(false, false);
    else LAUNCH(true, false);
#undef LAUNCH_PK
#undef LAUNCH_H
#undef LAUNCH
#undef LAUNCH1
    DEP_CHECK_LAUNCH();
    if (splits > 1) {
        const long n = (long)M * N;
        DEP_LAUNCH(splitk_reduce2, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, dep_gemm_predicate(), part, splits, M, N, C, ldc, bias, beta);
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}

// dW_ih + dW_hh of one GRU layer in one launch (gemm_bf16x3_tn_pair above).  A = the sweep's PK gate-gradient image for both (problem 1
// reads it through the column skip), B0 = the layer's input, B1 = the layer's own output shifted one step; same (M, N, K, splits, kchunk).
// The caller (dep_gemm_tn_pair, gemm.hip) has checked formats / alignment / sizes; returns DEP_OK after enqueueing both the walk and the reduce.
int dep_gemm_bf16x3_tn_pair_launch(int M, int N, int K, const float* A0, const float* A1, int lda, int skip_at1, int skip_by1,
                                   const float* B0, int ldb0, int seq_T0, int shift0, const float* B1, int ldb1, int seq_T1, int shift1,
                                   float* C0, int ldc0, float* C1, int ldc1, int splits, int kchunk, float* part0, float* part1, hipStream_t s) {
    DEP_CHECK_ARG(g_fmt_a == FMT_PK && g_fmt_b == FMT_F32 && M >= 512 && splits > 1 && part0 && part1);
    DEP_CHECK_ARG(K % 2 == 0 && kchunk % 2 == 0 && dep_gemm_predicate() == nullptr && g_xcd_lo == 0 && g_xcd_n == 8);
    static int persist = -1;
    if (persist < 0) { const char* e = getenv("DEP_GEMM_PERSIST"); persist = e ? atoi(e) : 768; if (persist < 8) persist = 8; persist = persist / 8 * 8; }
    GemmP p0{M, N, K, A0, lda, B0, ldb0, C0, ldc0, nullptr, 0.f, seq_T0, shift0, kchunk, splits, part0, dep_cdiv(N, BN), dep_cdiv(M, 256), 0, nullptr, 0, 8, 0, 0};
    GemmP p1 = p0;
    p1.A = A1; p1.B = B1; p1.ldb = ldb1; p1.C = C1; p1.ldc = ldc1; p1.seqT = seq_T1; p1.shiftB = shift1; p1.part = part1; p1.skip_at = skip_at1; p1.skip_by = skip_by1;
    const int ntiles = p0.gx * p0.gy * splits;
    const int cap = persist * 2 / 3 / 2;                          // two resident workgroups per CU with 256-row tiles, half of the slots per problem
    const int per_xcd = (ntiles + 7) / 8;
    dim3 g((per_xcd < cap / 8 ? per_xcd : cap / 8) * 8 * 2);
    DEP_LAUNCH((gemm_bf16x3_tn_pair<true, 256, FMT_PK, FMT_F32>), g, dim3(NT), 0, s, p0, p1);
    DEP_CHECK_LAUNCH();
    const long n = (long)M * N;
    DEP_LAUNCH(splitk_reduce2_pair, dim3(dep_cdiv(n, 256), 2), dim3(256), 0, s, part0, part1, splits, M, N, C0, ldc0, C1, ldc1);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
