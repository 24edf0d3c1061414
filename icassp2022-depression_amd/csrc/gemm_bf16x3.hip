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


// Operand tile of ROWS (128 or 256) rows x 32 k, NTH threads (256; 512 in the 256 x 256-tile kernel): a = tid&7, bq = tid>>3, RPI = NTH / 8 rows
// per pass (the comments below spell the 256-thread case, RPI = 32).
//   !TR (K-contiguous rows): r[i]      = row (mn0 + bq + 32 i),            k  = k0 + a*4 + 0..3     i < ROWS/32
//    TR (MN-contiguous)    : r[4jj + i] = k row (k0 + a*4 + i),           mn = mn0 + (bq + 32 jj)*4 + 0..3
template <bool TR, bool VEC, int ROWS, int FMT = FMT_F32, int NTH = 256>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int mn0, int MN, int k0, int Kend,
                                          int tid, float (&r)[ROWS * 8 / NTH][4], int seqT, int shift, int skip_at = 0, int skip_by = 0) {
    constexpr int RPI = NTH / 8;
    const int a = tid & 7, bq = tid >> 3;
    if (!TR) {
#pragma unroll
        for (int i = 0; i < ROWS / RPI; ++i) {
            // FMT_PK: slots 2j / 2j+1 are the hi-pair / lo-pair rows of logical rows (R, R+1), R = mn0 + 2 bq + 64 j  (FMT_PKH: the hi row only)
            if (FMT == FMT_PKH && (i & 1)) continue;
            const int mn = (FMT == FMT_PK || FMT == FMT_PKH) ? mn0 + 2 * bq + 2 * RPI * (i >> 1) + (i & 1) : mn0 + bq + RPI * i, k = k0 + a * 4;
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
        for (int jj = 0; jj < ROWS / (4 * RPI); ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (FMT == FMT_PKH && (i & 1)) continue;      // the lo rows are not there
                const int k = k0 + a * 4 + i, mn = mn0 + (bq + RPI * jj) * 4;
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
template <bool TR, int ROWS, int TERMS = 3, int FMT = FMT_F32, int NTH = 256>
__device__ __forceinline__ void store_tile(__bf16* Sh, __bf16* Sl, int tid, const float (&r)[ROWS * 8 / NTH][4]) {
    constexpr int RPI = NTH / 8;
    const int a = tid & 7, bq = tid >> 3;
    if constexpr (FMT == FMT_BF16) {
        static_assert(TR && TERMS == 1, "FMT_BF16: MN-contiguous operand of the single-product kernel");
        // rows k0 + 4a + i (i < 4) of four columns, two bf16 per word: column e sits in half e & 1 of word e >> 1
#pragma unroll
        for (int jj = 0; jj < ROWS / (4 * RPI); ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
                const unsigned w0 = __float_as_uint(r[jj * 4 + 0][e >> 1]), w1 = __float_as_uint(r[jj * 4 + 1][e >> 1]);
                const unsigned w2 = __float_as_uint(r[jj * 4 + 2][e >> 1]), w3 = __float_as_uint(r[jj * 4 + 3][e >> 1]);
                const u32x2v h = {__builtin_amdgcn_perm(w1, w0, sel), __builtin_amdgcn_perm(w3, w2, sel)};
                *reinterpret_cast<u32x2v*>(Sh + ((bq + RPI * jj) * 4 + e) * LDK + a * 4) = h;
            }
    } else if constexpr ((FMT == FMT_PK || FMT == FMT_PKH) && TR) {
        static_assert(FMT != FMT_PKH || TERMS == 1, "FMT_PKH carries no lo planes");
        // rows k0 + 4a + {0,1,2,3} of a thread = hi(k, k+1), lo(k, k+1), hi(k+2, k+3), lo(k+2, k+3) of its four columns: nothing to compute
#pragma unroll
        for (int jj = 0; jj < ROWS / (4 * RPI); ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = ((bq + RPI * jj) * 4 + e) * LDK + a * 4;
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
        for (int j = 0; j < ROWS / (2 * RPI); ++j) {
            const unsigned h0 = __float_as_uint(r[2 * j][0]), h1 = __float_as_uint(r[2 * j][1]), h2 = __float_as_uint(r[2 * j][2]), h3 = __float_as_uint(r[2 * j][3]);
            const int o = (2 * bq + 2 * RPI * j) * LDK + a * 4;
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
        for (int i = 0; i < ROWS / RPI; ++i) {
            f32x4 x = {r[i][0], r[i][1], r[i][2], r[i][3]};
            const int o = (bq + RPI * i) * LDK + a * 4;
            bf16x4 hi, lo;
            if constexpr (TERMS == 3) { split4(x, hi, lo); *reinterpret_cast<bf16x4*>(Sl + o) = lo; } else hi4(x, hi);
            *reinterpret_cast<bf16x4*>(Sh + o) = hi;
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < ROWS / (4 * RPI); ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 x = {r[jj * 4 + 0][e], r[jj * 4 + 1][e], r[jj * 4 + 2][e], r[jj * 4 + 3][e]};
                const int o = ((bq + RPI * jj) * 4 + e) * LDK + a * 4;
                bf16x4 hi, lo;
                if constexpr (TERMS == 3) { split4(x, hi, lo); *reinterpret_cast<bf16x4*>(Sl + o) = lo; } else hi4(x, hi);
                *reinterpret_cast<bf16x4*>(Sh + o) = hi;
            }
    }
}

// (Round 6: a 256 x 256-tile, eight-wave, double-buffered form of this kernel -- lock-step and ping-pong builds -- measured the same time / slower:
// profiles/r06_s4_gemm_256x256_tile_ab.txt, r06_s6_gemm_pingpong_ab.txt, DESIGN 4.7 -- and was deleted again; the NTH parameter of load_tile / store_tile above is what is left of it.)
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

// Round 6: the weight-gradient contractions fed by LDS-DMA.   C[m][n] = sum_k A[k][m] B[k][n],  A = the sweeps' PK gate-gradient
// image (M contiguous), B = fp32 rows (N contiguous, optionally the layer's own output shifted one step along its sequence).
// The kernel above stages operands global -> registers -> (convert) -> LDS with one k-tile of register prefetch; its workgroups spend most of a k-tile
// waiting for that one round trip (the loads alone cost 0.4 of cfg3's 1.06 ms).  Here
//   * a 256 x 256 output tile, eight waves (2 x 4, each 128 x 64): a third fewer operand bytes per flop through L2 -> LDS than 256 x 128;
//   * stages of 16 k-rows; every stage row of either operand is ONE `buffer_load_dwordx4 ... lds` (1 KiB, lane-linear): no staging registers, no
//     conversion pass, no ds_write, and -- no destination registers -- no compiler-placed vmcnt wait: four stages in LDS (128 KiB), three in flight,
//     counted `s_waitcnt vmcnt(8)` (never 0 in the loop), ONE workgroup barrier per slot;
//   * A's fragments are four k-pair words of four consecutive PK rows (ds_read2_b32, conflict-free: a lane group reads 32 consecutive words);
//     B's are eight fp32 of eight consecutive rows, split into (hi, lo) by the consuming wave exactly as the staging pass of the kernel above does;
//   * the two wave groups (waves 0-3 / 4-7: one of each per SIMD) run ONE BARRIER apart: while one issues its 24 MFMAs the other reads and splits
//     its fragments, so a SIMD's matrix pipe always has a wave feeding it.
// Split-K chunks, the order of k inside a chunk, operand roles, the three products per 16 k and their order are those of the kernel above: the
// partial sums -- and after the same reduce the weight gradients -- are bit-identical (tests/golden/device_bits.json).
// A pair (dW_ih / dW_hh of a GRU layer: same A) runs as neighbouring slots of one XCD, like gemm_bf16x3_tn_pair.
// Measured (tools/micro/gemm_tn_dma.hip, profiles/r06_s14_*): cfg2 pair 0.345 ms (shipped 0.41-0.42), cfg3 1024 x 1024 0.80 ms (1.06-1.14).
__device__ unsigned g_dma_zero[256];               // a row of zeros: the source of a B row whose shifted step falls outside its sequence
typedef __attribute__((address_space(3))) void* dma_ldsp;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int DM_BM = 256, DM_SK = 16, DM_NTH = 512, DM_NST = 4, DM_PPW = 4;
constexpr int dm_nst(int BN) { return BN == 256 ? DM_NST : 3; }       // 128-column tiles: three 24 KiB stages, TWO workgroups per CU (114 registers)
constexpr size_t dm_lds_bytes(int BN) { return (size_t)dm_nst(BN) * (DM_SK * 256 + DM_SK * BN) * 4; }

// BN = 256: waves 2 x 4, each 128 x 64.  BN = 128 (dW_hh of the BiLSTM-128 layers: N = H = 128): waves 4 x 2, each 64 x 64; a stage row of B is 512 bytes,
// copied by the low half of a wave (same row assignment, same piece count: the vmcnt arithmetic does not change).
template <int BN>
__global__ __launch_bounds__(DM_NTH) void gemm_bf16x3_tn_dma(GemmP p0, GemmP p1, int np) {
    extern __shared__ __attribute__((aligned(1024))) unsigned dsm[];
    constexpr int NST = dm_nst(BN), SK = DM_SK, PPW = DM_PPW;
    constexpr int WN = BN / 64, WM = 8 / WN, MI = DM_BM / WM / 32;       // 4, 2, 4  |  2, 4, 2
    constexpr int ROWA = 256, ROWB = BN, STW = SK * ROWA + SK * ROWB;
    const int x8 = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int prob = np == 2 ? (slot & 1) : 0;
    const GemmP& p = prob ? p1 : p0;
    if (p.only_if && *p.only_if == 0) return;
    const int gx = p.N / BN, gy = p.M / DM_BM;
    int t;
    {
        const int s2 = np == 2 ? (slot >> 1) : slot;
        const int ntiles = gx * gy * p.splits;
        const int q = ntiles / 8, r = ntiles % 8;
        const int lo = x8 < r ? x8 * (q + 1) : r * (q + 1) + (x8 - r) * q;
        const int hi = lo + (x8 < r ? q + 1 : q);
        t = lo + s2;
        if (t >= hi) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / WN, wn = w % WN, half = lane >> 5, l31 = lane & 31;
    const int grp = w >> 2;                        // waves 0-3 / 4-7: one wave of each group per SIMD
    const int bx = t % gx, by = (t / gx) % gy, bz = t / (gx * gy);
    const int m0 = by * DM_BM, n0 = bx * BN, kb = bz * p.kchunk, ke = min(p.K, kb + p.kchunk);
    const int nst = (ke - kb) / SK;                // (the launcher checked: every chunk is a multiple of 16 rows)
    const int mphys = m0 + ((p.skip_by && m0 >= p.skip_at) ? p.skip_by : 0);
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0xfffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0xfffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)g_dma_zero, 0, 1024u, 0x00020000);
    // this wave's B rows of a stage are k0 + w and k0 + w + 8: their steps inside the sequence, carried along (no division in the loop)
    int tt0 = 0, tt1 = 0;
    if (p.seqT > 0) { tt0 = (kb + w) % p.seqT; tt1 = (kb + w + 8) % p.seqT; }
    int issued = 0;                                // stages issued so far (they are issued in order)
    auto piece = [&](int q) {                      // (q is a compile-time constant at every call site)
        unsigned* base = dsm + (issued % NST) * STW;
        const int k0 = kb + issued * SK;
        const bool isA = q < 2;                    // wave w copies rows w, w + 8 of A's 16 stage rows, then of B's
        const int r = w + 8 * (q & 1);
        if (isA) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (dma_ldsp)(base + r * ROWA), 16, (unsigned)lane * 16u, ((unsigned)(k0 + r) * (unsigned)p.lda + (unsigned)mphys) * 4u, 0, 0);
        } else {
            const int kr = k0 + r;
            const int tt = ((q & 1) ? tt1 : tt0) + p.shiftB;
            const bool ok = p.seqT <= 0 || (tt >= 0 && tt < p.seqT);
            unsigned* dst = base + SK * ROWA + r * ROWB;
            if (BN == 256 || lane < 32) {          // (a 512-byte row: the low half of the wave)
                if (ok) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (dma_ldsp)dst, 16, (unsigned)lane * 16u, ((unsigned)(kr + p.shiftB) * (unsigned)p.ldb + (unsigned)n0) * 4u, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (dma_ldsp)dst, 16, (unsigned)lane * 16u, 0u, 0, 0);
            }
        }
    };
    auto stage_issued = [&]() {
        if (p.seqT > 0) {
            tt0 += SK; tt1 += SK;
            while (tt0 >= p.seqT) tt0 -= p.seqT;
            while (tt1 >= p.seqT) tt1 -= p.seqT;
        }
        ++issued;
    };
    auto issue = [&]() {
#pragma unroll
        for (int q = 0; q < PPW; ++q) piece(q);
        stage_issued();
    };
    // BN = 256: a stage's four pieces are issued BETWEEN the MFMAs of the wave's MFMA slot (an LDS-DMA piece costs ~100 cycles of issue: in the read slot that
    // is on the slot's critical path, among the MFMAs it mostly is not; prototype: -4 ... -8 %).  BN = 128 (three stages) keeps them at the head of the read
    // slot: issued a slot later, the next stage would have to be waited for with vmcnt(0).
    constexpr bool AMONG = BN == 256;
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) if (st < nst) issue();
    if (NST - 1 <= nst) wait_vm<(NST - 2) * PPW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                  // stage 0 is complete
    if (grp == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one barrier behind group 0
    for (int it = 0; it < nst; ++it) {
        const bool more = it + NST - 1 < nst;
        if (!AMONG && more) issue();               // stage it + NST - 1, into the buffer of stage it - 1 (both groups finished reading it before the barrier in front of this slot)
        const unsigned* sa = dsm + (it % NST) * STW;
        const unsigned* sb = sa + SK * ROWA;
        bf16x8 ah[MI], al[MI], bh[2], bl[2];
        {
            // Fragment words by hand-placed ds_read2st64_b32: each instruction fetches the two words of ONE operand register pair (rows e, e + 2 of A's hi / lo
            // plane; rows e, e + 1 of B), so the results land where the MFMA / the split read them.  Left to the compiler the loads pair words of DIFFERENT
            // fragments (columns m, m + 32) and ~100 v_mov per stage reassemble the operands: the read slot was VALU-bound and longer than the MFMA slot
            // (tools/micro/gemm_tn_dma.hip: 1130 -> 845 ticks per stage; profiles/r06_s23_*).
            typedef unsigned long long u64t;
            typedef u64t u64x2 __attribute__((ext_vector_type(2)));
            constexpr int UB = ROWB / 64;              // B's row stride in the instruction's 64-dword offset units (4 | 2); A's is 4
            u64t ra[MI][2][2], rbx[2][4];
            const unsigned abase = (unsigned)(unsigned long long)(dma_ldsp)(sa + (half * 8) * ROWA + wm * (DM_BM / WM) + l31);
            const unsigned bbase = (unsigned)(unsigned long long)(dma_ldsp)(sb + (half * 8) * ROWB + wn * 64 + l31);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const unsigned ad = abase + i * 128;
                asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:8" : "=v"(ra[i][0][0]) : "v"(ad));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:16 offset1:24" : "=v"(ra[i][0][1]) : "v"(ad));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:4 offset1:12" : "=v"(ra[i][1][0]) : "v"(ad));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:20 offset1:28" : "=v"(ra[i][1][1]) : "v"(ad));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned ad = bbase + j * 128;
                asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(rbx[j][0]) : "v"(ad), "n"(0 * UB), "n"(1 * UB));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(rbx[j][1]) : "v"(ad), "n"(2 * UB), "n"(3 * UB));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(rbx[j][2]) : "v"(ad), "n"(4 * UB), "n"(5 * UB));
                asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(rbx[j][3]) : "v"(ad), "n"(6 * UB), "n"(7 * UB));
            }
            // (the registers are operands of the wait: nothing that uses them can be scheduled in front of it)
            if constexpr (MI == 4)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(ra[0][0][0]), "+v"(ra[0][0][1]), "+v"(ra[0][1][0]), "+v"(ra[0][1][1]), "+v"(ra[1][0][0]), "+v"(ra[1][0][1]), "+v"(ra[1][1][0]), "+v"(ra[1][1][1]),
                               "+v"(ra[2][0][0]), "+v"(ra[2][0][1]), "+v"(ra[2][1][0]), "+v"(ra[2][1][1]), "+v"(ra[3][0][0]), "+v"(ra[3][0][1]), "+v"(ra[3][1][0]), "+v"(ra[3][1][1]),
                               "+v"(rbx[0][0]), "+v"(rbx[0][1]), "+v"(rbx[0][2]), "+v"(rbx[0][3]), "+v"(rbx[1][0]), "+v"(rbx[1][1]), "+v"(rbx[1][2]), "+v"(rbx[1][3]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(ra[0][0][0]), "+v"(ra[0][0][1]), "+v"(ra[0][1][0]), "+v"(ra[0][1][1]), "+v"(ra[1][0][0]), "+v"(ra[1][0][1]), "+v"(ra[1][1][0]), "+v"(ra[1][1][1]),
                               "+v"(rbx[0][0]), "+v"(rbx[0][1]), "+v"(rbx[0][2]), "+v"(rbx[0][3]), "+v"(rbx[1][0]), "+v"(rbx[1][1]), "+v"(rbx[1][2]), "+v"(rbx[1][3]));
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const u64x2 h = {ra[i][0][0], ra[i][0][1]}, l = {ra[i][1][0], ra[i][1][1]};
                ah[i] = __builtin_bit_cast(bf16x8, h); al[i] = __builtin_bit_cast(bf16x8, l);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u64x2 a0 = {rbx[j][0], rbx[j][1]}, a1 = {rbx[j][2], rbx[j][3]};
                bf16x4 h0, l0, h1, l1;
                split4(__builtin_bit_cast(f32x4, a0), h0, l0); split4(__builtin_bit_cast(f32x4, a1), h1, l1);
                bh[j] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                bl[j] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
        // every wave's pieces of stage it + 1 must have landed before the barrier in front of group 0's next read slot: group 1 is in its read slot then,
        // group 0 in its MFMA slot
        // (AMONG: this wave issues stage it + 3 only in its MFMA slot below -- one stage fewer is in flight at group 1's wait)
        if (grp == 1) { if (AMONG ? (it + NST - 2 < nst) : more) wait_vm<(NST - 2) * PPW - (AMONG ? PPW : 0)>(); else wait_vm<0>(); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
        // (b_hi a_lo), (b_lo a_hi), (b_hi a_hi) per accumulator, in that order -- the kernel above's; consecutive instructions go to different accumulators
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 1 ? bl[j] : bh[j], term == 0 ? al[i] : ah[i], acc[i][j], 0, 0, 0);
                if constexpr (AMONG) {             // pieces 0..3 behind the 4th, 8th, 12th and 16th MFMA
                    if (term < 2 && (i & 1) && more) piece(term * 2 + (i >> 1));
                }
            }
        if (AMONG && more) stage_issued();
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) { if (more) wait_vm<(NST - 2) * PPW>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    float* outp = p.part + (size_t)bz * p.M * p.N;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (DM_BM / WM) + i * 32 + l31;
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

// Round 6: the NT projections  C[m][n] = sum_k X[m][k] W[n][k] + bias[n]  (the layers' input projections) fed by LDS-DMA.
//   X : fp32 rows (K contiguous); a wave owns 32 rows x ALL 256 columns of the tile, so every element of X is split into (hi, lo) by exactly one wave
//       (the split of the staging pass above, same bits).  A stage row is 64 bytes: a DMA piece is 16 rows x 64 B, and the 16-byte chunk a lane
//       fetches is XOR-swizzled through the SOURCE address (chunk ^ (row >> 2) & 3) -- the LDS image is lane-linear -- so that the b128 fragment reads of
//       32 consecutive rows are conflict-free.
//   W : pre-split once per call into a STAGE IMAGE (pack_w_stage_image): per 256-column tile and 16-k stage one contiguous 16 KiB block that IS the LDS
//       image, [plane hi / lo][k octet][256 n] x 8 bf16: its DMA pieces are contiguous 1 KiB reads, its fragment reads conflict-free b128, nothing to convert.
// 128 x 256 tiles, four waves, three 24 KiB stages, counted vmcnt, one barrier per stage; TWO such workgroups per CU (2 x 76 KiB LDS) desynchronise by
// themselves: one's write-out runs beside the other's MFMAs.  Persistent over an XCD-contiguous tile list ((tile, stage) is one iteration space: the next
// tile's first stages are in flight during this tile's last; the write-out's stores drain behind the next tile's MFMAs -- the counted vmcnt allows for them).
// Per element: k ascending, (w_hi x_lo), (w_lo x_hi), (w_hi x_hi) per 16 k, bias added to the finished sum -- the kernel above's sequence.
// The MFMA takes X as its first operand (a lane then owns one COLUMN and 16 rows: a store instruction writes 2 rows x 128 contiguous bytes -- full lines;
// with W first a lane's 16-byte stores touch 32 rows x 32 bytes per instruction and the write-out, which bounds the K = 256 projections, is a third slower).
// Measured (tools/micro/gemm_nt_dma.hip, profiles/r06_s17_*): cfg2 0.24 ms (kernel above 0.29-0.33), cfg3 layer 0 0.85 (1.08-1.16), layer 1 0.31 (0.39).
constexpr int NTD_TH = 256, NTD_NST = 3, NTD_STW = 2048 + 4096, NTD_PPW = 6, NTD_SK = 16, NTD_BM = 128, NTD_BN = 256, NTD_MAXN = 1024;
constexpr size_t NTD_LDS_BYTES = (size_t)NTD_NST * NTD_STW * 4 + NTD_MAXN * 4;

struct NtdP {
    const float* X; int ldx;
    const unsigned* Wimg;         // [N / 256][K / 16][4096 words]
    const float* bias;
    float* C; int ldc;
    int M, N, K, gx;
    const unsigned* only_if;
};

// APK (the NN form, dX = dG W): the row operand is the sweeps' PK gate-gradient image -- physical rows 2j / 2j+1 hold the (hi, lo) bf16 of logical rows 2j, 2j+1,
// word by word -- so a lane reads eight hi words and eight lo words of its row PAIR and keeps its own row's halves (v_perm): nothing to convert.  Same stage
// geometry (a physical row is 64 bytes per stage either way), same pieces, same swizzle; the weight's stage image is packed from its [k][n] layout.
template <bool APK>
__global__ __launch_bounds__(NTD_TH, 2) void gemm_bf16x3_nt_dma(NtdP p) {
    if (p.only_if && *p.only_if == 0) return;
    extern __shared__ __attribute__((aligned(1024))) unsigned dsm[];
    constexpr int NST = NTD_NST, STW = NTD_STW, PPW = NTD_PPW, SK = NTD_SK;
    float* bias_l = reinterpret_cast<float*>(dsm + NST * STW);
    const int x8 = blockIdx.x & 7, slot = blockIdx.x >> 3, SL = gridDim.x >> 3;
    int lo, hi;
    {
        const int ntiles = p.gx * (p.M / NTD_BM);
        const int q = ntiles / 8, r = ntiles % 8;
        lo = x8 < r ? x8 * (q + 1) : r * (q + 1) + (x8 - r) * q;
        hi = lo + (x8 < r ? q + 1 : q);
    }
    if (lo + slot >= hi) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int nst = p.K / SK;
    for (int i = tid; i < p.N; i += NTD_TH) bias_l[i] = p.bias ? p.bias[i] : 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, 0xfffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wimg, 0, 0xfffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, 0xfffffff0u, 0x00020000);
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
    unsigned ixoff = (unsigned)(ti / p.gx) * (unsigned)NTD_BM * (unsigned)p.ldx * 4u, iwoff = (unsigned)(ti % p.gx) * (unsigned)nst * 16384u;
    auto piece = [&](int q) {                      // (q is a compile-time constant at every call site: two pieces of X, four of the weight image)
        unsigned* base = dsm + (icount % NST) * STW;
        if (q < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (dma_ldsp)(base + (32 * w + 16 * q) * 16), 16, xrel[q], ixoff + (unsigned)si * 64u, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (dma_ldsp)(base + 2048 + (w * 4 + q - 2) * 256), 16, (unsigned)lane * 16u, iwoff + (unsigned)si * 16384u + (unsigned)(w * 4 + q - 2) * 1024u, 0, 0);
    };
    auto stage_issued = [&]() {
        ++icount;
        if (++si == nst) {
            si = 0; ti += SL; iv = ti < hi;
            if (iv) { ixoff = (unsigned)(ti / p.gx) * (unsigned)NTD_BM * (unsigned)p.ldx * 4u; iwoff = (unsigned)(ti % p.gx) * (unsigned)nst * 16384u; }
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int q = 0; q < PPW; ++q) piece(q);
        stage_issued();
    };
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) if (iv) issue();
    const int xrow = APK ? 32 * w + (l31 & ~1) : 32 * w + l31;       // (APK: the hi row of the lane's row pair; its lo row is the next one)
    const int xs = (xrow >> 2) & 3;                                    // (rows 2j and 2j+1 share the swizzle: (row >> 2) & 3)
    const int xp0 = xrow * 16 + (((2 * half) ^ xs) << 2), xp1 = xrow * 16 + (((2 * half + 1) ^ xs) << 2);      // word offsets of the lane's two chunks inside a stage
    int tc = lo + slot, sc = 0, ccount = 0, since = 100;
    while (tc < hi) {
        // this wave's pieces of the stage about to be read have landed: one younger stage of its own may still be in flight -- and, in the first wait behind a
        // write-out, that tile's 128 stores (issued after the pieces waited for); vmcnt counts at most 63: 6 + 57 lets most of them stay in flight
        ++since;
        if (icount <= ccount + 1) wait_vm<0>();
        else if (since == 1 || since == 2) wait_vm<PPW + 57>();      // (both stages waited for here were issued in front of the write-out)
        else wait_vm<PPW>();
        __builtin_amdgcn_s_barrier();              // everybody's pieces landed; everybody finished reading the stage before
        const bool more = iv;                      // the next stage's six pieces go out BETWEEN this stage's MFMAs (into the buffer of the stage before): an LDS-DMA
                                                   // piece costs ~100 cycles of issue, which the matrix pipe hides there and nothing hides in front of the fragment reads
        const unsigned* sx = dsm + (ccount % NST) * STW;
        const unsigned* sw = sx + 2048;
        bf16x8 ah, al, bh[8], bl[8];
        if constexpr (APK) {
            // words k0..k3 / k4..k7 of the pair's hi row and of its lo row; this lane's row is the low (even row) or high (odd row) half of every word
            const u32x4 hA = *reinterpret_cast<const u32x4*>(sx + xp0), hB = *reinterpret_cast<const u32x4*>(sx + xp1);
            const u32x4 lA = *reinterpret_cast<const u32x4*>(sx + xp0 + 16), lB = *reinterpret_cast<const u32x4*>(sx + xp1 + 16);
            const unsigned sel = (l31 & 1) ? 0x07060302u : 0x05040100u;
            const u32x4 hv = {__builtin_amdgcn_perm(hA.y, hA.x, sel), __builtin_amdgcn_perm(hA.w, hA.z, sel), __builtin_amdgcn_perm(hB.y, hB.x, sel), __builtin_amdgcn_perm(hB.w, hB.z, sel)};
            const u32x4 lv = {__builtin_amdgcn_perm(lA.y, lA.x, sel), __builtin_amdgcn_perm(lA.w, lA.z, sel), __builtin_amdgcn_perm(lB.y, lB.x, sel), __builtin_amdgcn_perm(lB.w, lB.z, sel)};
            ah = __builtin_bit_cast(bf16x8, hv); al = __builtin_bit_cast(bf16x8, lv);
        } else {
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
            for (int j = 0; j < 8; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 0 ? al : ah, term == 1 ? bl[j] : bh[j], acc[j], 0, 0, 0);
                if ((j & 3) == 3 && more) piece(term * 2 + (j >> 2));
            }
        if (more) stage_issued();
        if (sc == nst - 1) {                       // the tile is complete: write it out, start the next from zero
            // lane owns column n = j*32 + l31; register 4 g + e holds row 8 g + 4 half + e
            const int n0 = (tc % p.gx) * NTD_BN, mb = (tc / p.gx) * NTD_BM + 32 * w;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = n0 + j * 32 + l31;
                const float bv = bias_l[n];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + 8 * (r >> 2) + 4 * half + (r & 3);
                    // non-temporal: the projection is read once, by the next kernel, long after it has left this L2 (A/B: -6 ... -8 % on the K = 256 projections)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[j][r] + bv), rc, ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 4u, 0, 2 /* nt */);
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

// the same image from a weight stored [k][n] (K x N fp32, row stride ldw): the NN form's B operand
__global__ void pack_w_stage_image_kn(const float* W, int ldw, int N, int K, unsigned* img) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // one (k octet, n), n fastest: coalesced reads
    const int oct = K / 8;
    if (i >= (long)N * oct) return;
    const int o = (int)(i / N), n = (int)(i % N);
    const float* src = W + (size_t)(o * 8) * ldw + n;
    const f32x4 x0 = {src[0], src[ldw], src[2 * (size_t)ldw], src[3 * (size_t)ldw]};
    const f32x4 x1 = {src[4 * (size_t)ldw], src[5 * (size_t)ldw], src[6 * (size_t)ldw], src[7 * (size_t)ldw]};
    bf16x4 h0, l0, h1, l1;
    split4(x0, h0, l0); split4(x1, h1, l1);
    const int ntile = n >> 8, nn = n & 255, stage = o >> 1, half = o & 1;
    unsigned* blk = img + ((size_t)ntile * (K / 16) + stage) * 4096;
    const u32x2v a = __builtin_bit_cast(u32x2v, h0), b = __builtin_bit_cast(u32x2v, h1), c = __builtin_bit_cast(u32x2v, l0), d = __builtin_bit_cast(u32x2v, l1);
    u32x4 hv = {a[0], a[1], b[0], b[1]}, lv = {c[0], c[1], d[0], d[1]};
    *reinterpret_cast<u32x4*>(blk + ((0 * 2 + half) * 256 + nn) * 4) = hv;
    *reinterpret_cast<u32x4*>(blk + ((1 * 2 + half) * 256 + nn) * 4) = lv;
}

// W (N x K fp32, row stride ldw) -> the stage image: block (n tile, stage) = [plane][k octet][256 n] x 8 bf16
__global__ void pack_w_stage_image(const float* W, int ldw, int N, int K, unsigned* img) {
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
// can the paired launch run in this thread's / process's configuration?  (An XCD restriction, an ablation setting or DEP_GEMM_BM=128 apply to the
// single launches only: dep_gemm_tn_pair then reports "not covered" and the caller issues the two contractions itself -- ADVICE r5.)
bool dep_gemm_bf16x3_pair_ok() {
    static int plain = -1;
    if (plain < 0) { const char* a = getenv("DEP_GEMM_ABLATE"); const char* b = getenv("DEP_GEMM_BM"); plain = ((a && atoi(a) != 0) || (b && atoi(b) == 128)) ? 0 : 1; }
    return plain && g_xcd_lo == 0 && g_xcd_n == 8;
}
// may this TN contraction (A = PK image, B = fp32 rows, split-K) take the LDS-DMA kernel?  DEP_GEMM_TN_DMA=0: the register-staged kernel everywhere.
static bool tn_dma_ok(int M, int N, int K, int lda, int ldb, int splits, int kchunk, const float* part, int skip_at, int skip_by) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("DEP_GEMM_TN_DMA"); off = (e && e[0] == '0') ? 1 : 0; }
    return !off && dep_gemm_bf16x3_pair_ok() && M % DM_BM == 0 && N % 128 == 0 && splits > 1 && part && kchunk % DM_SK == 0 && K % DM_SK == 0 &&
           lda % 4 == 0 && ldb % 4 == 0 && (skip_by == 0 || skip_at % DM_BM == 0);
}
static int tn_dma_launch(const GemmP& p0, const GemmP& p1, int np, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16x3_tn_dma<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dm_lds_bytes(256));
        (void)hipFuncSetAttribute((const void*)gemm_bf16x3_tn_dma<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dm_lds_bytes(128));
        attr = true;
    }
    const int BN = p0.N % 256 == 0 ? 256 : 128;
    const int ntiles = (p0.M / DM_BM) * (p0.N / BN) * p0.splits;
    const dim3 g((unsigned)((ntiles + 7) / 8 * 8 * np));
    if (BN == 256) DEP_LAUNCH(gemm_bf16x3_tn_dma<256>, g, dim3(DM_NTH), dm_lds_bytes(256), s, p0, p1, np);
    else DEP_LAUNCH(gemm_bf16x3_tn_dma<128>, g, dim3(DM_NTH), dm_lds_bytes(128), s, p0, p1, np);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
// may this NT projection take the LDS-DMA kernel?  `ws` holds the weight's stage image.  DEP_GEMM_NT_DMA=0: the register-staged kernel everywhere.
static bool nt_dma_ok(int M, int N, int K, int lda, int ldb, int ldc, const float* A, const float* B, const float* C, const float* bias, float beta, int splits,
                      void* ws, size_t ws_bytes) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("DEP_GEMM_NT_DMA"); off = (e && e[0] == '0') ? 1 : 0; }
    auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    return !off && dep_gemm_bf16x3_pair_ok() && M % NTD_BM == 0 && N % NTD_BN == 0 && N <= NTD_MAXN && K % NTD_SK == 0 && K >= 3 * NTD_SK && splits == 1 && beta == 0.f &&
           lda % 4 == 0 && ldb % 4 == 0 && a16(A) && a16(B) && C && ws && a16(ws) && ws_bytes >= (size_t)N * K * 4 && dep_gemm_predicate() == nullptr &&
           (size_t)M * lda * 4 < 0xfffffff0ull && (size_t)M * ldc * 4 < 0xfffffff0ull;
}
static int nt_dma_launch(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias, void* ws, hipStream_t s, bool nn_pk = false) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16x3_nt_dma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTD_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_bf16x3_nt_dma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTD_LDS_BYTES);
        attr = true;
    }
    if (nn_pk) DEP_LAUNCH(pack_w_stage_image_kn, dim3((unsigned)dep_cdiv((long)N * (K / 8), 256)), dim3(256), 0, s, B, ldb, N, K, (unsigned*)ws);
    else DEP_LAUNCH(pack_w_stage_image, dim3((unsigned)dep_cdiv((long)N * (K / 8), 256)), dim3(256), 0, s, B, ldb, N, K, (unsigned*)ws);
    DEP_CHECK_LAUNCH();
    NtdP p{A, lda, (const unsigned*)ws, bias, C, ldc, M, N, K, N / NTD_BN, dep_gemm_predicate()};
    const int per_xcd = dep_cdiv((M / NTD_BM) * p.gx, 8);
    const dim3 g((unsigned)((per_xcd < 64 ? per_xcd : 64) * 8));      // two resident workgroups per CU, 32 CUs per XCD
    if (nn_pk) DEP_LAUNCH(gemm_bf16x3_nt_dma<true>, g, dim3(NTD_TH), NTD_LDS_BYTES, s, p);
    else DEP_LAUNCH(gemm_bf16x3_nt_dma<false>, g, dim3(NTD_TH), NTD_LDS_BYTES, s, p);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_gemm_set_xcds(int lo, int n) { if (lo < 0 || n < 1 || lo + n > 8) return DEP_ERR_ARG; g_xcd_lo = lo; g_xcd_n = n; return DEP_OK; }

// Same contract as dep_gemm_internal (gemm.hip); `splits` is decided by the caller's shared heuristic.
int dep_gemm_bf16x3_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                           int ldb, float* C, int ldc, const float* bias, float beta, int seq_T, int shiftB,
                           int splits, int kchunk, float* part, bool vec, hipStream_t s, int terms, void* ws, size_t ws_bytes) {
    static int abl = -1, persist = -1, bm256 = -1;
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
    if (abl < 0) {      // measurement hook (profiles/r06_gemm_clock_ablation.txt): the results are garbage, and the process says so once
        const char* e = getenv("DEP_GEMM_ABLATE"); abl = e ? atoi(e) : 0;
        if (abl) fprintf(stderr, "libdep_rnn: DEP_GEMM_ABLATE=%d -- parts of the split-precision GEMM are compiled out of the loop, RESULTS ARE GARBAGE (timing only)\n", abl);
    }
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
    if (!transA && transB && fa == FMT_F32 && fb == FMT_F32 && terms == 3 && vec && !abl && seq_T <= 0 &&
        nt_dma_ok(M, N, K, lda, ldb, ldc, A, B, C, bias, beta, splits, ws, ws_bytes))
        return nt_dma_launch(M, N, K, A, lda, B, ldb, C, ldc, bias, ws, s);      // Round 6: the input projections whose shape fits take the LDS-DMA kernel
    // ... and so does dX = dG W (NN) over the PK gate-gradient image: the same kernel with the row operand's fragments read, not converted
    if (!transA && !transB && fa == FMT_PK && fb == FMT_F32 && terms == 3 && vec && !abl && seq_T <= 0 && ldb % 4 == 0 &&
        nt_dma_ok(M, N, K, lda, ldb, ldc, A, B, C, bias, beta, splits, ws, ws_bytes))
        return nt_dma_launch(M, N, K, A, lda, B, ldb, C, ldc, bias, ws, s, true);
    if (transA && !transB && fa == FMT_PK && fb == FMT_F32 && terms == 3 && vec && !abl && tn_dma_ok(M, N, K, lda, ldb, splits, kchunk, part, p.skip_at, p.skip_by)) {
        // Round 6: the weight-gradient contractions whose shape fits take the LDS-DMA kernel (bit-identical partial sums); the reduce below is shared
        const int rc = tn_dma_launch(p, p, 1, s);
        if (rc) return rc;
        const long n = (long)M * N;
        DEP_LAUNCH(splitk_reduce2, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, dep_gemm_predicate(), part, splits, M, N, C, ldc, bias, beta);
        DEP_CHECK_LAUNCH();
        return DEP_OK;
    }
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
    if (tn_dma_ok(M, N, K, lda, ldb0, splits, kchunk, part0, skip_at1, skip_by1) && ldb1 % 4 == 0) {
        const int rc = tn_dma_launch(p0, p1, 2, s);
        if (rc) return rc;
        const long n = (long)M * N;
        DEP_LAUNCH(splitk_reduce2_pair, dim3(dep_cdiv(n, 256), 2), dim3(256), 0, s, part0, part1, splits, M, N, C0, ldc0, C1, ldc1);
        DEP_CHECK_LAUNCH();
        return DEP_OK;
    }
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
