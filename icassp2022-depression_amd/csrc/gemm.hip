// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products, f32
// accumulate == an fmaf chain), LDS-tiled 128x128x32, 4 waves of 64x64, register-staged prefetch.
//
//   C[M,N] = opA(A)[M,K] * opB(B)[K,N] + bias[N] + beta*C
//
// Used for every time-parallel contraction of the hot path: the input projections x_t W_ih^T of
// nn.GRU / nn.LSTM (Classification/audio_gru_whole.py:105, text_bilstm_whole.py:105), the
// nn.Linear layers of the heads (audio_gru_whole.py:67,70) and, in backward, dX = dG W and the
// weight gradients dW = dG^T X (split-K over the B*T rows, deterministic two-pass reduction).
#include <atomic>
#include <string.h>
#include "dep_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int LD_K = 129;   // LDS leading dim for operands loaded K-contiguous (scalar transposing stores)
constexpr int LD_M = 132;   // LDS leading dim for operands loaded MN-contiguous (16-B vector stores)

struct GemmP {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias; float beta;
    int seqT, shiftB;
    int kchunk, splits;
    float* part;
    int gx, gy;             // tile grid (x: N tiles, y: M tiles); the launch is 1-D, see dep_xcd_tile            // split-K partials [splits][M][N] or nullptr
    const unsigned* only_if;    // run only if this device word is non-zero (dep_gemm_set_predicate), or nullptr
};

// Load one (128 x 32) operand tile into registers.  TR = operand stored MN-contiguous.
template <bool TR, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int mn0, int MN, int k0,
                                          int Kend, int tid, float (&r)[4][4], int seqT, int shift) {
    if (!TR) {   // element (mn,k) at P[mn*ld + k]
        const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mn = mn0 + rr + 32 * i, k = k0 + kq * 4;
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
    } else {     // element (mn,k) at P[(k+shift)*ld + mn]
        const int mq = tid & 31, kr = tid >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + kr + 8 * i, mn = mn0 + mq * 4;
            bool ok = k < Kend;
            if (seqT > 0) { const int tt = k % seqT + shift; ok = ok && tt >= 0 && tt < seqT; }
            const float* src = P + ((long)k + shift) * ld + mn;
            if (VEC) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok && mn < MN) v = *reinterpret_cast<const f32x4*>(src);
                r[i][0] = v[0]; r[i][1] = v[1]; r[i][2] = v[2]; r[i][3] = v[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[i][e] = (ok && mn + e < MN) ? src[e] : 0.f;
            }
        }
    }
}

template <bool TR>
__device__ __forceinline__ void store_tile(float* S, int tid, const float (&r)[4][4]) {
    if (!TR) {
        const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) S[(kq * 4 + e) * LD_K + rr + 32 * i] = r[i][e];
    } else {
        const int mq = tid & 31, kr = tid >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = {r[i][0], r[i][1], r[i][2], r[i][3]};
            *reinterpret_cast<f32x4*>(&S[(kr + 8 * i) * LD_M + mq * 4]) = v;
        }
    }
}

// TA: A stored (K,M) ; TB: B stored (N,K)  [note the asymmetry, matches the public API]
template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(NT) void gemm_mfma(GemmP p) {
    if (p.only_if && *p.only_if == 0) return;
    constexpr bool A_TR = TA;        // A MN-contiguous when stored (K,M)
    constexpr bool B_TR = !TB;       // B MN-contiguous when stored (K,N)
    constexpr int LDA_S = A_TR ? LD_M : LD_K;
    constexpr int LDB_S = B_TR ? LD_M : LD_K;
    __shared__ __attribute__((aligned(16))) float smem[BK * LDA_S + BK * LDB_S];
    float* As = smem;
    float* Bs = smem + BK * LDA_S;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    int bx, by, bz;
    dep_xcd_tile(p.gx, p.gy, p.splits, bx, by, bz);
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float ra[4][4], rb[4][4];
    load_tile<A_TR, VEC>(p.A, p.lda, m0, p.M, kbeg, kend, tid, ra, 0, 0);
    load_tile<B_TR, VEC>(p.B, p.ldb, n0, p.N, kbeg, kend, tid, rb, p.seqT, p.shiftB);

    const int half = lane >> 5, l31 = lane & 31;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        store_tile<A_TR>(As, tid, ra);
        store_tile<B_TR>(Bs, tid, rb);
        __syncthreads();
        if (k0 + BK < kend) {
            load_tile<A_TR, VEC>(p.A, p.lda, m0, p.M, k0 + BK, kend, tid, ra, 0, 0);
            load_tile<B_TR, VEC>(p.B, p.ldb, n0, p.N, k0 + BK, kend, tid, rb, p.seqT, p.shiftB);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int kr = kk * 2 + half;
            const float a0 = As[kr * LDA_S + wm * 64 + l31];
            const float a1 = As[kr * LDA_S + wm * 64 + 32 + l31];
            const float b0 = Bs[kr * LDB_S + wn * 64 + l31];
            const float b1 = Bs[kr * LDB_S + wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: D[i][j]: j = lane&31 (n), i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (m)
    const bool split = p.part != nullptr;
    float* outp = split ? p.part + (size_t)bz * p.M * p.N : p.C;
    const int ldo = split ? p.N : p.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= p.N) continue;
            const float bv = (!split && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (m < p.M) {
                    float v = acc[i][j][e] + bv;
                    float* dst = outp + (size_t)m * ldo + n;
                    if (!split && p.beta != 0.f) v += p.beta * *dst;
                    *dst = v;
                }
            }
        }
}

__global__ void splitk_reduce(const unsigned* only_if, const float* __restrict__ part, int splits, int M, int N, float* C, int ldc,
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

// Reference-quality fallback (any shape/alignment); selected with DEP_GEMM_NAIVE=1 for A/B checks.
__global__ void gemm_naive(GemmP p, int TA, int TB) {
    if (p.only_if && *p.only_if == 0) return;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= p.N || m >= p.M) return;
    float s = 0.f;
    for (int k = 0; k < p.K; ++k) {
        const float a = TA ? p.A[(size_t)k * p.lda + m] : p.A[(size_t)m * p.lda + k];
        float b;
        if (TB) {
            b = p.B[(size_t)n * p.ldb + k];
        } else {
            bool ok = true;
            if (p.seqT > 0) { const int tt = k % p.seqT + p.shiftB; ok = tt >= 0 && tt < p.seqT; }
            b = ok ? p.B[((long)k + p.shiftB) * p.ldb + n] : 0.f;
        }
        s = fmaf(a, b, s);
    }
    if (p.bias) s += p.bias[n];
    float* dst = p.C + (size_t)m * p.ldc + n;
    if (p.beta != 0.f) s += p.beta * *dst;
    *dst = s;
}

// Small problems (the heads' nn.Linear layers, the attention projection: a few hundred rows/columns): a 128x128 tile
// grid would occupy a handful of CUs and walk K serially.  Here a workgroup owns one 32x32 output tile, its four waves
// take a quarter of K each and feed v_mfma_f32_32x32x2_f32 straight from global memory (operand (row, k) -> lane
// (row & 31, k & 1): no LDS staging, everything is L2-resident at these sizes), and the four partial tiles are summed
// through LDS.  Strided element access covers all operand forms.
struct SmallP {
    int M, N, K;
    const float* A; long sam, sak;
    const float* B; long sbk, sbn;
    float* C; int ldc;
    const float* bias; float beta;
    const unsigned* only_if;
};

__global__ __launch_bounds__(256) void gemm_small(SmallP p) {
    if (p.only_if && *p.only_if == 0) return;
    __shared__ float red[4][16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int kq = ((p.K + 3) / 4 + 1) & ~1;
    const int kb = w * kq, ke = min(p.K, kb + kq);
    const bool mok = m0 + i < p.M, nok = n0 + i < p.N;
    const float* ap = p.A + (long)(m0 + i) * p.sam;
    const float* bp = p.B + (long)(n0 + i) * p.sbn;
    auto load8 = [&](int k, float (&a)[8], float (&b)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = k + 2 * e + h;
            a[e] = (mok && kk < ke) ? ap[(long)kk * p.sak] : 0.f;
            b[e] = (nok && kk < ke) ? bp[(long)kk * p.sbk] : 0.f;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float a0[8], b0[8], a1[8], b1[8];
    load8(kb, a0, b0);
    for (int k = kb; k < ke; k += 32) {
        load8(k + 16, a1, b1);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc, 0, 0, 0);
        load8(k + 32, a0, b0);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[w][e][lane] = acc[e];
    __syncthreads();
    const int l = threadIdx.x & 63, rq = threadIdx.x >> 6;
    const int n = n0 + (l & 31);
    if (n >= p.N) return;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = rq * 4 + rr;
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (m >= p.M) continue;
        float v = (red[0][r][l] + red[1][r][l]) + (red[2][r][l] + red[3][r][l]) + bv;
        float* dst = p.C + (size_t)m * p.ldc + n;
        if (p.beta != 0.f) v += p.beta * *dst;
        *dst = v;
    }
}

thread_local long g_split_target_override = 0;      // dep_gemm_set_split_target: per calling thread, 0 = the default / DEP_GEMM_SPLIT_TARGET
int choose_splits(int M, int N, int K) {
    const long tiles = (long)dep_cdiv(M, BM) * dep_cdiv(N, BN);
    if (tiles >= 256 || K < 1024) return 1;
    // fill the persistent bf16x3 grid without spilling into a second round: its 256x128 tiles run two per CU (512 slots),
    // i.e. 1024 tile-chunks counted in 128x128 tiles, rounded DOWN (measured on dW at cfg2: 504 workgroup-tiles 0.271 ms,
    // 384 -> 0.306 ms, 516 -> 0.366 ms)
    static long target = -1;
    if (target < 0) { const char* e = getenv("DEP_GEMM_SPLIT_TARGET"); target = e ? atol(e) : 1024; }
    long s = (g_split_target_override > 0 ? g_split_target_override : target) / tiles;
    const long maxs = K / 256;
    if (s > maxs) s = maxs;
    if (s > 128) s = 128;
    return s < 1 ? 1 : (int)s;
}

}  // namespace
// Split-K target of the calling thread's next contractions, in 128x128 tile-chunks (0 = default).  Round 5: a layer whose dW_ih and dW_hh run as
// ONE paired launch asks for 512: the pair then fills the persistent grid in one round (2 x 252 tile-jobs on 512 slots) with half the partial-sum
// traffic; the single launches want 1024.  Set for both paths of such a layer, so that DEP_DW_PAIR=0 stays bit-identical to the pair.
void dep_gemm_set_split_target(long target) { g_split_target_override = target; }
namespace {

bool naive_forced() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DEP_GEMM_NAIVE"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

}  // namespace

// ---- precision mode of the large contractions ----------------------------------------------------------
// 0 = exact fp32 MFMA everywhere ; 1 = 3-term bf16 split (gemm_bf16x3.hip) for contractions of at least
// g_split_min_macs multiply-adds (the time-parallel projections / dX / dW of the RNN stacks).  dep_gemm_f32 is
// always exact.  Default: DEP_GEMM_MODE env ("f32" -> 0), else 1.
int dep_gemm_bf16x3_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                           int ldb, float* C, int ldc, const float* bias, float beta, int seq_T, int shiftB,
                           int splits, int kchunk, float* part, bool vec, hipStream_t s, int terms, void* ws, size_t ws_bytes);
static std::atomic<int> g_split_mode{-1};            // process-global (documented in dep_rnn.h)
static std::atomic<long> g_split_min_macs{1L << 28};
static thread_local const unsigned* g_only_if = nullptr;
// Scratch of the calling thread's UNSPLIT contractions (gemm_bf16x3_nt_dma's weight image).  Not the `ws` argument: a workspace turns split-K on for
// small problems, and the RNN calls' projections must keep summing as they always did (tests/golden/device_bits.json).
static thread_local void* g_scratch = nullptr;
static thread_local size_t g_scratch_bytes = 0;
void dep_gemm_set_scratch(void* p, size_t bytes) { g_scratch = p; g_scratch_bytes = bytes; }
void dep_gemm_set_predicate(const unsigned* only_if) { g_only_if = only_if; }
const unsigned* dep_gemm_predicate() { return g_only_if; }
static thread_local int g_force_exact = 0;       // per calling thread: 1 = this call is exact fp32, 2 = this call is bf16x3 whatever the global mode
static void init_split_mode() {
    if (g_split_mode < 0) { const char* e = getenv("DEP_GEMM_MODE"); g_split_mode = (e && e[0] == 'f') ? 0 : ((e && !strcmp(e, "bf16")) ? 2 : ((e && !strcmp(e, "bf16s")) ? 3 : 1)); }   // "f32" | "bf16" | "bf16s" | default ("bf16x3")
}
extern "C" int dep_set_gemm_mode(int mode, long min_macs) {
    if (mode < 0 || mode > 3) { dep_set_error("dep_set_gemm_mode: mode must be 0 (f32), 1 (bf16x3 split), 2 (bf16 products) or 3 (bf16 products + bf16 storage)"); return DEP_ERR_ARG; }
    g_split_mode = mode;
    if (min_macs >= 0) g_split_min_macs = min_macs;
    return DEP_OK;
}
extern "C" int dep_get_gemm_mode(void) { init_split_mode(); return g_split_mode; }
bool dep_gemm_uses_bf16x3(int M, int N, int K, int seq_T) {
    init_split_mode();
    // same branch order as dep_gemm_internal
    if (naive_forced()) return false;
    if (seq_T <= 0 && (long)dep_cdiv(M, BM) * dep_cdiv(N, BN) < 32 && K <= 8192 && (long)M * N * K <= (1L << 27)) return false;    // gemm_small
    if (g_force_exact == 2) return true;
    return g_force_exact == 0 && g_split_mode >= 1 && (long)M * N * K >= g_split_min_macs;
}

extern "C" size_t dep_gemm_workspace_bytes(int transA, int transB, int M, int N, int K) {
    (void)transA; (void)transB;
    const int s = choose_splits(M, N, K);
    return s > 1 ? dep_align((size_t)s * M * N * sizeof(float)) : 0;
}

int dep_gemm_internal(int transA, int transB, int M, int N, int K, const float* A, int lda,
                      const float* B, int ldb, float* C, int ldc, const float* bias, float beta,
                      int seq_T, int shiftB, void* ws, size_t ws_bytes, hipStream_t s) {
    DEP_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C);
    DEP_CHECK_ARG(!(transA && transB));
    DEP_CHECK_ARG(!(seq_T > 0 && transB));
    GemmP p{M, N, K, A, lda, B, ldb, C, ldc, bias, beta, seq_T, shiftB, K, 1, nullptr, 1, 1, dep_gemm_predicate()};
    init_split_mode();
    const bool to_split = !naive_forced() && (g_force_exact == 2 || (g_force_exact == 0 && g_split_mode >= 1 && (long)M * N * K >= g_split_min_macs)) &&
                          !(seq_T <= 0 && (long)dep_cdiv(M, BM) * dep_cdiv(N, BN) < 32 && K <= 8192 && (long)M * N * K <= (1L << 27));
    if (dep_gemm_pk_pending() && !to_split) {      // (the launcher checks that the format matches the term count)
        dep_set_error("dep_gemm: a pre-split (PK) operand reached a contraction that does not run the three-term bf16x3 kernel");
        return DEP_ERR_ARG;
    }
    if (naive_forced()) {
        dim3 g(dep_cdiv(N, 128), M);
        DEP_LAUNCH(gemm_naive, g, dim3(128), 0, s, p, transA, transB);
        DEP_CHECK_LAUNCH();
        return DEP_OK;
    }
    if (seq_T <= 0 && (long)dep_cdiv(M, BM) * dep_cdiv(N, BN) < 32 && K <= 8192 && (long)M * N * K <= (1L << 27)) {
        SmallP q{M, N, K, A, transA ? 1L : (long)lda, transA ? (long)lda : 1L, B, transB ? 1L : (long)ldb,
                 transB ? (long)ldb : 1L, C, ldc, bias, beta, dep_gemm_predicate()};
        DepProfScope prof(transA ? DEP_PROF_GEMM_TN : (transB ? DEP_PROF_GEMM_NT : DEP_PROF_GEMM_NN), s, dep_gemm_predicate() == nullptr);
        DEP_LAUNCH(gemm_small, dim3(dep_cdiv(N, 32), dep_cdiv(M, 32)), dim3(256), 0, s, q);
        DEP_CHECK_LAUNCH();
        return DEP_OK;
    }
    int splits = choose_splits(M, N, K);
    if (splits > 1) {       // a smaller workspace than dep_gemm_workspace_bytes asked for lowers the split count, it does not drop it
        const size_t fit = ws ? ws_bytes / ((size_t)M * N * sizeof(float)) : 0;
        if (fit < (size_t)splits) splits = fit < 1 ? 1 : (int)fit;
    }
    int kchunk = K;
    if (splits > 1) {
        kchunk = dep_cdiv(dep_cdiv(K, splits), BK) * BK;
        splits = dep_cdiv(K, kchunk);
    }
    p.kchunk = kchunk; p.splits = splits; p.part = splits > 1 ? (float*)ws : nullptr;
    const bool a16 = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
    const bool b16 = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
    // contiguous-dimension divisibility: K for K-contiguous operands, M/N for MN-contiguous ones
    const bool adim = transA ? (M % 4 == 0) : (K % 4 == 0);
    const bool bdim = transB ? (K % 4 == 0) : (N % 4 == 0);
    const bool vec = a16 && b16 && adim && bdim;
    p.gx = dep_cdiv(N, BN); p.gy = dep_cdiv(M, BM);
    dim3 g(p.gx * p.gy * splits);
    DepProfScope prof(transA ? DEP_PROF_GEMM_TN : (transB ? DEP_PROF_GEMM_NT : DEP_PROF_GEMM_NN), s, dep_gemm_predicate() == nullptr);
    init_split_mode();
    if (g_force_exact == 2 || (g_force_exact == 0 && g_split_mode >= 1 && (long)M * N * K >= g_split_min_macs))
        return dep_gemm_bf16x3_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, beta, seq_T, shiftB,
                                      splits, kchunk, p.part, vec, s, (g_force_exact != 2 && g_split_mode >= 2) ? 1 : 3,
                                      (splits == 1 && ws) ? ws : g_scratch, (splits == 1 && ws) ? ws_bytes : g_scratch_bytes);      // an unsplit call may use its workspace as scratch     // the public bf16x3 entry is always 3 terms
#define LAUNCH(TA, TB)                                                                    \
    do {                                                                                   \
        if (vec) DEP_LAUNCH((gemm_mfma<TA, TB, true>), g, dim3(NT), 0, s, p);      \
        else     DEP_LAUNCH((gemm_mfma<TA, TB, false>), g, dim3(NT), 0, s, p);     \
    } while (0)
    if (!transA && transB) LAUNCH(false, true);
    else if (!transA && !transB) LAUNCH(false, false);
    else LAUNCH(true, false);
#undef LAUNCH
    DEP_CHECK_LAUNCH();
    if (splits > 1) {
        const long n = (long)M * N;
        DEP_LAUNCH(splitk_reduce, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, p.only_if, p.part, splits, M, N, C, ldc,
                           bias, beta);
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}

// Two weight-gradient contractions of equal shape in one launch (A operands: the PK gate-gradient image):
//   C0 (M x N) = A0^T shift0(B0)      C1 (M x N) = A1'^T shift1(B1)      (A' = A through the column skip)
// GRU layer: dW_ih and dW_hh (A1 = A0: the shared [dr | dz] columns are fetched once); BiLSTM layer: dW_hh of the two directions (each alone fills
// only half of the persistent grid).
// Returns 1 when the pair was enqueued, 0 when this configuration is not covered (the caller then issues the two calls itself), < 0 on error.
// Split count, K chunks and summation order per problem are those dep_gemm_internal would use: bit-identical results.
int dep_gemm_tn_pair(int M, int N, int K, const float* A0, const float* A1, int lda, int skip_at1, int skip_by1,
                     const float* B0, int ldb0, int seq_T0, int shift0, const float* B1, int ldb1, int seq_T1, int shift1,
                     float* C0, int ldc0, float* C1, int ldc1, void* ws, size_t ws_bytes, hipStream_t s) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("DEP_DW_PAIR"); on = (e && e[0] == '0') ? 0 : 1; }
    init_split_mode();
    if (!on || naive_forced() || g_force_exact != 0 || g_split_mode != 1 || (long)M * N * K < g_split_min_macs || M < 512 || dep_gemm_predicate()) return 0;
    if (!dep_gemm_pk_pending() || !dep_gemm_bf16x3_pair_ok()) return 0;
    auto a16 = [](const void* q, int ld) { return ((uintptr_t)q % 16 == 0) && (ld % 4 == 0); };
    if (!(a16(A0, lda) && a16(A1, lda) && a16(B0, ldb0) && a16(B1, ldb1) && M % 4 == 0 && N % 4 == 0)) return 0;
    int splits = choose_splits(M, N, K);
    if (splits <= 1 || !ws) return 0;
    const size_t one = dep_align((size_t)splits * M * N * sizeof(float));
    if (ws_bytes < 2 * one) return 0;                 // (a smaller workspace would change the split count of one problem: not bit-identical any more)
    const int kchunk = dep_cdiv(dep_cdiv(K, splits), BK) * BK;
    splits = dep_cdiv(K, kchunk);
    DepProfScope prof(DEP_PROF_GEMM_TN, s, true);
    const int rc = dep_gemm_bf16x3_tn_pair_launch(M, N, K, A0, A1, lda, skip_at1, skip_by1, B0, ldb0, seq_T0, shift0, B1, ldb1, seq_T1, shift1,
                                                  C0, ldc0, C1, ldc1, splits, kchunk, (float*)ws, (float*)((char*)ws + one), s);
    return rc == DEP_OK ? 1 : rc;
}

extern "C" int dep_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda,
                            const float* B, int ldb, float* C, int ldc, const float* bias, float beta,
                            int seq_T, int shiftB, void* workspace, size_t workspace_bytes, void* stream) {
    g_force_exact = 1;                  // the public fp32 entry is always exact
    const int rc = dep_gemm_internal(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, beta, seq_T, shiftB,
                                     workspace, workspace_bytes, (hipStream_t)stream);
    g_force_exact = 0;
    return rc;
}

// Same contract, precision as dep_rnn_forward / _backward would choose for a contraction of this size under the current
// dep_set_gemm_mode (0 exact, 1 three-term split, 2 single bf16 products) -- the mode-following entry.
extern "C" int dep_gemm(int transA, int transB, int M, int N, int K, const float* A, int lda,
                        const float* B, int ldb, float* C, int ldc, const float* bias, float beta,
                        int seq_T, int shiftB, void* workspace, size_t workspace_bytes, void* stream) {
    return dep_gemm_internal(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, beta, seq_T, shiftB,
                             workspace, workspace_bytes, (hipStream_t)stream);
}

// Same contract, 3-term bf16 split products (fp32 operands and accumulation): see gemm_bf16x3.hip.
extern "C" int dep_gemm_bf16x3(int transA, int transB, int M, int N, int K, const float* A, int lda,
                               const float* B, int ldb, float* C, int ldc, const float* bias, float beta,
                               int seq_T, int shiftB, void* workspace, size_t workspace_bytes, void* stream) {
    g_force_exact = 2;                  // thread-local: the process-global mode is not touched (reentrancy)
    const int rc = dep_gemm_internal(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, beta, seq_T, shiftB,
                                     workspace, workspace_bytes, (hipStream_t)stream);
    g_force_exact = 0;
    return rc;
}
