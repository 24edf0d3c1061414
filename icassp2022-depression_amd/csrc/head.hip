// The MLP head of the three models as three launches instead of fifteen (SURVEY 8b `dep_head_*`):
//     [Dropout(p)] -> Linear(Hin, H1) -> ReLU -> Dropout(p) -> [Linear(H1, C)]
// (AudioBiLSTM.fc_audio, Classification/audio_gru_whole.py:66-73 / Regression/audio_bilstm_perm.py:60-67; TextBiLSTM.fc_out,
// Classification/text_bilstm_whole.py:60-66).  At B = 512 the composed form (dropout, 2 + 4 small GEMMs, ReLU/dropout forward and
// backward, two bias column sums) is ~95 us of a 4.3 ms step, nearly all of it latency: the arithmetic is 0.2 GFLOP, but every
// one of the fifteen kernels starts with a cold fetch of its operands.  What these kernels are built around is therefore
// "every global load of a workgroup is issued before anything waits": a first version that walked W1 in eight LDS panels (one
// load round trip each) and the batch in a 512-step loop took 19 + 13 + 208 us.
// Everything is exact fp32 FMA in a fixed order (deterministic, like the GEMMs it replaces); the dropout masks are the same
// Philox draws the stand-alone kernels make (element index = row * width + column).
//
//   forward        : one 8-wave workgroup per 4 rows; thread (j, half) holds half of row j of W1 in registers (all its loads
//                    in flight while the input rows are dropped and staged in LDS), the halves meet through LDS; the second
//                    Linear is one wave per row.
//   backward, rows : dz1 = relu'/dropout(dz2 W2) kept for the weight pass; dx = dz1 W1 with wave w holding rows
//                    [w H1/8, (w+1) H1/8) of W1 as 16-byte column pieces, eight partial sums per output added in wave order.
//   backward, W    : dW1 = dz1^T a0 (4 output rows per workgroup, the batch split over the eight waves, 16-byte loads of a0,
//                    the four dz1 columns staged in LDS), db1 beside it; the last workgroups make dW2 = dz2^T a1 and db2
//                    with the same code.
#include "dep_common.h"

namespace {

constexpr int HR = 4;            // rows per workgroup of the row passes
constexpr int HT = 512;          // threads per workgroup (8 waves)
constexpr int HMAX = 256;        // widest layer handled here; wider heads take the composed path
constexpr int HC = 16;           // most output columns (= MAXC of dep_head_loss)
constexpr int JR = 4;            // output rows per workgroup of the weight pass
constexpr int BCH = 1024;        // batch rows staged per round of the weight pass

struct HeadF {
    const float *x, *W1, *b1, *W2, *b2;
    float *a0, *z1, *a1, *z2;
    int B, Hin, H1, C;
    float p, scale;
    uint64_t seed;
    uint32_t site0, site1;
    int first;
};

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// NV = Hin / 8: 16-byte pieces of W1 a thread holds (its half of row j)
template <int NV>
__global__ __launch_bounds__(HT) void head_mlp_fwd_kernel(HeadF q) {
    __shared__ __attribute__((aligned(16))) float xs[HR][HMAX];
    __shared__ __attribute__((aligned(16))) float part[HR][HMAX];
    __shared__ __attribute__((aligned(16))) float as1[HR][HMAX];
    constexpr int Hin = NV * 8, KH = Hin / 2;
    const int tid = threadIdx.x, row0 = blockIdx.x * HR;
    const int j = tid & (HMAX - 1), half = tid >> 8;
    const bool mine = j < q.H1;
    f32x4 w[NV];
    {
        const float* wr = q.W1 + (size_t)(mine ? j : 0) * Hin + half * KH;
#pragma unroll
        for (int m = 0; m < NV; ++m) w[m] = *reinterpret_cast<const f32x4*>(wr + 4 * m);
    }
    const bool drop0 = q.first && q.p > 0.f;
    constexpr int NX = (HR * Hin + HT - 1) / HT;
    float xv[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {                                 // loads first (clamped), so that they fly together
        const int e = tid + k * HT, r = e / Hin, row = row0 + r;
        const bool ok = e < HR * Hin && row < q.B;
        xv[k] = q.x[ok ? (size_t)row * Hin + (e - r * Hin) : 0];
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int e = tid + k * HT, r = e / Hin, i = e - r * Hin, row = row0 + r;
        if (e >= HR * Hin) continue;
        float v = 0.f;
        if (row < q.B) {
            const size_t at = (size_t)row * Hin + i;
            v = xv[k];
            if (drop0) { v *= dep_dropmask1(q.seed, q.site0, (uint64_t)at, q.p, q.scale); q.a0[at] = v; }
        }
        xs[r][i] = v;
    }
    __syncthreads();
    float acc[HR] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < NV; ++m) {
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[r][half * KH + 4 * m]);
            acc[r] = fmaf(xv[0], w[m][0], acc[r]); acc[r] = fmaf(xv[1], w[m][1], acc[r]);
            acc[r] = fmaf(xv[2], w[m][2], acc[r]); acc[r] = fmaf(xv[3], w[m][3], acc[r]);
        }
    }
    if (half == 1) {
#pragma unroll
        for (int r = 0; r < HR; ++r) part[r][j] = acc[r];
    }
    __syncthreads();
    if (half == 0 && mine) {
        const float bj = q.b1[j];
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            const int row = row0 + r;
            float a = 0.f;
            if (row < q.B) {
                const size_t at = (size_t)row * q.H1 + j;
                const float z = (acc[r] + part[r][j]) + bj;
                a = fmaxf(z, 0.f);
                if (q.p > 0.f) a *= dep_dropmask1(q.seed, q.site1, (uint64_t)at, q.p, q.scale);
                q.z1[at] = z; q.a1[at] = a;
            }
            as1[r][j] = a;
        }
    }
    if (q.C <= 0) return;
    __syncthreads();
    const int wv = tid >> 6, lane = tid & 63, row = row0 + wv;
    if (wv >= HR || row >= q.B) return;
    for (int c = 0; c < q.C; ++c) {
        float s = 0.f;
        for (int k = lane; k < q.H1; k += 64) s = fmaf(as1[wv][k], q.W2[(size_t)c * q.H1 + k], s);
        s = wsum(s);
        if (lane == 0) q.z2[(size_t)row * q.C + c] = s + q.b2[c];
    }
}

struct HeadB {
    const float *dz2, *a0, *z1, *a1, *W1, *W2;
    float *dW1, *db1, *dW2, *db2, *dx, *dz1;
    int B, Hin, H1, C;
    float p, scale;
    uint64_t seed;
    uint32_t site0, site1;
    int first;
};

// NJ = H1 / 8: rows of W1 a wave holds
template <int NJ>
__global__ __launch_bounds__(HT) void head_mlp_bwd_rows_kernel(HeadB q) {
    __shared__ __attribute__((aligned(16))) float dzs[HR][HMAX];
    __shared__ __attribute__((aligned(16))) float ps[HT / 64][HR][HMAX];
    constexpr int H1 = NJ * 8;
    const int tid = threadIdx.x, row0 = blockIdx.x * HR, wv = tid >> 6, lane = tid & 63, i4 = lane * 4;
    const bool want_dx = q.dx != nullptr, col = i4 < q.Hin;
    f32x4 w[NJ];
    if (want_dx) {
        const float* wr = q.W1 + (size_t)(wv * NJ) * q.Hin + (col ? i4 : 0);
#pragma unroll
        for (int m = 0; m < NJ; ++m) w[m] = *reinterpret_cast<const f32x4*>(wr + (size_t)m * q.Hin);
    }
    constexpr int NX = (HR * H1 + HT - 1) / HT;
    float zv[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int e = tid + k * HT, r = e / H1, row = row0 + r;
        const bool ok = e < HR * H1 && row < q.B;
        zv[k] = q.z1[ok ? (size_t)row * H1 + (e - r * H1) : 0];
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int e = tid + k * HT, r = e / H1, j = e - r * H1, row = row0 + r;
        if (e >= HR * H1) continue;
        float v = 0.f;
        if (row < q.B) {
            const size_t at = (size_t)row * H1 + j;
            float da = 0.f;
            for (int c = 0; c < q.C; ++c) da = fmaf(q.dz2[(size_t)row * q.C + c], q.W2[(size_t)c * H1 + j], da);
            v = zv[k] > 0.f ? da : 0.f;
            if (q.p > 0.f) v *= dep_dropmask1(q.seed, q.site1, (uint64_t)at, q.p, q.scale);
            q.dz1[at] = v;
        }
        dzs[r][j] = v;
    }
    if (!want_dx) return;
    __syncthreads();
    f32x4 acc[HR];
#pragma unroll
    for (int r = 0; r < HR; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < NJ; ++m) {
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            const float d = dzs[r][wv * NJ + m];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][e] = fmaf(d, w[m][e], acc[r][e]);
        }
    }
#pragma unroll
    for (int r = 0; r < HR; ++r) *reinterpret_cast<f32x4*>(&ps[wv][r][i4]) = acc[r];
    __syncthreads();
    const bool drop0 = q.first && q.p > 0.f;
    for (int e = tid; e < HR * q.Hin; e += HT) {
        const int r = e / q.Hin, i = e - r * q.Hin, row = row0 + r;
        if (row >= q.B) continue;
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < HT / 64; ++u) v += ps[u][r][i];
        const size_t at = (size_t)row * q.Hin + i;
        if (drop0) v *= dep_dropmask1(q.seed, q.site0, (uint64_t)at, q.p, q.scale);
        q.dx[at] = v;
    }
}

// out[row0 + u][i] = sum_b D[b][row0 + u] A[b][i] ,  bias[row0 + u] = sum_b D[b][row0 + u]     (u < JR, i < ncols <= HMAX)
__device__ __forceinline__ void weight_rows(const float* __restrict__ D, int ldD, int nrows, int row0, const float* __restrict__ A,
                                            int ldA, int ncols, int B, float* out, float* bias) {
    __shared__ __attribute__((aligned(16))) float ds[BCH][JR];
    __shared__ __attribute__((aligned(16))) float ps[HT / 64][JR][HMAX];
    __shared__ float pb[HT / 64][JR];
    static_assert(JR == 4, "one 16-byte broadcast per batch row");
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, i4 = lane * 4;
    const bool col = i4 < ncols;
    const float* Ac = A + (col ? i4 : 0);
    f32x4 acc[JR];
    float sb[JR];
#pragma unroll
    for (int u = 0; u < JR; ++u) { acc[u] = f32x4{0.f, 0.f, 0.f, 0.f}; sb[u] = 0.f; }
    for (int b0 = 0; b0 < B; b0 += BCH) {
        const int nb = (B - b0 < BCH) ? B - b0 : BCH;
        __syncthreads();
        for (int e0 = 0; e0 < nb * JR; e0 += HT * 4) {
            float dv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {                          // clamped, unconditional: four loads in flight
                int e = e0 + tid + k * HT; e = e < nb * JR ? e : nb * JR - 1;
                const int bb = e / JR, u = e - bb * JR;
                dv[k] = D[(size_t)(b0 + bb) * ldD + (row0 + u < nrows ? row0 + u : nrows - 1)];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + tid + k * HT;
                if (e < nb * JR) { const int bb = e / JR, u = e - bb * JR; ds[bb][u] = (row0 + u < nrows) ? dv[k] : 0.f; }
            }
        }
        __syncthreads();
        const int per = (nb + HT / 64 - 1) / (HT / 64);
        const int lo = wv * per, hi = (lo + per < nb) ? lo + per : nb;
        constexpr int NB = 16;                                     // a0 rows in flight per thread
        for (int bb = lo; bb < hi; bb += NB) {
            f32x4 av[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int b = bb + k < hi ? bb + k : hi - 1;
                av[k] = *reinterpret_cast<const f32x4*>(Ac + (size_t)(b0 + b) * ldA);
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                f32x4 d = *reinterpret_cast<const f32x4*>(&ds[bb + k < hi ? bb + k : hi - 1][0]);
                if (bb + k >= hi) d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < JR; ++u) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[u][e] = fmaf(d[u], av[k][e], acc[u][e]);
                    sb[u] += d[u];
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < JR; ++u) {
        *reinterpret_cast<f32x4*>(&ps[wv][u][i4]) = acc[u];
        if (lane == 0) pb[wv][u] = sb[u];
    }
    __syncthreads();
    for (int e = tid; e < JR * HMAX; e += HT) {
        const int u = e / HMAX, i = e - u * HMAX;
        if (i >= ncols || row0 + u >= nrows) continue;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < HT / 64; ++k) v += ps[k][u][i];
        out[(size_t)(row0 + u) * ncols + i] = v;
    }
    if (tid < JR && row0 + tid < nrows) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < HT / 64; ++k) v += pb[k][tid];
        bias[row0 + tid] = v;
    }
}

__global__ __launch_bounds__(HT) void head_mlp_bwd_w_kernel(HeadB q) {
    const int n1 = (q.H1 + JR - 1) / JR;
    if ((int)blockIdx.x < n1) weight_rows(q.dz1, q.H1, q.H1, blockIdx.x * JR, q.a0, q.Hin, q.Hin, q.B, q.dW1, q.db1);
    else weight_rows(q.dz2, q.C, q.C, ((int)blockIdx.x - n1) * JR, q.a1, q.H1, q.H1, q.B, q.dW2, q.db2);
}

bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
bool width_ok(int h) { return h >= 8 && h <= HMAX && (h & (h - 1)) == 0; }

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int dep_head_mlp_supported(int Hin, int H1, int C) { return width_ok(Hin) && width_ok(H1) && C >= 0 && C <= HC; }

extern "C" int dep_head_mlp_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, float* a0,
                                float* z1, float* a1, float* z2, int B, int Hin, int H1, int C, float p, uint64_t seed,
                                uint32_t site0, uint32_t site1, int first_dropout, void* stream) {
    DEP_CHECK_ARG(x && W1 && b1 && z1 && a1 && B > 0 && p >= 0.f && p < 1.f);
    DEP_CHECK_ARG(C == 0 || (W2 && b2 && z2));
    DEP_CHECK_ARG(!(first_dropout && p > 0.f) || a0);
    if (!dep_head_mlp_supported(Hin, H1, C) || !al16(W1)) { dep_set_error("dep_head_mlp_fwd: shape not covered (see dep_head_mlp_supported)"); return DEP_ERR_ARG; }
    HeadF q{x, W1, b1, W2, b2, a0, z1, a1, z2, B, Hin, H1, C, p, 1.0f / (1.0f - p), seed, site0, site1, first_dropout};
    const dim3 g(dep_cdiv(B, HR)), b(HT);
    switch (Hin / 8) {
        case 1: DEP_LAUNCH(head_mlp_fwd_kernel<1>, g, b, 0, S_, q); break;
        case 2: DEP_LAUNCH(head_mlp_fwd_kernel<2>, g, b, 0, S_, q); break;
        case 4: DEP_LAUNCH(head_mlp_fwd_kernel<4>, g, b, 0, S_, q); break;
        case 8: DEP_LAUNCH(head_mlp_fwd_kernel<8>, g, b, 0, S_, q); break;
        case 16: DEP_LAUNCH(head_mlp_fwd_kernel<16>, g, b, 0, S_, q); break;
        default: DEP_LAUNCH(head_mlp_fwd_kernel<32>, g, b, 0, S_, q); break;
    }
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" int dep_head_mlp_bwd(const float* dz2, const float* a0, const float* z1, const float* a1, const float* W1,
                                const float* W2, float* dW1, float* db1, float* dW2, float* db2, float* dx, float* dz1, int B,
                                int Hin, int H1, int C, float p, uint64_t seed, uint32_t site0, uint32_t site1,
                                int first_dropout, void* stream) {
    DEP_CHECK_ARG(dz2 && a0 && z1 && a1 && W1 && W2 && dW1 && db1 && dW2 && db2 && dz1 && B > 0 && C > 0 && p >= 0.f && p < 1.f);
    if (!dep_head_mlp_supported(Hin, H1, C) || !al16(W1) || !al16(a0) || !al16(a1)) {
        dep_set_error("dep_head_mlp_bwd: shape not covered (see dep_head_mlp_supported)"); return DEP_ERR_ARG;
    }
    HeadB q{dz2, a0, z1, a1, W1, W2, dW1, db1, dW2, db2, dx, dz1, B, Hin, H1, C, p, 1.0f / (1.0f - p), seed, site0, site1, first_dropout};
    const dim3 g(dep_cdiv(B, HR)), b(HT);
    switch (H1 / 8) {
        case 1: DEP_LAUNCH(head_mlp_bwd_rows_kernel<1>, g, b, 0, S_, q); break;
        case 2: DEP_LAUNCH(head_mlp_bwd_rows_kernel<2>, g, b, 0, S_, q); break;
        case 4: DEP_LAUNCH(head_mlp_bwd_rows_kernel<4>, g, b, 0, S_, q); break;
        case 8: DEP_LAUNCH(head_mlp_bwd_rows_kernel<8>, g, b, 0, S_, q); break;
        case 16: DEP_LAUNCH(head_mlp_bwd_rows_kernel<16>, g, b, 0, S_, q); break;
        default: DEP_LAUNCH(head_mlp_bwd_rows_kernel<32>, g, b, 0, S_, q); break;
    }
    DEP_LAUNCH(head_mlp_bwd_w_kernel, dim3(dep_cdiv(H1, JR) + dep_cdiv(C, JR)), b, 0, S_, q);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
