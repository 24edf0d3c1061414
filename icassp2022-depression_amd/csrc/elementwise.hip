// Memory-bound and small kernels of the hot path: LayerNorm, dropout, ReLU, bias-gradient column
// sums, the fused output-nonlinearity + loss + gradient kernel, attention_net_with_w, Adam/AdamW.
// Reference call sites are cited on each entry point in include/dep_rnn.h.
#include "dep_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
// block-wide sum / max through LDS scratch `red` (>= 16 floats); every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
    for (int i = 1; i < nw; ++i) s = fmaxf(s, red[i]);
    return s;
}

// ------------------------------------------------------------------------------ LayerNorm
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mr, int rows, int F, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * F;
    float s = 0.f;
    for (int i = lane; i < F; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)F;
    float v = 0.f;
    for (int i = lane; i < F; i += 64) { const float d = xr[i] - mean; v = fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)F + eps);
    float* yr = y + row * F;
    if (gamma) { for (int i = lane; i < F; i += 64) yr[i] = (xr[i] - mean) * rstd * gamma[i] + beta[i]; }
    else       { for (int i = lane; i < F; i += 64) yr[i] = (xr[i] - mean) * rstd; }       // plain x-hat (dep_ln_fold_*)
    if (mr && lane == 0) { mr[row * 2] = mean; mr[row * 2 + 1] = rstd; }
}

// F = 256*NV: a row lives in NV float4 per lane -- one pass over HBM, 16-byte accesses
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y,
                                                         float* __restrict__ mr, int rows, float eps) {
    constexpr int F = 256 * NV;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * F);
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = xr[i * 64 + lane]; s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    const float mean = wave_sum(s) / (float)F;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)F + eps);
    f32x4* yr = reinterpret_cast<f32x4*>(y + row * F);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f32x4 o = (v[i] - mean) * rstd;
        if (gamma) {
            o = o * reinterpret_cast<const f32x4*>(gamma)[i * 64 + lane] + reinterpret_cast<const f32x4*>(beta)[i * 64 + lane];
        }
        yr[i * 64 + lane] = o;
    }
    if (mr && lane == 0) { mr[row * 2] = mean; mr[row * 2 + 1] = rstd; }
}

// LayerNorm's affine folded into the FIRST linear map that consumes it (dep_ln_fold_fwd / dep_ln_fold_bwd):
//     (xhat*gamma + beta) W^T + b  ==  xhat (W*gamma)^T + (b + W beta)
// forward : Wf[j,f] = W[j,f] gamma[f] ,  bf[j] = b[j] + sum_f W[j,f] beta[f]            (one wave per row j)
__global__ __launch_bounds__(256) void ln_fold_fwd_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ Wf, float* __restrict__ bf, int J, int F) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= J) return;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) {
        const float w = W[(size_t)j * F + f];
        Wf[(size_t)j * F + f] = w * gamma[f];
        s = fmaf(w, beta[f], s);
    }
    s = wave_sum(s);
    if (lane == 0) bf[j] = b[j] + s;
}
// backward: given P = dL/dWf (J,F) and q = dL/dbf (J):
//     dW[j,f] = P[j,f] gamma[f] + q[j] beta[f] ,  db = q ,  dgamma[f] = sum_j P[j,f] W[j,f] ,  dbeta[f] = sum_j W[j,f] q[j]
// block = 16 columns x 64 row groups (1024 threads): 64-byte row segments, 64 partial sums per column reduced in LDS.  (With
// 64 columns x 16 row groups only F / 64 = 4 CUs moved the 2.3 MB of the cfg2 fold: 14 us, CU-bandwidth bound.)
constexpr int LFC = 16, LFR = 64;
__global__ __launch_bounds__(1024) void ln_fold_bwd_kernel(const float* __restrict__ W, const float* __restrict__ P,
                                                           const float* __restrict__ q, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ dW,
                                                           float* __restrict__ db, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int J, int F) {
    __shared__ float red[2][LFR][LFC];
    const int c = threadIdx.x % LFC, rg = threadIdx.x / LFC;
    const int f = blockIdx.x * LFC + c;
    float dg = 0.f, dbt = 0.f;
    if (f < F) {
        const float g = gamma[f], bt = beta[f];
#pragma unroll 4
        for (int j = rg; j < J; j += LFR) {
            const float w = W[(size_t)j * F + f], pj = P[(size_t)j * F + f], qj = q[j];
            dW[(size_t)j * F + f] = fmaf(pj, g, qj * bt);
            dg = fmaf(pj, w, dg);
            dbt = fmaf(w, qj, dbt);
        }
    }
    red[0][rg][c] = dg; red[1][rg][c] = dbt;
    __syncthreads();
    if (rg < 2 && f < F) {                                         // row group 0 sums dgamma, row group 1 dbeta
        float a = 0.f;
#pragma unroll 8
        for (int r = 0; r < LFR; ++r) a += red[rg][r][c];
        (rg == 0 ? dgamma : dbeta)[f] = a;
    }
    if (blockIdx.x == 0) for (int j = threadIdx.x; j < J; j += 1024) db[j] = q[j];
}

// partial[blk][2][F]: per-block column sums of dy*xhat and dy over the block's row range
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mr, float* __restrict__ partial,
                                                           int rows, int F, int rows_per_blk) {
    const long r0 = (long)blockIdx.x * rows_per_blk;
    const long r1 = min((long)rows, r0 + rows_per_blk);
    for (int c = threadIdx.x; c < F; c += blockDim.x) {
        float dg = 0.f, db = 0.f;
        for (long r = r0; r < r1; ++r) {
            const float mean = mr[r * 2], rstd = mr[r * 2 + 1];
            const float g = dy[r * F + c];
            dg = fmaf(g, (x[r * F + c] - mean) * rstd, dg);
            db += g;
        }
        partial[((size_t)blockIdx.x * 2) * F + c] = dg;
        partial[((size_t)blockIdx.x * 2 + 1) * F + c] = db;
    }
}
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ mr,
                                                        float* __restrict__ dx, int rows, int F) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float mean = mr[row * 2], rstd = mr[row * 2 + 1];
    const float* xr = x + row * F; const float* gr = dy + row * F;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < F; i += 64) {
        const float g = gr[i] * gamma[i];
        s1 += g; s2 = fmaf(g, (xr[i] - mean) * rstd, s2);
    }
    s1 = wave_sum(s1) / (float)F; s2 = wave_sum(s2) / (float)F;
    for (int i = lane; i < F; i += 64) {
        const float g = gr[i] * gamma[i], xh = (xr[i] - mean) * rstd;
        dx[row * F + i] = rstd * (g - s1 - xh * s2);
    }
}

// ------------------------------------------------------------------------------ dropout / relu
__global__ void dropout_kernel(const float* x, float* y, long n, float p, float scale, uint64_t seed, uint32_t site) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = x[i] * dep_dropmask1(seed, site, (uint64_t)i, p, scale);
}
__global__ void dropout_mask_kernel(float* m, long n, float p, float scale, uint64_t seed, uint32_t site) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    m[i] = dep_dropmask1(seed, site, (uint64_t)i, p, scale);
}
__global__ void relu_dropout_fwd_kernel(const float* z, float* a, long n, float p, float scale, uint64_t seed, uint32_t site) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = fmaxf(z[i], 0.f);
    if (p > 0.f) v *= dep_dropmask1(seed, site, (uint64_t)i, p, scale);
    a[i] = v;
}
__global__ void relu_dropout_bwd_kernel(const float* da, const float* z, float* dz, long n, float p, float scale,
                                        uint64_t seed, uint32_t site) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = z[i] > 0.f ? da[i] : 0.f;
    if (p > 0.f) v *= dep_dropmask1(seed, site, (uint64_t)i, p, scale);
    dz[i] = v;
}

// out[n] = sum_m x[m*ld + n] ; block = 64 columns x 4 row-groups
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int M, int N, int ld, float* out) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    // eight loads in flight per thread (one dependent chain of M/4 loads took 12 us for the head's 512-row bias gradients)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    if (c < N) {
        int m = rg;
        for (; m + 28 < M; m += 32) {
            const float* q = x + (size_t)m * ld + c;
            s0 += q[0]; s1 += q[(size_t)4 * ld]; s2 += q[(size_t)8 * ld]; s3 += q[(size_t)12 * ld];
            s4 += q[(size_t)16 * ld]; s5 += q[(size_t)20 * ld]; s6 += q[(size_t)24 * ld]; s7 += q[(size_t)28 * ld];
        }
        for (; m < M; m += 4) s0 += x[(size_t)m * ld + c];
    }
    const float s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < N) out[c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ------------------------------------------------------------------------------ head + loss
constexpr int MAXC = 16;
__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }

__global__ void head_loss_kernel(int kind, const float* __restrict__ z, const void* __restrict__ target,
                                 float* __restrict__ out, float* __restrict__ loss_rows, float* __restrict__ dz,
                                 int B, int C, float inv_norm) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool wide = (kind & DEP_LOSS_LABELS_I64) != 0;       // class labels as torch.long, read in place
    kind &= ~DEP_LOSS_LABELS_I64;
    const float* zr = z + (size_t)b * C;
    float v[MAXC];
    if (kind == DEP_LOSS_CE_ON_SOFTMAX || kind == DEP_LOSS_CE_LOGITS) {
        float mx = zr[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, zr[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) { v[c] = expf(zr[c] - mx); s += v[c]; }
        for (int c = 0; c < C; ++c) { v[c] /= s; if (out) out[(size_t)b * C + c] = v[c]; }   // softmax(z)
        if (!target) return;
        const int y = wide ? (int)((const int64_t*)target)[b] : ((const int32_t*)target)[b];
        if (kind == DEP_LOSS_CE_LOGITS) {
            if (loss_rows) loss_rows[b] = -((zr[y] - mx) - logf(s));
            if (dz) for (int c = 0; c < C; ++c) dz[(size_t)b * C + c] = (v[c] - (c == y ? 1.f : 0.f)) * inv_norm;
        } else {
            // CrossEntropyLoss applied to the probabilities: a second softmax over p
            float pm = v[0];
            for (int c = 1; c < C; ++c) pm = fmaxf(pm, v[c]);
            float s2 = 0.f; float e2[MAXC];
            for (int c = 0; c < C; ++c) { e2[c] = expf(v[c] - pm); s2 += e2[c]; }
            if (loss_rows) loss_rows[b] = -((v[y] - pm) - logf(s2));
            if (dz) {
                float dp[MAXC]; float dot = 0.f;
                for (int c = 0; c < C; ++c) { dp[c] = (e2[c] / s2 - (c == y ? 1.f : 0.f)) * inv_norm; dot = fmaf(dp[c], v[c], dot); }
                for (int c = 0; c < C; ++c) dz[(size_t)b * C + c] = v[c] * (dp[c] - dot);
            }
        }
        return;
    }
    const bool relu = (kind == DEP_LOSS_L1_RELU || kind == DEP_LOSS_SMOOTHL1_RELU);
    const bool l1 = (kind == DEP_LOSS_L1_RELU);
    float lsum = 0.f;
    for (int c = 0; c < C; ++c) {
        const float zz = zr[c];
        const float o = relu ? fmaxf(zz, 0.f) : zz;
        if (out) out[(size_t)b * C + c] = o;
        if (!target) continue;
        const float d = o - ((const float*)target)[(size_t)b * C + c];
        const float a = fabsf(d);
        float g;
        if (l1) { lsum += a; g = sgn(d); }
        else if (a < 1.0f) { lsum += 0.5f * d * d; g = d; }
        else { lsum += a - 0.5f; g = sgn(d); }
        if (dz) dz[(size_t)b * C + c] = (relu && !(zz > 0.f)) ? 0.f : g * inv_norm;
    }
    if (target && loss_rows) loss_rows[b] = lsum;
}

__global__ __launch_bounds__(256) void reduce_loss_kernel(const float* __restrict__ rows, int B, float inv_norm,
                                                          float* out, int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s += rows[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + s * inv_norm;
}

// ------------------------------------------------------------------------------ Adam / AdamW
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                            int decoupled, float step_size, float inv_sqrt_bc2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float pv = p[i], gv = g[i];
    if (decoupled) pv *= (1.0f - lr * wd);
    else if (wd != 0.f) gv = fmaf(wd, pv, gv);
    const float mv = b1 * m[i] + (1.0f - b1) * gv;
    const float vv = b2 * v[i] + (1.0f - b2) * gv * gv;
    m[i] = mv; v[i] = vv;
    const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
    p[i] = pv - step_size * (mv / denom);
}

__global__ void fill_kernel(float* p, long n, float v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void axpby_kernel(const float* x, float* y, long n, float a, float b) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}
// several small copies (optionally sums of two sources) in ONE launch: blockIdx.y = job.  The bidirectional stacks gather
// their direction-stacked weights / folded biases with it: twelve 5 us launches per forward otherwise.
struct CopyJobs { const float* src[16]; const float* add[16]; float* dst[16]; long n[16]; int count; };
__global__ void multi_copy_kernel(CopyJobs j) {
    const int k = blockIdx.y;
    if (k >= j.count) return;
    const float* __restrict__ s = j.src[k]; const float* __restrict__ a = j.add[k]; float* __restrict__ d = j.dst[k];
    const long n = j.n[k], stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = a ? s[i] + a[i] : s[i];
}
__global__ void sigmoid_gate_kernel(const float* g, const float* x, float* y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = dep_sigmoid(g[i]) * x[i];
}

// ------------------------------------------------------------------------------ host-loop helpers (round 4)
// The training loops' bookkeeping that used to run as ATen kernels (index_select, cat, max / eq / sum): row gather of a
// mini-batch out of the HBM-resident corpus, a strided 2-D copy (feature concat), and arg-max + correct count of the head output.
__global__ void gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx, float* __restrict__ dst,
                                   long row_floats, int vec) {
    const long r = blockIdx.y;
    const float* sr = src + (size_t)idx[r] * row_floats;
    float* dr = dst + (size_t)r * row_floats;
    const long stride = (long)gridDim.x * blockDim.x, i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {                                       // rows are 16-byte aligned multiples of four floats
        const f32x4* s = reinterpret_cast<const f32x4*>(sr); f32x4* d = reinterpret_cast<f32x4*>(dr);
        for (long i = i0; i < (row_floats >> 2); i += stride) d[i] = s[i];
    } else {
        for (long i = i0; i < row_floats; i += stride) dr[i] = sr[i];
    }
}
__global__ void copy2d_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, long rows, long cols) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols, c = i - r * cols;
        dst[r * ldd + c] = src[r * lds + c];
    }
}
// acc[0] += *loss (float64 sum of the fp32 step losses, the order Python's `total_loss += loss.item()` adds them in);
// acc[1] = max(acc[1], *status), acc[2] = max(acc[2], *soft): the sweeps' status / fallback words folded into the epoch's flags
__global__ void loss_accumulate_kernel(const float* loss, const unsigned* status, const unsigned* soft, double* acc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (loss) acc[0] += (double)loss[0];
    if (status) { const double v = (double)*status; if (v > acc[1]) acc[1] = v; }
    if (soft) { const double v = (double)*soft; if (v > acc[2]) acc[2] = v; }
}
// first maximum per row (torch.max(1) on the reference's CPU path returns the first index among equals)
__global__ void argmax_count_kernel(const float* __restrict__ p, const void* __restrict__ labels, int i64, int B, int C,
                                    long long* __restrict__ count, long long* __restrict__ pred) {
    __shared__ int wsum[4];
    int ok = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* row = p + (size_t)b * C;
        int best = 0; float bv = row[0];
        for (int c = 1; c < C; ++c) { const float v = row[c]; if (v > bv) { bv = v; best = c; } }
        if (pred) pred[b] = best;
        if (labels) {
            const long long y = i64 ? reinterpret_cast<const long long*>(labels)[b] : (long long)reinterpret_cast<const int*>(labels)[b];
            ok += (y == best) ? 1 : 0;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ok += __shfl_xor(ok, m, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ok;
    __syncthreads();
    if (threadIdx.x == 0 && count) *count += (long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// ------------------------------------------------------------------------------ attention
__global__ void attn_hsum_kernel(const float* __restrict__ hn, int K, long BH, float* __restrict__ hsum) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BH) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += hn[(size_t)k * BH + i];
    hsum[i] = s;
}
__global__ void attn_bcast_kernel(const float* __restrict__ src, int K, long BH, float* __restrict__ dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BH) return;
    const float s = src[i];
    for (int k = 0; k < K; ++k) dst[(size_t)k * BH + i] = s;
}

// one workgroup per utterance.  LDS: q[H] | sc[T] | red[16]
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ out, const float* __restrict__ pre,
                                                       float* __restrict__ ctx, float* __restrict__ alpha,
                                                       int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q = smem; float* sc = smem + H; float* red = sc + T;
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float* ob = out + (size_t)b * T * 2 * H;
    for (int j = threadIdx.x; j < H; j += blockDim.x) q[j] = fmaxf(pre[(size_t)b * H + j], 0.f);
    __syncthreads();
    for (int t = w; t < T; t += nw) {
        const float* r = ob + (size_t)t * 2 * H;
        float s = 0.f;
        for (int j = lane; j < H; j += 64) s = fmaf(q[j], tanhf(r[j] + r[H + j]), s);
        s = wave_sum(s);
        if (lane == 0) sc[t] = s;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < T; t += blockDim.x) mx = fmaxf(mx, sc[t]);
    mx = block_max(mx, red);
    float se = 0.f;
    for (int t = threadIdx.x; t < T; t += blockDim.x) { const float e = expf(sc[t] - mx); sc[t] = e; se += e; }
    se = block_sum(se, red);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) { const float a = sc[t] / se; sc[t] = a; alpha[(size_t)b * T + t] = a; }
    __syncthreads();
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) { const float* r = ob + (size_t)t * 2 * H; s = fmaf(sc[t], r[j] + r[H + j], s); }
        ctx[(size_t)b * H + j] = s;
    }
}

// LDS: q[H] | dc[H] | al[T] | ds[T] | red[16]
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ dctx, const float* __restrict__ out,
                                                       const float* __restrict__ alpha, const float* __restrict__ pre,
                                                       float* __restrict__ dout, float* __restrict__ dpre, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q = smem; float* dc = q + H; float* al = dc + H; float* ds = al + T; float* red = ds + T;
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float* ob = out + (size_t)b * T * 2 * H;
    float* dob = dout + (size_t)b * T * 2 * H;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        q[j] = fmaxf(pre[(size_t)b * H + j], 0.f);
        dc[j] = dctx[(size_t)b * H + j];
    }
    for (int t = threadIdx.x; t < T; t += blockDim.x) al[t] = alpha[(size_t)b * T + t];
    __syncthreads();
    for (int t = w; t < T; t += nw) {
        const float* r = ob + (size_t)t * 2 * H;
        float s = 0.f;
        for (int j = lane; j < H; j += 64) s = fmaf(dc[j], r[j] + r[H + j], s);
        s = wave_sum(s);
        if (lane == 0) ds[t] = s;                       // d alpha
    }
    __syncthreads();
    float dot = 0.f;
    for (int t = threadIdx.x; t < T; t += blockDim.x) dot = fmaf(al[t], ds[t], dot);
    dot = block_sum(dot, red);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) ds[t] = al[t] * (ds[t] - dot);   // d scores
    __syncthreads();
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        float dq = 0.f;
        const float qj = q[j], dcj = dc[j];
        for (int t = 0; t < T; ++t) {
            const float* r = ob + (size_t)t * 2 * H;
            const float m = tanhf(r[j] + r[H + j]);
            dq = fmaf(ds[t], m, dq);
            const float dh = al[t] * dcj + ds[t] * qj * (1.0f - m * m);
            dob[(size_t)t * 2 * H + j] = dh;
            dob[(size_t)t * 2 * H + H + j] = dh;
        }
        dpre[(size_t)b * H + j] = pre[(size_t)b * H + j] > 0.f ? dq : 0.f;
    }
}

inline int nblk(long n, int t = 256) { return dep_cdiv(n, t); }

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int dep_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean_rstd,
                                 int rows, int F, float eps, void* stream) {
    DEP_CHECK_ARG(x && y && rows > 0 && F > 0 && ((gamma != nullptr) == (beta != nullptr)));
    const bool al16 = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
    const dim3 g(dep_cdiv(rows, 4)), b(256);
    if (al16 && F == 256) DEP_LAUNCH(ln_fwd_vec_kernel<1>, g, b, 0, S_, x, gamma, beta, y, mean_rstd, rows, eps);
    else if (al16 && F == 512) DEP_LAUNCH(ln_fwd_vec_kernel<2>, g, b, 0, S_, x, gamma, beta, y, mean_rstd, rows, eps);
    else if (al16 && F == 1024) DEP_LAUNCH(ln_fwd_vec_kernel<4>, g, b, 0, S_, x, gamma, beta, y, mean_rstd, rows, eps);
    else DEP_LAUNCH(ln_fwd_kernel, g, b, 0, S_, x, gamma, beta, y, mean_rstd, rows, F, eps);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" int dep_ln_fold_fwd(const float* W, const float* b, const float* gamma, const float* beta, float* Wf,
                               float* bf, int J, int F, void* stream) {
    DEP_CHECK_ARG(W && b && gamma && beta && Wf && bf && J > 0 && F > 0);
    DEP_LAUNCH(ln_fold_fwd_kernel, dim3(dep_cdiv(J, 4)), dim3(256), 0, S_, W, b, gamma, beta, Wf, bf, J, F);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" int dep_ln_fold_bwd(const float* W, const float* dWf, const float* dbf, const float* gamma, const float* beta,
                               float* dW, float* db, float* dgamma, float* dbeta, int J, int F, void* stream) {
    DEP_CHECK_ARG(W && dWf && dbf && gamma && beta && dW && db && dgamma && dbeta && J > 0 && F > 0);
    DEP_LAUNCH(ln_fold_bwd_kernel, dim3(dep_cdiv(F, LFC)), dim3(1024), 0, S_, W, dWf, dbf, gamma, beta, dW, db, dgamma, dbeta, J, F);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

static int ln_bwd_blocks(int rows) { int b = dep_cdiv(rows, 64); return b > 512 ? 512 : b; }

extern "C" size_t dep_layernorm_bwd_workspace_bytes(int rows, int F) {
    return dep_align((size_t)ln_bwd_blocks(rows) * 2 * F * sizeof(float));
}

extern "C" int dep_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean_rstd,
                                 float* dx, float* dgamma, float* dbeta, int rows, int F, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    DEP_CHECK_ARG(dy && x && gamma && mean_rstd && dgamma && dbeta && rows > 0 && F > 0);
    const int nb = ln_bwd_blocks(rows);
    if (!workspace || workspace_bytes < (size_t)nb * 2 * F * sizeof(float)) {
        dep_set_error("dep_layernorm_bwd: workspace too small"); return DEP_ERR_WORKSPACE;
    }
    const int rpb = dep_cdiv(rows, nb);
    DEP_LAUNCH(ln_bwd_param_kernel, dim3(nb), dim3(256), 0, S_, dy, x, mean_rstd, (float*)workspace, rows, F, rpb);
    DEP_CHECK_LAUNCH();
    // partial is [nb][2F]: column sums of its two halves (parallel over row groups, unlike the old serial finish)
    DEP_LAUNCH(colsum_kernel, dim3(dep_cdiv(F, 64)), dim3(256), 0, S_, (const float*)workspace, nb, F, 2 * F, dgamma);
    DEP_LAUNCH(colsum_kernel, dim3(dep_cdiv(F, 64)), dim3(256), 0, S_, (const float*)workspace + F, nb, F, 2 * F, dbeta);
    DEP_CHECK_LAUNCH();
    if (dx) {
        DEP_LAUNCH(ln_bwd_dx_kernel, dim3(dep_cdiv(rows, 4)), dim3(256), 0, S_, dy, x, gamma, mean_rstd, dx, rows, F);
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}

extern "C" int dep_dropout(const float* x, float* y, long n, float p, uint64_t seed, uint32_t site, void* stream) {
    DEP_CHECK_ARG(x && y && n > 0 && p >= 0.f && p < 1.f);
    if (p == 0.f) {
        if (x != y) DEP_LAUNCH(axpby_kernel, dim3(nblk(n)), dim3(256), 0, S_, x, y, n, 1.0f, 0.0f);
    } else {
        DEP_LAUNCH(dropout_kernel, dim3(nblk(n)), dim3(256), 0, S_, x, y, n, p, 1.0f / (1.0f - p), seed, site);
    }
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_dropout_mask(float* mask, long n, float p, uint64_t seed, uint32_t site, void* stream) {
    DEP_CHECK_ARG(mask && n > 0 && p >= 0.f && p < 1.f);
    DEP_LAUNCH(dropout_mask_kernel, dim3(nblk(n)), dim3(256), 0, S_, mask, n, p, 1.0f / (1.0f - p), seed, site);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_relu_dropout_fwd(const float* z, float* a, long n, float p, uint64_t seed, uint32_t site, void* stream) {
    DEP_CHECK_ARG(z && a && n > 0 && p >= 0.f && p < 1.f);
    DEP_LAUNCH(relu_dropout_fwd_kernel, dim3(nblk(n)), dim3(256), 0, S_, z, a, n, p, 1.0f / (1.0f - p), seed, site);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_relu_dropout_bwd(const float* da, const float* z, float* dz, long n, float p, uint64_t seed,
                                    uint32_t site, void* stream) {
    DEP_CHECK_ARG(da && z && dz && n > 0 && p >= 0.f && p < 1.f);
    DEP_LAUNCH(relu_dropout_bwd_kernel, dim3(nblk(n)), dim3(256), 0, S_, da, z, dz, n, p, 1.0f / (1.0f - p), seed, site);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_colsum(const float* x, int M, int N, int ld, float* out, void* stream) {
    DEP_CHECK_ARG(x && out && M > 0 && N > 0 && ld >= N);
    DEP_LAUNCH(colsum_kernel, dim3(dep_cdiv(N, 64)), dim3(256), 0, S_, x, M, N, ld, out);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_head_loss(int kind, const float* z, const void* target, float* out, float* loss_rows, float* dz,
                             int B, int C, float norm, void* stream) {
    DEP_CHECK_ARG(z && B > 0 && C > 0 && C <= MAXC && (kind & ~DEP_LOSS_LABELS_I64) >= 0 && (kind & ~DEP_LOSS_LABELS_I64) <= 4 && norm > 0.f);
    DEP_CHECK_ARG(!(kind & DEP_LOSS_LABELS_I64) || (kind & ~DEP_LOSS_LABELS_I64) == DEP_LOSS_CE_ON_SOFTMAX || (kind & ~DEP_LOSS_LABELS_I64) == DEP_LOSS_CE_LOGITS);
    DEP_CHECK_ARG(target || (!dz && !loss_rows));
    DEP_LAUNCH(head_loss_kernel, dim3(nblk(B, 128)), dim3(128), 0, S_, kind, z, target, out, loss_rows, dz, B, C, 1.0f / norm);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_reduce_loss(const float* loss_rows, int B, float norm, float* loss_out, int accumulate, void* stream) {
    DEP_CHECK_ARG(loss_rows && loss_out && B > 0 && norm > 0.f);
    DEP_LAUNCH(reduce_loss_kernel, dim3(1), dim3(256), 0, S_, loss_rows, B, 1.0f / norm, loss_out, accumulate);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int decoupled, int step, void* stream) {
    DEP_CHECK_ARG(p && g && m && v && n > 0 && step >= 1);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    DEP_LAUNCH(adam_kernel, dim3(nblk(n)), dim3(256), 0, S_, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                       decoupled, step_size, inv_sqrt_bc2);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_fill(float* p, long n, float value, void* stream) {
    DEP_CHECK_ARG(p && n > 0);
    DEP_LAUNCH(fill_kernel, dim3(nblk(n)), dim3(256), 0, S_, p, n, value);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
// dst[k][i] = src[k][i] (+ add[k][i] when add[k] != NULL), k < count <= 16, one launch
int dep_multi_copy(int count, const float* const* src, const float* const* add, float* const* dst, const long* n, hipStream_t s) {
    DEP_CHECK_ARG(count > 0 && count <= 16 && src && dst && n);
    CopyJobs j{};
    long mx = 0;
    for (int k = 0; k < count; ++k) {
        DEP_CHECK_ARG(src[k] && dst[k] && n[k] > 0);
        j.src[k] = src[k]; j.add[k] = add ? add[k] : nullptr; j.dst[k] = dst[k]; j.n[k] = n[k];
        if (n[k] > mx) mx = n[k];
    }
    j.count = count;
    int gx = dep_cdiv(mx, 256 * 4); if (gx > 512) gx = 512; if (gx < 1) gx = 1;
    DEP_LAUNCH(multi_copy_kernel, dim3(gx, count), dim3(256), 0, s, j);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" int dep_axpby(const float* x, float* y, long n, float a, float b, void* stream) {
    DEP_CHECK_ARG(x && y && n > 0);
    DEP_LAUNCH(axpby_kernel, dim3(nblk(n)), dim3(256), 0, S_, x, y, n, a, b);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_sigmoid_gate(const float* g, const float* x, float* y, long n, void* stream) {
    DEP_CHECK_ARG(g && x && y && n > 0);
    DEP_LAUNCH(sigmoid_gate_kernel, dim3(nblk(n)), dim3(256), 0, S_, g, x, y, n);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" int dep_gather_rows(const float* src, const long long* idx, float* dst, long nrows, long row_floats, void* stream) {
    DEP_CHECK_ARG(src && idx && dst && nrows > 0 && row_floats > 0 && nrows <= 65535);
    const int vec = (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 && (row_floats & 3) == 0) ? 1 : 0;
    int gx = dep_cdiv(vec ? row_floats >> 2 : row_floats, 256); if (gx > 64) gx = 64; if (gx < 1) gx = 1;
    DEP_LAUNCH(gather_rows_kernel, dim3(gx, (unsigned)nrows), dim3(256), 0, S_, src, idx, dst, row_floats, vec);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_copy2d(const float* src, long lds, float* dst, long ldd, long rows, long cols, void* stream) {
    DEP_CHECK_ARG(src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= cols);
    int gx = nblk(rows * cols); if (gx > 2048) gx = 2048;
    DEP_LAUNCH(copy2d_kernel, dim3(gx), dim3(256), 0, S_, src, lds, dst, ldd, rows, cols);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_loss_accumulate(const float* loss, const unsigned* status, const unsigned* soft, double* acc, void* stream) {
    DEP_CHECK_ARG(acc && (loss || status || soft));
    DEP_LAUNCH(loss_accumulate_kernel, dim3(1), dim3(64), 0, S_, loss, status, soft, acc);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
extern "C" int dep_argmax_count(const float* p, const void* labels, int labels_i64, int B, int C, long long* count,
                                long long* pred, void* stream) {
    DEP_CHECK_ARG(p && B > 0 && C > 0 && (count || pred) && (!count || labels));
    DEP_LAUNCH(argmax_count_kernel, dim3(1), dim3(256), 0, S_, p, labels, labels_i64, B, C, count, pred);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" int dep_attn_fwd(const float* out, const float* h_n, int K, const float* Wa, const float* ba, float* ctx,
                            float* alpha, float* pre, float* hsum, int B, int T, int H, void* stream) {
    DEP_CHECK_ARG(out && h_n && Wa && ba && ctx && alpha && pre && hsum && B > 0 && T > 0 && H > 0 && K > 0);
    const long BH = (long)B * H;
    DEP_LAUNCH(attn_hsum_kernel, dim3(nblk(BH)), dim3(256), 0, S_, h_n, K, BH, hsum);
    DEP_CHECK_LAUNCH();
    int rc = dep_gemm_internal(0, 1, B, H, H, hsum, H, Wa, H, pre, H, ba, 0.f, 0, 0, nullptr, 0, S_);
    if (rc) return rc;
    if (!dep_attn2_fwd(out, pre, ctx, alpha, B, T, H, S_)) {
        const size_t lds = (size_t)(H + T + 16) * sizeof(float);
        DEP_LAUNCH(attn_fwd_kernel, dim3(B), dim3(256), lds, S_, out, pre, ctx, alpha, T, H);
    }
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" size_t dep_attn_bwd_workspace_bytes(int B, int T, int H) {
    (void)T;
    return dep_align((size_t)2 * B * H * sizeof(float)) + dep_gemm_workspace_bytes(1, 0, H, H, B);
}

extern "C" int dep_attn_bwd(const float* dctx, const float* out, const float* Wa, const float* alpha, const float* pre,
                            const float* hsum, int K, float* dout, float* dh_n, float* dWa, float* dba, int B, int T,
                            int H, void* workspace, size_t workspace_bytes, void* stream) {
    DEP_CHECK_ARG(dctx && out && Wa && alpha && pre && hsum && dout && dh_n && dWa && dba && B > 0 && T > 0 && H > 0);
    if (!workspace || workspace_bytes < dep_attn_bwd_workspace_bytes(B, T, H)) {
        dep_set_error("dep_attn_bwd: workspace too small"); return DEP_ERR_WORKSPACE;
    }
    float* dpre = (float*)workspace;
    float* dhs = dpre + (size_t)B * H;
    char* gws = (char*)workspace + dep_align((size_t)2 * B * H * sizeof(float));
    const size_t gws_bytes = workspace_bytes - dep_align((size_t)2 * B * H * sizeof(float));
    if (!dep_attn2_bwd(dctx, out, alpha, pre, dout, dpre, B, T, H, S_)) {
        const size_t lds = (size_t)(2 * H + 2 * T + 16) * sizeof(float);
        DEP_LAUNCH(attn_bwd_kernel, dim3(B), dim3(256), lds, S_, dctx, out, alpha, pre, dout, dpre, T, H);
    }
    DEP_CHECK_LAUNCH();
    // dWa (H,H) = dpre^T (H,B) * hsum (B,H) ; dba = colsum(dpre) ; dhsum = dpre * Wa
    int rc = dep_gemm_internal(1, 0, H, H, B, dpre, H, hsum, H, dWa, H, nullptr, 0.f, 0, 0, gws, gws_bytes, S_);
    if (rc) return rc;
    rc = dep_colsum(dpre, B, H, H, dba, stream);
    if (rc) return rc;
    rc = dep_gemm_internal(0, 0, B, H, H, dpre, H, Wa, H, dhs, H, nullptr, 0.f, 0, 0, nullptr, 0, S_);
    if (rc) return rc;
    const long BH = (long)B * H;
    DEP_LAUNCH(attn_bcast_kernel, dim3(nblk(BH)), dim3(256), 0, S_, dhs, K, BH, dh_n);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
