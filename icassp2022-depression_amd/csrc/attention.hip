// attention_net_with_w (Classification/text_bilstm_whole.py:74-99), forward and backward, for H in {64, 128, 256} (round 3).
//
// One workgroup per utterance, as before -- but the first kernels (elementwise.hip: attn_fwd_kernel / attn_bwd_kernel, still
// used for other widths) were latency chains: one wave per time step with a single 1 KB row in flight, then one THREAD per
// feature walking all T rows again, a global load and its round trip per iteration: 134 + 190 us at cfg3 for 157 MB that the
// chip streams in ~40 us.  Here a row of `out` (fwd | bwd halves, 2H floats) is read as 16-byte pieces by H/4 lanes, every
// thread keeps AU rows (2 AU loads) in flight, and the sum h_t = fwd + bwd halves the passes need again is kept in LDS
// (T H floats: 150 KB at T = 300, H = 128 -- the CACHE instantiation; longer sequences re-read `out`).  Partial sums over the
// 16 row slots of a workgroup are added in slot order: deterministic.
#include "dep_common.h"

namespace {

constexpr int AT = 512;          // threads per workgroup (8 waves)
constexpr int AU = 4;            // rows in flight per thread
constexpr int LDS_MAX = 160 * 1024;

__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// red: AT / 64 floats.  Every thread gets the result; a barrier in front protects red against the previous use.
__device__ __forceinline__ float bmax(float v, float* red) {
    v = wmax(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < AT / 64; ++i) s = fmaxf(s, red[i]);
    return s;
}
__device__ __forceinline__ float bsum(float v, float* red) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < AT / 64; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ f32x4 tanh4(f32x4 v) { return f32x4{tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3])}; }
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]))); }

// HQ = H / 4 lanes per row; a wave covers 64 / HQ rows, the workgroup R = 8 * 64 / HQ rows per pass
template <int HQ, bool CACHE>
__global__ __launch_bounds__(AT) void attn_fwd2_kernel(const float* __restrict__ out, const float* __restrict__ pre,
                                                       float* __restrict__ ctx, float* __restrict__ alpha, int T) {
    constexpr int H = HQ * 4, RPW = 64 / HQ, R = (AT / 64) * RPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sc = smem;                                  // [T], padded to 4
    float* red = sc + ((T + 3) & ~3);                  // [32]
    float* vs = red + 32;                              // CACHE: h_t for the whole utterance; afterwards the R partial rows
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jc = lane % HQ, g = w * RPW + lane / HQ;
    const float* ob = out + (size_t)b * T * 2 * H + 4 * jc;
    f32x4 q4 = *reinterpret_cast<const f32x4*>(pre + (size_t)b * H + 4 * jc);
#pragma unroll
    for (int e = 0; e < 4; ++e) q4[e] = fmaxf(q4[e], 0.f);
    for (int t0 = 0; t0 < T; t0 += R * AU) {
        f32x4 a[AU], c[AU];
#pragma unroll
        for (int u = 0; u < AU; ++u) {                 // clamped, unconditional: 2 AU loads in flight
            const int t = t0 + u * R + g, tc = t < T ? t : T - 1;
            a[u] = *reinterpret_cast<const f32x4*>(ob + (size_t)tc * 2 * H);
            c[u] = *reinterpret_cast<const f32x4*>(ob + (size_t)tc * 2 * H + H);
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int t = t0 + u * R + g;
            const f32x4 v = a[u] + c[u];
            if (CACHE && t < T) *reinterpret_cast<f32x4*>(&vs[(size_t)t * H + 4 * jc]) = v;
            float s = dot4(q4, tanh4(v));
#pragma unroll
            for (int m = HQ / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
            if (jc == 0 && t < T) sc[t] = s;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = tid; t < T; t += AT) mx = fmaxf(mx, sc[t]);
    mx = bmax(mx, red);
    float se = 0.f;
    for (int t = tid; t < T; t += AT) { const float e = expf(sc[t] - mx); sc[t] = e; se += e; }
    se = bsum(se, red);
    for (int t = tid; t < T; t += AT) { const float al = sc[t] / se; sc[t] = al; alpha[(size_t)b * T + t] = al; }
    __syncthreads();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < T; t0 += R * AU) {
        f32x4 a[AU], c[AU];
        if (!CACHE) {
#pragma unroll
            for (int u = 0; u < AU; ++u) {
                const int t = t0 + u * R + g, tc = t < T ? t : T - 1;
                a[u] = *reinterpret_cast<const f32x4*>(ob + (size_t)tc * 2 * H);
                c[u] = *reinterpret_cast<const f32x4*>(ob + (size_t)tc * 2 * H + H);
            }
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int t = t0 + u * R + g;
            if (t >= T) continue;
            const f32x4 v = CACHE ? *reinterpret_cast<const f32x4*>(&vs[(size_t)t * H + 4 * jc]) : a[u] + c[u];
            const float al = sc[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(al, v[e], acc[e]);
        }
    }
    __syncthreads();                                   // every read of the cached rows is done: reuse their head
    *reinterpret_cast<f32x4*>(&vs[g * H + 4 * jc]) = acc;
    __syncthreads();
    for (int j = tid; j < H; j += AT) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) s += vs[k * H + j];
        ctx[(size_t)b * H + j] = s;
    }
}

template <int HQ, bool CACHE>
__global__ __launch_bounds__(AT) void attn_bwd2_kernel(const float* __restrict__ dctx, const float* __restrict__ out,
                                                       const float* __restrict__ alpha, const float* __restrict__ pre,
                                                       float* __restrict__ dout, float* __restrict__ dpre, int T) {
    constexpr int H = HQ * 4, RPW = 64 / HQ, R = (AT / 64) * RPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Tp = (T + 3) & ~3;
    float* al = smem;                                  // [Tp]
    float* ds = al + Tp;                               // [Tp]
    float* red = ds + Tp;                              // [32]
    float* vs = red + 32;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jc = lane % HQ, g = w * RPW + lane / HQ;
    const float* ob = out + (size_t)b * T * 2 * H + 4 * jc;
    float* dob = dout + (size_t)b * T * 2 * H + 4 * jc;
    f32x4 q4 = *reinterpret_cast<const f32x4*>(pre + (size_t)b * H + 4 * jc);
#pragma unroll
    for (int e = 0; e < 4; ++e) q4[e] = fmaxf(q4[e], 0.f);
    const f32x4 dc4 = *reinterpret_cast<const f32x4*>(dctx + (size_t)b * H + 4 * jc);
    for (int t = tid; t < T; t += AT) al[t] = alpha[(size_t)b * T + t];
    for (int t0 = 0; t0 < T; t0 += R * AU) {
        f32x4 a[AU], c[AU];
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int t = t0 + u * R + g, tc = t < T ? t : T - 1;
            a[u] = *reinterpret_cast<const f32x4*>(ob + (size_t)tc * 2 * H);
            c[u] = *reinterpret_cast<const f32x4*>(ob + (size_t)tc * 2 * H + H);
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int t = t0 + u * R + g;
            const f32x4 v = a[u] + c[u];
            if (CACHE && t < T) *reinterpret_cast<f32x4*>(&vs[(size_t)t * H + 4 * jc]) = v;
            float s = dot4(dc4, v);
#pragma unroll
            for (int m = HQ / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
            if (jc == 0 && t < T) ds[t] = s;           // d alpha
        }
    }
    __syncthreads();
    float dot = 0.f;
    for (int t = tid; t < T; t += AT) dot = fmaf(al[t], ds[t], dot);
    dot = bsum(dot, red);
    for (int t = tid; t < T; t += AT) ds[t] = al[t] * (ds[t] - dot);           // d scores
    __syncthreads();
    f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < T; t0 += R * AU) {
        f32x4 a[AU], c[AU];
        if (!CACHE) {
#pragma unroll
            for (int u = 0; u < AU; ++u) {
                const int t = t0 + u * R + g, tc = t < T ? t : T - 1;
                a[u] = *reinterpret_cast<const f32x4*>(ob + (size_t)tc * 2 * H);
                c[u] = *reinterpret_cast<const f32x4*>(ob + (size_t)tc * 2 * H + H);
            }
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int t = t0 + u * R + g;
            if (t >= T) continue;
            const f32x4 v = CACHE ? *reinterpret_cast<const f32x4*>(&vs[(size_t)t * H + 4 * jc]) : a[u] + c[u];
            const f32x4 m = tanh4(v);
            const float at = al[t], dt = ds[t];
            f32x4 dh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dq[e] = fmaf(dt, m[e], dq[e]);
                dh[e] = at * dc4[e] + dt * q4[e] * (1.0f - m[e] * m[e]);
            }
            *reinterpret_cast<f32x4*>(dob + (size_t)t * 2 * H) = dh;
            *reinterpret_cast<f32x4*>(dob + (size_t)t * 2 * H + H) = dh;
        }
    }
    __syncthreads();
    *reinterpret_cast<f32x4*>(&vs[g * H + 4 * jc]) = dq;
    __syncthreads();
    for (int j = tid; j < H; j += AT) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) s += vs[k * H + j];
        dpre[(size_t)b * H + j] = pre[(size_t)b * H + j] > 0.f ? s : 0.f;
    }
}

bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <typename K>
void allow_lds(K kern, size_t bytes) {
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

bool v1_forced() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DEP_ATTN_V1"); v = (e && atoi(e) != 0) ? 1 : 0; }
    return v == 1;
}

}  // namespace

// Both return 1 when they launched, 0 when the shape is left to the first-generation kernels (elementwise.hip).
int dep_attn2_fwd(const float* out, const float* pre, float* ctx, float* alpha, int B, int T, int H, hipStream_t s) {
    if (v1_forced() || !(H == 64 || H == 128 || H == 256) || !al16(out) || !al16(pre)) return 0;
    const int R = (AT / 64) * (64 / (H / 4));
    const size_t fixed = (size_t)(((T + 3) & ~3) + 32) * 4;
    const size_t full = fixed + (size_t)(T > R ? T : R) * H * 4, small = fixed + (size_t)R * H * 4;
    const bool cache = full <= (size_t)LDS_MAX;
    const size_t lds = cache ? full : small;
    if (lds > (size_t)LDS_MAX) return 0;
#define GO(HQ)                                                                                                   \
    do {                                                                                                         \
        if (cache) { allow_lds(attn_fwd2_kernel<HQ, true>, lds);                                                 \
                     DEP_LAUNCH((attn_fwd2_kernel<HQ, true>), dim3(B), dim3(AT), lds, s, out, pre, ctx, alpha, T); } \
        else { allow_lds(attn_fwd2_kernel<HQ, false>, lds);                                                      \
               DEP_LAUNCH((attn_fwd2_kernel<HQ, false>), dim3(B), dim3(AT), lds, s, out, pre, ctx, alpha, T); }      \
    } while (0)
    if (H == 64) GO(16); else if (H == 128) GO(32); else GO(64);
#undef GO
    return 1;
}

int dep_attn2_bwd(const float* dctx, const float* out, const float* alpha, const float* pre, float* dout, float* dpre, int B,
                  int T, int H, hipStream_t s) {
    if (v1_forced() || !(H == 64 || H == 128 || H == 256) || !al16(out) || !al16(pre) || !al16(dctx) || !al16(dout)) return 0;
    const int R = (AT / 64) * (64 / (H / 4));
    const size_t fixed = (size_t)(2 * ((T + 3) & ~3) + 32) * 4;
    const size_t full = fixed + (size_t)(T > R ? T : R) * H * 4, small = fixed + (size_t)R * H * 4;
    const bool cache = full <= (size_t)LDS_MAX;
    const size_t lds = cache ? full : small;
    if (lds > (size_t)LDS_MAX) return 0;
#define GO(HQ)                                                                                                   \
    do {                                                                                                         \
        if (cache) { allow_lds(attn_bwd2_kernel<HQ, true>, lds);                                                 \
                     DEP_LAUNCH((attn_bwd2_kernel<HQ, true>), dim3(B), dim3(AT), lds, s, dctx, out, alpha, pre, dout, dpre, T); } \
        else { allow_lds(attn_bwd2_kernel<HQ, false>, lds);                                                      \
               DEP_LAUNCH((attn_bwd2_kernel<HQ, false>), dim3(B), dim3(AT), lds, s, dctx, out, alpha, pre, dout, dpre, T); }      \
    } while (0)
    if (H == 64) GO(16); else if (H == 128) GO(32); else GO(64);
#undef GO
    return 1;
}
