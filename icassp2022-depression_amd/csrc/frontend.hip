// Audio feature front-end: the elementwise / small-reduction pieces of
//   log-mel spectrogram (librosa.feature.melspectrogram + log, Classification/audio_features_whole.py:59-60) and
//   NetVLAD pooling (loupe_keras.NetVLAD, audio_features_whole.py:64-66).
// The contractions (windowed frames x DFT basis, power x mel filterbank, frames x cluster weights, assignment^T x frames,
// VLAD x hidden weights) run on dep_gemm_f32; these kernels are the glue between them.  All HBM-bound, one pass each.
#include "dep_common.h"

namespace {

// out[i][k] = ypad[i*hop + k] * hann[k], ypad = reflect-padded y (n_fft/2 on both sides, numpy 'reflect': no edge repeat),
// hann = periodic Hann window (scipy get_window('hann', n_fft, fftbins=True))
__global__ void frame_window_kernel(const float* __restrict__ y, long n, int n_fft, int hop, int n_frames, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n_frames * n_fft) return;
    const int k = (int)(idx % n_fft);
    const long i = idx / n_fft;
    long p = i * hop + k - n_fft / 2;
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
    const float w = 0.5f - 0.5f * cospif(2.0f * (float)k / (float)n_fft);
    out[idx] = y[p] * w;
}

// reim row r: [re(0..bins) | im(0..bins)] (ld floats) -> power[r][b] = re^2 + im^2
__global__ void power_kernel(const float* __restrict__ reim, int rows, int bins, int ld, float* __restrict__ power) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * bins) return;
    const int b = (int)(idx % bins);
    const long r = idx / bins;
    const float re = reim[r * ld + b], im = reim[r * ld + bins + b];
    power[idx] = re * re + im * im;
}

__global__ void log_floor_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float floor_) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) y[idx] = logf(fmaxf(floor_, x[idx]));
}

// softmax over the last axis, C <= 64: one thread per row (C = 16 clusters in the path)
__global__ void row_softmax_kernel(const float* __restrict__ z, float* __restrict__ p, int rows, int C) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* zr = z + (size_t)r * C;
    float m = zr[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, zr[c]);
    float e[64], s = 0.f;
    for (int c = 0; c < C; ++c) { e[c] = expf(zr[c] - m); s += e[c]; }
    const float inv = 1.0f / s;
    for (int c = 0; c < C; ++c) p[(size_t)r * C + c] = e[c] * inv;
}

// NetVLAD tail: v[f][k] = vkf[k*F + f] - a_sum[k] * w2[f*K + k]; L2-normalise over f per cluster, then over everything
// (tf.nn.l2_normalize: x * rsqrt(max(sum x^2, 1e-12))); out in (F,K) row-major order.  One workgroup (F*K = 1280 values).
__global__ void vlad_normalize_kernel(const float* __restrict__ vkf, const float* __restrict__ a_sum, const float* __restrict__ w2,
                                      float* __restrict__ out, int F, int K) {
    extern __shared__ float sm[];                    // [F*K] values, [K] cluster norms, [1] total
    float* v = sm; float* cn = sm + F * K; float* tot = cn + K;
    const int tid = threadIdx.x, n = F * K;
    for (int i = tid; i < n; i += blockDim.x) { const int f = i / K, k = i % K; v[i] = vkf[(size_t)k * F + f] - a_sum[k] * w2[i]; }
    __syncthreads();
    if (tid < K) { float s = 0.f; for (int f = 0; f < F; ++f) { const float x = v[f * K + tid]; s += x * x; } cn[tid] = rsqrtf(fmaxf(s, 1e-12f)); }
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) v[i] *= cn[i % K];
    __syncthreads();
    if (tid == 0) { float s = 0.f; for (int i = 0; i < n; ++i) s += v[i] * v[i]; tot[0] = rsqrtf(fmaxf(s, 1e-12f)); }
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) out[i] = v[i] * tot[0];
}

}  // namespace

extern "C" int dep_frame_window(const float* y, long n, int n_fft, int hop, int n_frames, float* out, void* stream) {
    DEP_CHECK_ARG(y && out && n > n_fft / 2 && n_fft > 0 && hop > 0 && n_frames > 0);
    DEP_CHECK_ARG((long)(n_frames - 1) * hop + n_fft <= n + n_fft);      // the last frame ends inside the padded signal
    const long tot = (long)n_frames * n_fft;
    DEP_LAUNCH(frame_window_kernel, dim3(dep_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, y, n, n_fft, hop, n_frames, out);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" int dep_power_spectrum(const float* reim, int rows, int bins, int ld, float* power, void* stream) {
    DEP_CHECK_ARG(reim && power && rows > 0 && bins > 0 && ld >= 2 * bins);
    const long tot = (long)rows * bins;
    DEP_LAUNCH(power_kernel, dim3(dep_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, reim, rows, bins, ld, power);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" int dep_log_floor(const float* x, float* y, long n, float floor_value, void* stream) {
    DEP_CHECK_ARG(x && y && n > 0 && floor_value > 0.f);
    DEP_LAUNCH(log_floor_kernel, dim3(dep_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, floor_value);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" int dep_row_softmax(const float* z, float* p, int rows, int C, void* stream) {
    DEP_CHECK_ARG(z && p && rows > 0 && C > 0 && C <= 64);
    DEP_LAUNCH(row_softmax_kernel, dim3(dep_cdiv(rows, 128)), dim3(128), 0, (hipStream_t)stream, z, p, rows, C);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

extern "C" int dep_vlad_normalize(const float* vkf, const float* a_sum, const float* w2, float* out, int F, int K, void* stream) {
    DEP_CHECK_ARG(vkf && a_sum && w2 && out && F > 0 && K > 0 && (size_t)(F * K + K + 1) * sizeof(float) <= 64 * 1024);
    DEP_LAUNCH(vlad_normalize_kernel, dim3(1), dim3(256), (size_t)(F * K + K + 1) * sizeof(float), (hipStream_t)stream,
                       vkf, a_sum, w2, out, F, K);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
