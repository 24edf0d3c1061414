// Micro-benchmark: what does the memory system deliver for the backward sweep's per-step access pattern, with nothing else
// going on?  256 workgroups (member c of tile bt) x 128 threads walk t = T-1 .. 0; per step each thread loads one 16-byte piece
// of 6 arrays and stores one piece of 4 arrays -- rows (utterances) 16 bt .. 16 bt + 15, units 32 c .. 32 c + 31 -- in either
// the batch-major layout the sweeps use ([B][T][H]: 1 KB contiguous per (row, array, step), rows 300 KB apart) or a
// time-major one ([T][B][H]: the whole step contiguous).   hipcc --offload-arch=gfx950 -O3 stream_pattern.hip -o stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int B = 512, T = 300, H = 256;

template <bool TMAJOR, int DEPTH>
__global__ __launch_bounds__(256) void walk(const float* __restrict__ in, float* __restrict__ out, float* sink) {
    const int bt = blockIdx.x % 32, c = blockIdx.x / 32;
    const int row = (bt * 16 + (threadIdx.x >> 3)) % B, col = c * 32 + (threadIdx.x & 7) * 4;
    const size_t arr = (size_t)B * T * H;
    f32x4 acc = {0, 0, 0, 0};
    f32x4 ring[DEPTH][6];
    auto off = [&](int t) { return TMAJOR ? ((size_t)t * B + row) * H + col : ((size_t)row * T + t) * H + col; };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int a = 0; a < 6; ++a) ring[d][a] = *reinterpret_cast<const f32x4*>(in + a * arr + off(T - 1 - d));
    for (int t0 = T - 1; t0 >= 0; t0 -= DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int t = t0 - d;
            if (t < 0) break;
            f32x4 v[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) v[a] = ring[d][a];
            if (t - DEPTH >= 0) {
#pragma unroll
                for (int a = 0; a < 6; ++a) ring[d][a] = *reinterpret_cast<const f32x4*>(in + a * arr + off(t - DEPTH));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) *reinterpret_cast<f32x4*>(out + a * arr + off(t)) = v[a] + v[a + 2];
            acc += v[4] + v[5];
        }
    }
    if (acc.x == 12345.f) sink[0] = acc.y;
}

int main() {
    const size_t arr = (size_t)B * T * H;
    float *in, *out, *sink;
    hipMalloc(&in, 6 * arr * 4); hipMalloc(&out, 4 * arr * 4); hipMalloc(&sink, 64);
    hipMemset(in, 0, 6 * arr * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double gb = 10.0 * arr * 4 / 1e9;
    auto run = [&](const char* name, auto kern) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(128), 0, 0, in, out, sink);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(128), 0, 0, in, out, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-28s %.3f ms  %.2f TB/s  (%.2f GB per launch)\n", name, ms, gb / ms, gb);
    };
    // per-CU streaming rate when only 64 workgroups (one CU each) are active: what a burst can draw
    {
        auto run64 = [&](const char* name, auto kern, int threads) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(64), dim3(threads), 0, 0, in, out, sink);
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(64), dim3(threads), 0, 0, in, out, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            const double g = gb / 4 * (threads / 128.0);
            printf("%-28s %.3f ms  %.1f GB/s per CU (64 CUs active, %d threads each)\n", name, ms, g / 64 / ms * 1e3, threads);
        };
        run64("64 WGs, depth 1", walk<false, 1>, 128);
        run64("64 WGs, depth 4", walk<false, 4>, 128);
        run64("64 WGs, depth 4, 256 thr", walk<false, 4>, 256);
        run64("64 WGs, depth 8, 256 thr", walk<false, 8>, 256);
    }
    run("batch-major, depth 1", walk<false, 1>);
    run("batch-major, depth 2", walk<false, 2>);
    run("batch-major, depth 4", walk<false, 4>);
    run("time-major,  depth 1", walk<true, 1>);
    run("time-major,  depth 2", walk<true, 2>);
    run("time-major,  depth 4", walk<true, 4>);
    return 0;
}
