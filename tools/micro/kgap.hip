// What does a kernel boundary cost behind a kernel that leaves dirty lines in the L2s?  
// A writer (or reader) kernel over `mb` MiB followed by a one-workgroup kernel on the same stream, 200 pairs, against the writer alone and
// the tiny kernel alone:  gap = (pair - writer - tiny) per iteration.  `nt` = the writer uses non-temporal stores.
//     hipcc --offload-arch=gfx950 -O3 tools/micro/kgap.hip -o tools/micro/kgap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void writer(f32x4* p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = f32x4{v, v, v, v};
}
__global__ void writer_nt(f32x4* p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(f32x4{v, v, v, v}, p + i);
}
__global__ void reader(const f32x4* p, size_t n4, float* out) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s[0] + s[1] + s[2] + s[3] == 1.2345e30f) out[0] = s[0];
}
__global__ void tiny(float* out) { if (threadIdx.x == 0) out[1] += 1.f; }

static float run(int kind, f32x4* buf, size_t n4, float* out, int iters, bool with_tiny, bool with_big) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const dim3 g(2048), t(256);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i) {
            if (with_big) {
                if (kind == 0) hipLaunchKernelGGL(writer, g, t, 0, 0, buf, n4, (float)i);
                else if (kind == 1) hipLaunchKernelGGL(writer_nt, g, t, 0, 0, buf, n4, (float)i);
                else hipLaunchKernelGGL(reader, g, t, 0, 0, buf, n4, out);
            }
            if (with_tiny) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, out);
        }
        hipEventRecord(b, 0); hipEventSynchronize(b);
    }
    float ms = 0.f; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / iters;
}

int main() {
    const size_t maxb = (size_t)1024 << 20;
    f32x4* buf; float* out;
    hipMalloc(&buf, maxb); hipMalloc(&out, 256); hipMemset(buf, 0, maxb); hipMemset(out, 0, 256);
    const int iters = 200;
    const float t_tiny = run(0, buf, 0, out, iters, true, false);
    printf("tiny kernel alone: %.2f us per launch (back to back)\n", t_tiny);
    const char* names[3] = {"write", "write-nt", "read"};
    for (int kind = 0; kind < 3; ++kind)
        for (size_t mb : {(size_t)1, (size_t)8, (size_t)32, (size_t)128, (size_t)512}) {
            const size_t n4 = (mb << 20) / 16;
            const float big = run(kind, buf, n4, out, iters, false, true);
            const float pair = run(kind, buf, n4, out, iters, true, true);
            printf("%-9s %4zu MiB: alone %8.2f us (%.2f TB/s)  + tiny %8.2f us  -> boundary cost %6.2f us beyond the tiny kernel's own %.2f\n",
                   names[kind], mb, big, (double)(mb << 20) / big / 1e6, pair, pair - big - t_tiny, t_tiny);
        }
    return 0;
}
