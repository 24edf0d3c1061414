// How fast can all CUs pull operand panels that HIT in L2 (or MALL)?  The big GEMMs move 5.5-6.3 TB/s from L2 to the CUs
// whatever their schedule; this separates "chip-wide L2 -> CU feed limit" from "too few bytes in flight".
// Every workgroup streams a REGION of `bytes` (power of two) over and over with D 16-byte loads in flight per thread;
// workgroups of one XCD (blockIdx & 7) share the region, so for regions <= ~3 MB every load after the first pass is an L2 hit,
// <= 200 MB a MALL hit, larger: HBM.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2bw.hip -o tools/micro/l2bw && ./tools/micro/l2bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ p, size_t nvec, int iters, f32x4* out, size_t xcd_stride) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const f32x4* base = p + (size_t)xcd * xcd_stride;
    const size_t mask = nvec - 1;
    size_t idx = ((size_t)slot * 2654435761u * 64 + threadIdx.x) & mask;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = __builtin_nontemporal_load(&base[(idx + (size_t)d * 256) & mask]);
#pragma unroll
        for (int d = 0; d < D; ++d) acc += v[d];
        idx = (idx + (size_t)D * 256) & mask;
    }
    if (acc[0] == 1.2345e30f) out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int D>
__global__ __launch_bounds__(256) void rd_plain(const f32x4* __restrict__ p, size_t nvec, int iters, f32x4* out, size_t xcd_stride) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const f32x4* base = p + (size_t)xcd * xcd_stride;
    const size_t mask = nvec - 1;
    size_t idx = ((size_t)slot * 2654435761u * 64 + threadIdx.x) & mask;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = base[(idx + (size_t)d * 256) & mask];
#pragma unroll
        for (int d = 0; d < D; ++d) acc += v[d];
        idx = (idx + (size_t)D * 256) & mask;
    }
    if (acc[0] == 1.2345e30f) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int D>
static void run(const char* what, const f32x4* p, size_t bytes, int wgs, f32x4* out, bool shared_per_xcd, bool nt) {
    const size_t nvec = bytes / 16;
    const size_t total = 6ull << 30;                       // ~6 GB of loads per measurement
    int iters = (int)(total / ((size_t)wgs * 256 * 16 * D));
    if (iters < 4) iters = 4;
    const size_t stride = shared_per_xcd ? nvec : 0;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a, 0);
        if (nt) hipLaunchKernelGGL(rd<D>, dim3(wgs), dim3(256), 0, 0, p, nvec, iters, out, stride);
        else    hipLaunchKernelGGL(rd_plain<D>, dim3(wgs), dim3(256), 0, 0, p, nvec, iters, out, stride);
        hipEventRecord(b, 0); hipEventSynchronize(b);
    }
    float ms = 0.f; hipEventElapsedTime(&ms, a, b);
    const double moved = (double)wgs * 256 * 16 * D * iters;
    printf("%-10s region %8.2f MB/xcd  wgs %5d  depth %2d  %s : %8.3f ms  %7.2f TB/s  (%5.1f GB/s per CU)\n", what, bytes / 1048576.0, wgs,
           D, nt ? "nt   " : "plain", ms, moved / ms / 1e9, moved / ms / 1e6 / 256);
}

int main() {
    const size_t cap = 8ull * (256ull << 20);              // 8 per-XCD regions of up to 256 MB
    f32x4* p; f32x4* out;
    if (hipMalloc(&p, cap) != hipSuccess || hipMalloc(&out, 4096 * 256 * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(p, 0, cap);
    const size_t sizes[] = {1ull << 20, 2ull << 20, 16ull << 20, 256ull << 20};
    for (size_t s : sizes) {
        for (int wgs : {256, 512, 1024, 2048}) {
            run<4>("L2/MALL", p, s, wgs, out, true, false);
            run<8>("L2/MALL", p, s, wgs, out, true, false);
        }
        run<16>("L2/MALL", p, s, 512, out, true, false);
        run<8>("L2/MALL", p, s, 1024, out, true, true);
    }
    return 0;
}
