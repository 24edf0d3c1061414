// Round 4: WHY does the backward sweep's exchange payload -- rewritten in place every other step, live set 1 MB per
// XCD -- reach HBM (PMC: ~0.8 GB of write-backs per launch for 1.26 GB stored), while the fused forward's (0.4 MB per XCD) mostly
// does not?  Each workgroup rewrites ITS region `iters` times; variants differ in what else happens.  Run under
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...   (and FETCH_SIZE)       -> bytes written back per launch vs bytes stored
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2wb.hip -o tools/micro/l2wb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

// MODE bit 0: 8-byte stores per lane instead of 16; bit 1: the neighbour workgroup of the same XCD reads the region with sc1 loads
// after every rewrite; bit 2: every workgroup also streams `stream_kb` KB of one-touch HBM reads per iteration through the L2;
// bit 3: ... and one-touch writes of the same size
template <int MODE>
__global__ __launch_bounds__(256) void rewrite(float* regions, int region_floats, int iters, const f32x4* stream_src, f32x4* stream_dst,
                                               int stream_f4, float* sink) {
    float* mine = regions + (size_t)blockIdx.x * region_floats;
    const float* nb = regions + (size_t)((blockIdx.x + 8) % gridDim.x) * region_floats;       // same XCD (block % 8)
    const int tid = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    size_t spos = (size_t)blockIdx.x * stream_f4 * iters;
    for (int it = 0; it < iters; ++it) {
        const float v = (float)it;
        if (MODE & 1) {
            for (int i = tid * 2; i < region_floats; i += 512) { float2 x = {v, v + 1.f}; *reinterpret_cast<float2*>(mine + i) = x; }
        } else {
            for (int i = tid * 4; i < region_floats; i += 1024) { f32x4 x = {v, v + 1.f, v + 2.f, v + 3.f}; *reinterpret_cast<f32x4*>(mine + i) = x; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (MODE & 2) {
            for (int i = tid; i < region_floats; i += 256 * 8)
                acc[0] += __uint_as_float(__hip_atomic_load((gu32*)(nb + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        if (MODE & 4) {
            for (int i = tid; i < stream_f4; i += 256) acc += stream_src[spos + i];
        }
        if (MODE & 8) {
            // bits 4..6: flavour of the one-touch stores: 0 plain, 1 nt, 2 sc0 sc1, 3 sc0 sc1 nt, 4 sc1, 5 sc1 nt, 6 sc0, 7 sc0 nt
            constexpr int FL = (MODE >> 4) & 7;
            for (int i = tid; i < stream_f4; i += 256) {
                f32x4* q = stream_dst + spos + i;
                if (FL == 0) *q = acc;
                else if (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(q), "v"(acc) : "memory");
                else if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(acc) : "memory");
                else if (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(q), "v"(acc) : "memory");
                else if (FL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(acc) : "memory");
                else if (FL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(q), "v"(acc) : "memory");
                else if (FL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(q), "v"(acc) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(q), "v"(acc) : "memory");
            }
        }
        spos += stream_f4;
        for (int k = 0; k < 8; ++k) __builtin_amdgcn_s_sleep(64);       // ~2 us per iteration, like a sweep step
        __syncthreads();
    }
    if (acc[0] + acc[1] == 1.2345e30f) sink[tid] = acc[0];
}

int main(int argc, char** argv) {
    const int region_kb = argc > 1 ? atoi(argv[1]) : 32, iters = 300, stream_kb = 20;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int wgs = pr.multiProcessorCount;
    const int region_floats = region_kb * 256, stream_f4 = stream_kb * 64;
    float* regions; float* sink; f32x4* src; f32x4* dst;
    hipMalloc(&regions, (size_t)wgs * region_floats * 4); hipMalloc(&sink, 4096);
    const size_t sbytes = (size_t)wgs * stream_f4 * iters * 16;
    hipMalloc(&src, sbytes); hipMalloc(&dst, sbytes);
    hipMemset(regions, 0, (size_t)wgs * region_floats * 4); hipMemset(src, 0, sbytes);
    printf("%d workgroups x %d KB region (%.2f MB live per XCD), %d rewrites: %.1f MB stored per launch; stream %d KB / iteration / workgroup = %.1f MB\n",
           wgs, region_kb, wgs / 8 * region_kb / 1024.0, iters, (double)wgs * region_kb * iters / 1024.0, stream_kb, (double)wgs * stream_kb * iters / 1024.0);
#define RUN(M) do { hipLaunchKernelGGL(rewrite<M>, dim3(wgs), dim3(256), 0, 0, regions, region_floats, iters, src, dst, stream_f4, sink); hipDeviceSynchronize(); } while (0)
    RUN(0); RUN(4); RUN(12); RUN(12 + 16); RUN(12 + 32); RUN(12 + 48); RUN(12 + 64); RUN(12 + 80); RUN(12 + 96); RUN(12 + 112);
    printf("kernels: rewrite<0> rewrites only | <4> + one-touch reads | <12> + one-touch reads and writes; <12 + 16 f>: flavour f of the one-touch stores: 1 nt, 2 sc0 sc1, 3 sc0 sc1 nt, 4 sc1, 5 sc1 nt, 6 sc0, 7 sc0 nt\n");
    return 0;
}
