// Does a vector load that hits in L2 wait for an OLDER load of ANOTHER wave of the same CU that misses to HBM?
// One workgroup per CU (big LDS request), two waves.  Wave 1 issues `nmiss` loads to cold lines (a fresh 4 KB-strided region
// every iteration); 200 cycles later wave 0 issues one sc1 load to a hot line and times its round trip.
//   hipcc --offload-arch=gfx950 -O3 inorder.hip -o inorder
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(128) void probe(f32x4* cold, const unsigned* hot, long long* out, int nmiss, int iters, int stores) {
    extern __shared__ float lds[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long sum = 0, mx = 0; f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        const size_t base = ((size_t)(blockIdx.x * iters + it) * 64 + lane) * 1024;     // 16 KB apart per lane: every lane its own line / page
        if (w == 1) {
            if (stores) { const f32x4 v = {1.f, 2.f, 3.f, (float)it}; for (int i = 0; i < nmiss; ++i) cold[base + i * 8] = v; }
            else for (int i = 0; i < nmiss; ++i) acc += cold[base + i * 8];
        }
        if (w == 0) {
            __builtin_amdgcn_s_sleep(3);            // ~200 cycles
            const long long t0 = __builtin_readcyclecounter();
            const unsigned v = __hip_atomic_load(hot + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t1 = __builtin_readcyclecounter();
            if (v == 0xdeadbeef) acc.x += 1.f;
            if (it >= 2) { sum += t1 - t0; if (t1 - t0 > mx) mx = t1 - t0; }
        }
    }
    if (acc.x == 1.2345e30f) out[100] = 1;
    if (w == 0 && lane == 0) { out[blockIdx.x * 2] = sum / (iters - 2); out[blockIdx.x * 2 + 1] = mx; }
}

int main() {
    const int WG = 256, iters = 34;
    f32x4* cold; unsigned* hot; long long* out;
    const size_t coldN = (size_t)WG * iters * 64 * 1024 + 4096;
    hipMalloc(&cold, coldN * 16); hipMalloc(&hot, 4096); hipMalloc(&out, 8192);
    hipMemset(cold, 0, coldN * 16); hipMemset(hot, 0, 4096);
    long long h[512];
    for (int same = 0; same < 2; ++same)      // 0: the other wave LOADS cold lines, 1: it STORES to them
        for (int nmiss : {0, 1, 4, 8}) {
            hipMemset(out, 0, 8192);
            hipLaunchKernelGGL(probe, dim3(WG), dim3(128), 100 * 1024, 0, cold, hot, out, nmiss, iters, same);
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            long long s = 0, m = 0; for (int i = 0; i < WG; ++i) { s += h[2 * i]; if (h[2 * i + 1] > m) m = h[2 * i + 1]; }
            printf("other wave: %d cold %s -> hot-line sc1 load round trip mean %lld, max %lld cycles\n",
                   nmiss, same ? "stores" : "loads ", s / WG, m);
        }
    return 0;
}
