// Round 6 ("settle the diagnosis with numbers"): the shader clock WHILE a library kernel runs.
// A probe wave per workgroup records, for `nwin` consecutive windows of `win_ticks` ticks of the constant 100 MHz counter (s_memrealtime),
// the shader-cycle counter (s_memtime) delta -> MHz per window.  Launched on a second stream BEFORE the kernel under test so that its
// waves are resident (the persistent GEMMs and the exclusive sweeps leave no VGPRs for a late wave); tools/gemm_clock.py drives it.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/clkprobe.hip -o tools/micro/libclkprobe.so
#include <hip/hip_runtime.h>
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }
// out[blk][0] = xcc; out[blk][1 + 3 w + {0,1,2}] = realtime at window start, shader cycles, realtime ticks
__global__ __launch_bounds__(64) void clk_probe(long long* out, int nwin, int win_ticks) {
    long long* o = out + (size_t)blockIdx.x * (1 + 3 * nwin);
    if (threadIdx.x == 0) o[0] = xcc_id();
    for (int w = 0; w < nwin; ++w) {
        const long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
        long long r1 = r0;
        while (r1 - r0 < win_ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
        const long long c1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) { o[1 + 3 * w] = r0; o[2 + 3 * w] = c1 - c0; o[3 + 3 * w] = r1 - r0; }
    }
}
__global__ void clk_now(long long* out) { if (threadIdx.x == 0) out[0] = __builtin_amdgcn_s_memrealtime(); }
extern "C" int clk_probe_launch(void* stream, long long* out, int nblk, int nwin, int win_ticks) {
    hipLaunchKernelGGL(clk_probe, dim3(nblk), dim3(64), 0, (hipStream_t)stream, out, nwin, win_ticks);
    return (int)hipGetLastError();
}
extern "C" int clk_now_launch(void* stream, long long* out) {
    hipLaunchKernelGGL(clk_now, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
