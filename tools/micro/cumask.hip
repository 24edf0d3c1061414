// How do CU-masked streams map workgroups to XCDs / CUs on MI355X?  (run the latency-bound backward sweep
// on four XCDs and the power-bound dW GEMMs on the other four.)  For a few mask patterns: launch 256 and 512 workgroups of 256
// threads that record HW_REG_XCC_ID and HW_REG_HW_ID, print the XCD of blockIdx 0..15, the XCD histogram, the number of
// distinct (xcd, se, cu) slots used and whether blockIdx % 8 -> XCD is still a function.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <map>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the workgroup resident for a while so that the launch spreads over the CUs instead of reusing the first ones
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; }
}

static void run(const char* name, const std::vector<unsigned>& mask, int wgs, unsigned* d_out) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
    hipMemsetAsync(d_out, 0xff, 2 * wgs * sizeof(unsigned), s);
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, s, d_out, 200000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * wgs);
    hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, int> hist; std::set<std::pair<unsigned, unsigned>> slots;
    bool fn = true; std::map<int, unsigned> by8;
    for (int i = 0; i < wgs; ++i) {
        const unsigned x = h[2 * i], hw = h[2 * i + 1];
        hist[x]++;
        slots.insert({x, hw & 0xfff00u});        // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13] (wave / simd bits dropped)
        if (by8.count(i % 8) && by8[i % 8] != x) fn = false;
        by8[i % 8] = x;
    }
    printf("%-26s wgs %4d : xcd of block 0..15:", name, wgs);
    for (int i = 0; i < 16 && i < wgs; ++i) printf(" %u", h[2 * i]);
    printf(" | hist:");
    for (auto& kv : hist) printf(" x%u=%d", kv.first, kv.second);
    printf(" | distinct cu slots %zu | block%%8 -> xcd is %s\n", slots.size(), fn ? "a function" : "NOT a function");
    hipStreamDestroy(s);
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("%s, %d CUs\n", pr.name, pr.multiProcessorCount);
    unsigned* d_out; hipMalloc(&d_out, 2 * 4096 * sizeof(unsigned));
    const int words = 8;                                   // 256 bits
    std::vector<unsigned> all(words, 0xffffffffu), lo(words, 0), hi(words, 0), even(words, 0x55555555u), odd(words, 0xaaaaaaaau);
    for (int i = 0; i < words / 2; ++i) { lo[i] = 0xffffffffu; hi[words / 2 + i] = 0xffffffffu; }
    std::vector<unsigned> nib_lo(words, 0x0f0f0f0fu), nib_hi(words, 0xf0f0f0f0u);
    std::vector<unsigned> b8_lo(words, 0x00ff00ffu), w_alt(words, 0);
    for (int i = 0; i < words; i += 2) w_alt[i] = 0xffffffffu;
    for (int wgs : {256, 512}) {
        run("all", all, wgs, d_out);
        run("bits 0..127", lo, wgs, d_out);
        run("bits 128..255", hi, wgs, d_out);
        run("even bits", even, wgs, d_out);
        run("odd bits", odd, wgs, d_out);
        run("low nibble of each byte", nib_lo, wgs, d_out);
        run("high nibble of each byte", nib_hi, wgs, d_out);
        run("low byte of each 16", b8_lo, wgs, d_out);
        run("even 32-bit words", w_alt, wgs, d_out);
    }
    return 0;
}
