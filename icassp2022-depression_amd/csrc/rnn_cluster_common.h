// Shared device helpers of the cluster-parallel sweeps (rnn_cluster_bwd.hip, rnn_cluster16.hip, rnn_cluster_lstm.hip):
// the in-launch workgroup hand-off protocol (cdna_hip_programming.md Guideline 16, recipe R1) and a few small utilities.
#pragma once
#include "dep_common.h"

namespace depc {

constexpr int BT = 16;                      // utterances per tile = MFMA N
constexpr int LPAD = 4;                     // LDS row padding (floats)
constexpr int CT = 256;                     // compute threads per workgroup
constexpr unsigned SPIN_LIMIT = 1u << 20;   // bounded spins: ~1 s, then the status word is raised and the kernel leaves
constexpr unsigned HELLO_LIMIT_SOFT = 1u << 13;    // hello of a kernel that has a fallback: a few ms (a normal hello takes ~20 us)
// exchange buffer header: word [0] status, word [1] "soft" flag (both cleared by dep_rnn_forward) | flags (<= 1024 words) |
// hello (<= 512 words) | trace (zeroed before every launch)
//   soft flag: a forward kernel that needs every CU to itself (rnn_fused2.hip) could not assemble its clusters -- a foreign
//   workgroup sat in the dispatcher.  It leaves WITHOUT raising the status; the co-schedule-tolerant per-layer kernels
//   enqueued behind it run only when this word is set (dep_rnn_forward, api.hip) and redo the forward.
constexpr size_t FLAG_OFF = 256, HELLO_OFF = 4352, TRACE_OFF = 6400;     // inside a header slot; flags: 1024 words (per-wave flags of the GRU backward), hello: 512 words
// The buffer starts with HDR_SLOTS such headers of HDR_SLOT bytes; the payload follows.  A launch uses slot `hdr_slot` (status
// and soft flag are always slot 0's words [0], [1]).  Round 3: every sweep of a forward (or backward) call takes its own slot, so
// that ONE memset per call clears them all -- six 4.5 us fill launches per training step were four too many.
constexpr size_t HDR_SLOT = 8192;
constexpr int HDR_SLOTS = DEP_HDR_SLOTS;
constexpr size_t PAYLOAD_OFF = HDR_SLOT * HDR_SLOTS;
inline char* hdr_base(void* xbuf, int slot) { return (char*)xbuf + (size_t)(slot > 0 && slot < HDR_SLOTS ? slot : 0) * HDR_SLOT; }
// flags / hello / trace words of a slot are zero at launch: either the call's single memset did it (`clean`, first chunk only)
// or the launcher does it here
inline int hdr_prepare(void* xbuf, int slot, bool clean, hipStream_t s) {
    if (clean) return DEP_OK;
    if (hipMemsetAsync(hdr_base(xbuf, slot) + FLAG_OFF, 0, HDR_SLOT - FLAG_OFF, s) != hipSuccess) { dep_set_error("hipMemsetAsync failed"); return DEP_ERR_HIP; }
    return DEP_OK;
}

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ float2 f2(float a, float b) { return make_float2(a, b); }

// agent-scope (sc1: L1 bypass, write-through) and workgroup-scope (plain) accesses of the shared words
__device__ __forceinline__ unsigned ld_agent(unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_local(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ float ldf_agent(const float* p) {
    return __uint_as_float(__hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float2 ld2_agent(const float* p) {
    const u64 x = __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)x), __uint_as_float((unsigned)(x >> 32)));
}

// Workgroup barrier that orders LDS only.  __syncthreads() also drains vmcnt, i.e. it would put every global store
// issued earlier in the step (y, dropped y, saved gates, dgi) on the step's critical path; here those stay in flight and
// are only drained by the explicit s_waitcnt vmcnt(0) that precedes the next flag publication.
__device__ __forceinline__ void bar_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// v_exp_f32 / v_rcp_f32 gate nonlinearities: absolute error ~2e-7, far inside the 1e-4 parity budget and several times
// shorter than the ocml expf / tanhf sequences that sat on the per-step critical path
// (__builtin_amdgcn_rcpf is the bare v_rcp_f32, 1 ulp; __frcp_rn expands to the ten-instruction IEEE division sequence)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

// 3-term bf16 split of one fp32 value for the split-precision sweeps: bf16(x) << 16 | bf16(x - bf16(x))
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned split_word(float x) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef __bf16 b2v __attribute__((ext_vector_type(2)));
    const f2v v = {x, 0.f};
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2v)) & 0xffffu;
    const f2v d = {x - __uint_as_float(hi << 16), 0.f};
    const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(d, b2v)) & 0xffffu;
    return (hi << 16) | lo;
}

// 16-bit fixed-point storage of the saved GRU gates (round 4, dep_sweep_args.sv16): r, z in (0, 1) as unorm16 (round(x * 65535),
// |error| <= 7.6e-6), n in (-1, 1) as snorm16 (round(x * 32767), |error| <= 1.5e-5) -- V_CVT_PKNORM_{U,I}16_F32.  Same order as
// the split products' own error; halves 6H of the 16H saved-gate bytes a step writes in the forward and reads in the backward.
__device__ __forceinline__ unsigned pack_unorm2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(a, b)); }
__device__ __forceinline__ unsigned pack_snorm2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_i16(a, b)); }
__device__ __forceinline__ float2 unpack_unorm2(unsigned w) {
    return make_float2((float)(w & 0xffffu) * (1.0f / 65535.0f), (float)(w >> 16) * (1.0f / 65535.0f));
}
__device__ __forceinline__ float2 unpack_snorm2(unsigned w) {
    return make_float2((float)((int)(w << 16) >> 16) * (1.0f / 32767.0f), (float)((int)w >> 16) * (1.0f / 32767.0f));
}

// two adjacent values -> packed bf16 pairs: hi = (bf16(x1) << 16 | bf16(x0)), lo likewise for the residuals
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef __bf16 b2v __attribute__((ext_vector_type(2)));
    const f2v v = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2v));
    const f2v d = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(d, b2v));
}

// Same-XCD fast path.  Correctness never depends on placement: every member announces the XCD it runs on through the
// placement-independent protocol (sc1 store / sc1 polls); only if ALL members of the cluster report the same XCD do the
// per-step payload and flag stores drop the write-through bit -- that XCD's L2 is then the coherence point for writers
// (plain stores are acknowledged by L2) and readers (sc1 loads bypass L1 and are served by L2), and a step's hand-off
// costs L2 round trips instead of trips through the fabric.  1 = same XCD, 0 = not, -1 = gave up (status raised).
// Every wave of the workgroup must call it (it contains two workgroup barriers).
__device__ __forceinline__ int cluster_same_xcd(unsigned* hello, int NC, int c, unsigned* status, unsigned* soft = nullptr) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = 0x100u | (xcc & 0xffu);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) st_agent(hello + c, xcc);
    int verdict = 1;
    if (w == 0) {
        for (unsigned spins = 0;; ++spins) {
            const unsigned v = lane < NC ? ld_agent(hello + lane) : xcc;
            if (__all(v != 0)) { verdict = __all(v == xcc) ? 1 : 0; break; }
            if (soft) {
                // giving up: take the own hello word back FIRST, so that a member dispatched in this very window can never see a
                // complete cluster with a departed member in it; whoever got past the hello regardless leaves
                // quietly at its first flag wait (wait_flags polls the soft word too)
                if (spins > HELLO_LIMIT_SOFT) { if (lane == 0) { st_agent(hello + c, 0); st_agent(soft, 1); } verdict = -1; break; }
                if ((spins & 15) == 15 && ld_agent(soft) != 0) { verdict = -1; break; }
            } else if (spins > SPIN_LIMIT) { st_agent(status, 5); verdict = -1; break; }
            if ((spins & 63) == 63 && ld_agent(status) != 0) { verdict = -1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    const int dead = __syncthreads_or(verdict < 0);
    const int same = __syncthreads_and(verdict == 1);
    return dead ? -1 : same;
}

// Poll the NC epoch flags of this cluster (lane i < NC reads flag i) until all reached `epoch` (flags only grow).
// Called by every wave; returns false when the bounded spin gave up or another workgroup raised the status word --
// the wave then simply leaves (the hardware barrier counts live waves only; the other waves give up the same way).
// `soft` (kernels with a fallback behind them): a set soft word means another cluster gave this launch up -- leave without raising
// the status, the fallback kernels redo the work.
__device__ __forceinline__ bool wait_flags(unsigned* tflags, int NC, unsigned epoch, unsigned* status, unsigned code, unsigned* soft = nullptr) {
    const int lane = threadIdx.x & 63;
    for (unsigned spins = 0;; ++spins) {
        const bool ok = lane >= NC || ld_agent(tflags + lane) >= epoch;
        if (__all(ok)) return true;
        if (spins > SPIN_LIMIT) { st_agent(status, code); return false; }
        if ((spins & 63) == 63 && (ld_agent(status) != 0 || (soft && ld_agent(soft) != 0))) return false;
        // no s_sleep between polls: the flag line's round trip (~500 ticks under load) paces the loop by itself, and the 64 clocks
        // the sleep added per failed poll were on every step's chain (A/B, three pairs: fused forward 1.028 -> 1.022 ms, backward
        // sweeps 1.575 -> 1.567 ms per step)
    }
}

// An LDS counter that the streaming waves of the fused kernels watch ("the critical waves have issued their fragment requests").  The
// accesses MUST be LDS instructions: through a generic pointer hipcc emits flat_load / flat_atomic, which travel the VECTOR memory path
// -- the poll then queues behind the very HBM traffic it is there to schedule around (and its s_waitcnt vmcnt(0) waits for all of it).
typedef __attribute__((address_space(3))) unsigned lds_u32;
__device__ __forceinline__ void sig_raise(unsigned* sig) { __hip_atomic_fetch_add((lds_u32*)sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned sig_read(unsigned* sig) { return __hip_atomic_load((lds_u32*)sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Non-temporal access to the sweeps' ONE-TOUCH HBM streams (round 4).  tools/micro/l2wb.hip + PMC: stream stores (and, less so, loads)
// that allocate in the XCD's L2 evict the exchange payload -- rewritten in place every other step -- before its next rewrite, and
// every eviction is a write-back to HBM plus fabric traffic in front of the hand-off; with the nt bit the payload stays.
// (__builtin_nontemporal_load / _store compile to PLAIN accesses here; the buffer intrinsics' aux bit 1 does set `nt`.)
// An NtArr addresses ONE tile's rows of an array (base = the tile's first row), so 32-bit offsets always suffice.
struct NtArr { __amdgpu_buffer_rsrc_t rs; const char* base; };
__device__ __forceinline__ NtArr nt_arr(const void* arr, size_t first_byte, size_t bytes) {
    NtArr a;
    a.base = reinterpret_cast<const char*>(arr) + first_byte;
    a.rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.base, 0, (unsigned)(bytes < 0xfffffff0ull ? bytes : 0xfffffff0ull), 0x00020000);
    return a;
}
__device__ __forceinline__ f32x4 nt_ld4(const NtArr& a, const float* q) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(a.rs, (unsigned)(reinterpret_cast<const char*>(q) - a.base), 0, 2 /* nt */);
    return __builtin_bit_cast(f32x4, v);
}
// 8 bytes (four 16-bit fixed-point values) at byte address q
__device__ __forceinline__ float2 nt_ld2w(const NtArr& a, const void* q) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(a.rs, (unsigned)(reinterpret_cast<const char*>(q) - a.base), 0, 2 /* nt */);
    return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
}
__device__ __forceinline__ void nt_st2w(const NtArr& a, void* q, unsigned w0, unsigned w1) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = {w0, w1};
    __builtin_amdgcn_raw_buffer_store_b64(v, a.rs, (unsigned)(reinterpret_cast<const char*>(q) - a.base), 0, 2 /* nt */);
}
__device__ __forceinline__ void nt_st4(const NtArr& a, float* q, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), a.rs, (unsigned)(reinterpret_cast<const char*>(q) - a.base), 0, 2 /* nt */);
}

inline int nofast_env() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DEP_CLUSTER_NOFAST"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}
inline int trace_env() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DEP_TRACE"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}

}  // namespace depc
