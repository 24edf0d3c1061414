// Internal helpers shared by the gfx950 kernels of libdep_rnn.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dep_rnn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void dep_set_error(const char* fmt, ...);

#define DEP_CHECK_ARG(cond)                                                          \
    do {                                                                             \
        if (!(cond)) {                                                               \
            dep_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond);     \
            return DEP_ERR_ARG;                                                      \
        }                                                                            \
    } while (0)

#define DEP_CHECK_LAUNCH()                                                           \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            dep_set_error("%s:%d: HIP: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return DEP_ERR_HIP;                                                      \
        }                                                                            \
    } while (0)

static inline size_t dep_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int dep_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device math ------------------------------------------------------------------
__device__ __forceinline__ float dep_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- Philox4x32-10 counter-based RNG (dropout masks) --------------------------------
// key = seed ; counter = (group index lo, hi, site, 0).  One call yields the 4 uniforms of the
// 4-element group `g4` (element indices 4*g4 .. 4*g4+3) of dropout site `site`.
__device__ __forceinline__ void dep_philox4(uint64_t seed, uint32_t site, uint64_t g4, uint32_t (&r)[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)g4, c1 = (uint32_t)(g4 >> 32), c2 = site, c3 = 0x2545F491u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
// pre-scaled keep mask (0 or 1/(1-p)) for the 4 elements of group g4
__device__ __forceinline__ f32x4 dep_dropmask4(uint64_t seed, uint32_t site, uint64_t g4, float p, float scale) {
    uint32_t r[4];
    dep_philox4(seed, site, g4, r);
    f32x4 m;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u = (float)(r[i] >> 8) * (1.0f / 16777216.0f);   // [0,1)
        m[i] = (u >= p) ? scale : 0.0f;
    }
    return m;
}
__device__ __forceinline__ float dep_dropmask1(uint64_t seed, uint32_t site, uint64_t idx, float p, float scale) {
    const f32x4 m = dep_dropmask4(seed, site, idx >> 2, p, scale);
    return m[idx & 3];
}

// dropout sites (Philox counter word 2): distinct per place a mask is drawn in one step
enum { DEP_SITE_RNN0 = 16 /* + layer */, DEP_SITE_USER = 0 };

// ---- XCD-aware tile order -----------------------------------------------------------------------
// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed, speed only).  Remap the
// linear block id so that every XCD walks a contiguous run of the (x fastest, y, z) tile order: neighbouring
// tiles share an operand panel and now hit the same 4 MB L2 instead of refetching it on 6-8 different XCDs.
// Bijective for any block count (cdna_hip_programming.md T1).
__device__ __forceinline__ void dep_xcd_tile(int gx, int gy, int gz, int& bx, int& by, int& bz) {
    const int n = gx * gy * gz, orig = blockIdx.x;
    const int q = n / 8, r = n % 8, xcd = orig % 8;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    bx = id % gx; by = (id / gx) % gy; bz = id / (gx * gy);
}

// ---- launch-instance log (dep_instance_log_*, include/dep_rnn.h) ------------------------------------
// Every kernel launch of the library goes through DEP_LAUNCH: while the log is on it records the kernel expression as written at the
// launch site plus the enclosing function's signature (which carries the template arguments of templated launchers), once per distinct
// pair.  tests/test_instance_coverage_gpu.py uses it to tie the template instances a bench.py step launches to the ones the
// oracle-comparing tests launched.  Off (the default): one relaxed atomic load per launch.
bool dep_ilog_on();
void dep_ilog_note(const char* kern, const char* where);
// (dep_order_log_*: the same hook also appends "K <kernel>" to the enqueue-order log while that is on; collectives add "C ..." in comm.hip)
void dep_olog_add(char kind, const char* text, long n);
#define DEP_LAUNCH(kern, grid, blk, lds, stream, ...)                                          \
    do {                                                                                       \
        if (dep_ilog_on()) dep_ilog_note(#kern, __PRETTY_FUNCTION__);                          \
        hipLaunchKernelGGL(kern, grid, blk, lds, stream, __VA_ARGS__);                         \
    } while (0)

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg) ----
enum { DEP_PROF_GRU_FWD = 0, DEP_PROF_GRU_BWD = 1, DEP_PROF_LSTM_FWD = 2, DEP_PROF_LSTM_BWD = 3,
       DEP_PROF_GEMM_NT = 4, DEP_PROF_GEMM_NN = 5, DEP_PROF_GEMM_TN = 6, DEP_PROF_NCAT = 7 };
// GEMM predicate of the calling thread: launches of dep_gemm_internal made while it is set return at kernel entry unless the
// device word is non-zero (the conditional fallback of dep_rnn_forward).  nullptr = unconditional.
void dep_gemm_set_predicate(const unsigned* only_if);
const unsigned* dep_gemm_predicate();
// A-operand column skip of the calling thread's next TN contractions (bf16x3 kernel only): logical column m of op(A) is stored
// column m + (m >= at ? by : 0).  dW_hh of a GRU reads [dr | dz] and [dn*r] out of [dr | dz | dn | dn*r] with (2H, H).  (0, 0) = off.
void dep_gemm_set_a_colskip(int at, int by);
// Storage format of the calling thread's next contractions' operands (gemm_bf16x3.hip): 0 = fp32, 1 = PK, the pre-split row-pair
// (hi, lo) bf16 image a producer kernel wrote in place of the fp32 array.  Only the bf16x3 kernel reads PK; dep_gemm_internal
// refuses (DEP_ERR_ARG) a PK operand on any other path.  Reset to (0, 0) after the calls.
void dep_gemm_set_operand_formats(int fmt_a, int fmt_b);
void dep_gemm_set_scratch(void* p, size_t bytes);      // per calling thread: scratch of its unsplit contractions (gemm.hip)
bool dep_gemm_pk_pending();
bool dep_gemm_bf16x3_pair_ok();
// true when dep_gemm_internal would run the bf16x3 kernel for a contraction of this size (it is the one that honours the skip)
bool dep_gemm_uses_bf16x3(int M, int N, int K, int seq_T);
void dep_gemm_set_split_target(long target);      // split-K target of this thread's next contractions (0 = default); gemm.hip
// two TN contractions of equal shape over PK A operands in one launch (gemm.hip): 1 = enqueued, 0 = not covered, < 0 = error
int dep_gemm_tn_pair(int M, int N, int K, const float* A0, const float* A1, int lda, int skip_at1, int skip_by1,
                     const float* B0, int ldb0, int seq_T0, int shift0, const float* B1, int ldb1, int seq_T1, int shift1,
                     float* C0, int ldc0, float* C1, int ldc1, void* ws, size_t ws_bytes, hipStream_t s);
int dep_gemm_bf16x3_tn_pair_launch(int M, int N, int K, const float* A0, const float* A1, int lda, int skip_at1, int skip_by1,
                                   const float* B0, int ldb0, int seq_T0, int shift0, const float* B1, int ldb1, int seq_T1, int shift1,
                                   float* C0, int ldc0, float* C1, int ldc1, int splits, int kchunk, float* part0, float* part1, hipStream_t s);
// process-wide: may dep_rnn_forward use kernels that need every CU to themselves (dep_rnn_set_exclusive, include/dep_rnn.h)
bool dep_exclusive_on();
bool dep_prof_on();
void dep_prof_begin(int cat, hipStream_t s);
void dep_prof_end(hipStream_t s);
struct DepProfScope {
    hipStream_t s; bool on;
    DepProfScope(int cat, hipStream_t st, bool enable = true) : s(st), on(enable && dep_prof_on()) { if (on) dep_prof_begin(cat, s); }
    ~DepProfScope() { if (on) dep_prof_end(s); }
};

// ---- internal launchers shared across translation units ---------------------------
struct dep_sweep_args {
    int B, T, H;
    int cell, dirs;          // GRU: dirs = 1
    int training;
    int impl;
    // per direction d (LSTM) / single (GRU)
    const float* w_hh[2];    // (G*H, H) row-major
    const float* b_hh[2];    // (G*H)  GRU only (LSTM biases are folded into GI by the GEMM)
    const float* wp[2];      // packed fragment-order copies (MFMA path), see pack kernels
    int split;               // cluster sweeps: recurrent product on the bf16 matrix cores (3-term split), wp packed to match
    // activations
    const float* gi;         // (B,T,dirs*G*H): input projection incl. bias_ih (LSTM: + bias_hh)
    float* y;  int ldy;      // (B,T,ldy) hidden sequence; direction d writes columns [d*H,(d+1)*H)
    float* ydrop;            // same layout: dropout(y) for the next layer, or NULL
    float drop_p; uint64_t seed; uint32_t site;
    float* pooled; float pool_scale;    // GRU top layer: (B,H) sum_t h_t * scale, or NULL
    float* h_n;              // (dirs,B,H) final states or NULL
    // reserve (training): GRU r,z,n,hn each (B,T,H) ; LSTM gates (B,T,dirs*4H) + c (B,T,dirs*H)
    float* sv0; float* sv1; float* sv2; float* sv3;
    const unsigned* only_if;  // cluster forward: run only when this device word is non-zero (fallback behind an exclusive kernel), or NULL
    int sv16;                 // GRU cluster sweeps: sv0..sv2 (r, z, n) are 16-bit fixed point (rnn_cluster_common.h), sv3 (hn) stays fp32
    int hdr_slot, hdr_clean;  // cluster sweeps: exchange-header slot of this launch; clean = the caller zeroed it (rnn_cluster_common.h)
    hipStream_t stream;
};
int dep_launch_sweep_fwd(const dep_sweep_args& a);

struct dep_sweep_bwd_args {
    int B, T, H;
    int cell, dirs;
    int impl;
    const float* w_hh[2];
    const float* wpT[2];     // packed transposed copies (MFMA path)
    int split;               // GRU cluster sweep: dgates W_hh on the bf16 matrix cores (3-term split), wpT packed to match
    const float* y; int ldy; // forward hidden sequence of this layer (h_{t-1} operand)
    const float* dy; int lddy;           // (B,T,lddy) grad of y (columns [d*H,(d+1)*H) per direction) or NULL
    float drop_p; uint64_t seed; uint32_t site;   // dropout applied to dy on load (p == 0: none)
    const float* dpooled; float pool_scale;       // GRU top layer or NULL
    const float* dh_n;       // (dirs,B,H) or NULL
    const float* sv0; const float* sv1; const float* sv2; const float* sv3;
    float* dgi;              // (B,T,dirs*G*H) written: grad of the input projection
    float* dghn;             // GRU: (B,T,H) grad of the n-gate recurrent pre-activation (dn*r)
    int lddg, lddghn;        // row strides of dgi / dghn (0 = packed: dirs*G*H and H).  The GRU cluster sweep accepts 4H / 4H with
                             // dghn = dgi + 3H: one (B,T,4H) array [dr | dz | dn | dn*r], so that dW_hh is ONE contraction
    float* dbpart;           // partial bias sums, see dep_sweep_dbpart_floats
    int dbpart_rows;         // number of partial rows provided
    int hdr_slot, hdr_clean; // cluster sweeps: exchange-header slot of this launch; clean = the caller zeroed it
    int sv16;                // saved gates r, z, n are 16-bit fixed point (must match the forward that wrote them)
    int dg_pk;               // GRU cluster sweep (burst kernel, 4H-wide rows): dgi / dghn as the PK image of gemm_bf16x3.hip instead of fp32
    int bf16st;              // bf16-storage mode (dep_set_gemm_mode(3)): y and hn are bf16 arrays, the gate gradients the PKH image (hi rows only)
    hipStream_t stream;
};
bool dep_cluster_bwd_pk_ok(int H, int T);
#define DEP_FUSED2_BWD_DEFAULT 1      /* both GRU layers' BPTT as one all-gather launch (rnn_fused2_bwd.hip); DEP_FUSED2_BWD overrides */
bool dep_cluster_lstm_bwd_pk_ok(int T);
bool dep_cluster_lstm_sv16_ok();
int dep_launch_sweep_bwd(const dep_sweep_bwd_args& a);
int dep_sweep_num_wg(int B, int H, int impl);       // batch tiles (rows of dbpart) per direction
bool dep_sweep_use_mfma(int H, int impl);

// packed weight sizes / kernels (MFMA path)
size_t dep_pack_floats(int G, int H);               // floats of one packed (G*H x H) matrix
int dep_pack_whh(const float* w_hh, float* wp, float* wpT, int G, int H, hipStream_t s);
// bias-gradient finish: sums partial rows
int dep_finish_db(const dep_sweep_bwd_args& a, float* const* db_ih, float* const* db_hh);

// cluster-parallel sweeps (rnn_cluster.hip)
bool dep_cluster_ok(int cell, int H, int B, int dirs);
size_t dep_cluster_xbuf_bytes(int cell, int H, int B, int dirs);
int dep_pack_cluster_bwd(const float* w_hh, float* out, int G, int H, hipStream_t s);
int dep_launch_cluster_fwd(const dep_sweep_args& a, void* xbuf, size_t xbuf_bytes);
int dep_launch_cluster_bwd(const dep_sweep_bwd_args& a, void* xbuf, size_t xbuf_bytes);
bool dep_cluster16_ok(int cell, int H, int B);
bool dep_cluster_lstm_ok(int H, int B, int dirs);
size_t dep_cluster_lstm_xbuf_bytes(int H, int B, int dirs);
int dep_launch_cluster_lstm_fwd(const dep_sweep_args& a, void* xbuf, size_t xbuf_bytes);
int dep_launch_cluster_lstm_bwd(const dep_sweep_bwd_args& a, void* xbuf, size_t xbuf_bytes);
// Utterances one co-resident cluster launch may cover: `members` workgroups per 16-utterance tile, `per_cu` of them resident
// per CU (rnn_cluster.hip; CU count from the device, DEP_NUM_CUS overrides), at most `max_wgs` workgroups (flag words).
int dep_cluster_chunk(int members, int per_cu, int max_wgs);
// rnn_fused2.hip: both layers of a 2-layer GRU (H = 256) in one cluster launch, layer 1 one step behind layer 0
struct dep_fused2_args {
    int B, T, training;
    const float* wp0; const float* wp1; const float* wpi;   // packed W_hh l0, W_hh l1, W_ih l1 (dep_pack_cluster_fwd_split images)
    const float* b_hh0; const float* b_ih1; const float* b_hh1;
    const float* gi;                                          // layer-0 input projection incl. b_ih (B*T, 3H)
    float* y0; float* y0d; float* y1;
    size_t ostride;                                           // floats between consecutive per-layer arrays of the reserve
    float drop_p; uint64_t seed; uint32_t site;
    float* pooled; float pool_scale; float* hn0; float* hn1;
    float* sv[2][4];
    int sv16;                                                 // saved gates r, z, n as 16-bit fixed point (rnn_cluster_common.h)
    int bf16st;                                               // bf16-storage mode: h, dropout(h), hn are written as bf16 (2-byte elements at the same positions)
    int soft_fallback;                                        // 1: a failed hello sets the workspace's soft flag instead of the status word
    int hdr_clean;                                            // slot 0 was zeroed by the caller (dep_cluster_reset_status)
    hipStream_t stream;
};
bool dep_fused2_ok(int cell, int H, int L, int dirs);
size_t dep_fused2_xbuf_bytes(int B);
int dep_launch_fused2_fwd(const dep_fused2_args& a, void* xbuf, size_t xbuf_bytes);
// rnn_fused2_bwd.hip: BPTT of both layers in one cluster launch, layer 0 one step behind layer 1
struct dep_fused2_bwd_args {
    int B, T;
    const float* wh1; const float* wi1; const float* wh0;     // dep_pack_cluster_bwd_split images of W_hh l1, W_ih l1, W_hh l0
    const float* y1; const float* y0; const float* sv1; const float* sv0; size_t svstride;
    const float* dy; const float* dpooled; float pool_scale; const float* dhn1; const float* dhn0;
    float drop_p; uint64_t seed; uint32_t site;
    float* dgi1; float* dghn1; float* dgi0; float* dghn0; float* dbpart1; float* dbpart0; int dbpart_rows;
    int lddg, lddghn;        // row strides of dgi / dghn: 3H / H, or 4H / 4H with dghn = dgi + 3H (one (B*T, 4H) array [dr | dz | dn | dn*r] per layer)
    int sv16;                // saved gates r, z, n are 16-bit fixed point (must match the forward that wrote them)
    int bf16st;              // bf16-storage mode (dep_set_gemm_mode(3)): y and hn are bf16 arrays, the gate gradients the PKH image (needs sv16 and dg_pk)
    int dg_pk;               // dgi1 / dgi0 are (B*T, 4H) arrays [dr | dz | dn | dn*r] written as the PK image of gemm_bf16x3.hip (needs sv16, T even); dghn unused
    hipStream_t stream;
};
size_t dep_fused2_bwd_xbuf_bytes(int B);
bool dep_fused2_bwd_fits(int B, int T);       // 32-bit offsets into a layer's arrays: else the per-layer sweeps
int dep_launch_fused2_bwd(const dep_fused2_bwd_args& a, void* xbuf, size_t xbuf_bytes);
// comm.hip: all-reduce of [buf, buf+n) on comm_stream after everything enqueued so far on `compute`
int dep_comm_enqueue_after(dep_comm* c, float* buf, long n, hipStream_t compute, hipStream_t comm_stream);
// clears the sticky status word of a cluster exchange buffer (once per dep_rnn_forward; the sweeps themselves never clear it)
constexpr int DEP_HDR_SLOTS = 4;                              // exchange-header slots at the head of the buffer (rnn_cluster_common.h)
int dep_cluster_reset_status(void* xbuf, hipStream_t s);      // status, soft flag and EVERY header slot
int dep_cluster_reset_flags(void* xbuf, hipStream_t s);       // every header slot, status words kept (dep_rnn_backward)
int dep_pack_cluster16_fwd_split(const float* w_hh, float* out, int H, hipStream_t s);
int dep_pack_cluster_bwd_split(const float* w_hh, float* out, int H, hipStream_t s);
int dep_pack_cluster_split_multi(int n, const float* const* src, float* const* dst, const int* bwd, int H, hipStream_t s);
int dep_multi_copy(int count, const float* const* src, const float* const* add, float* const* dst, const long* n, hipStream_t s);
// attention.hip: the loads-in-flight attention kernels (H in {64,128,256}); 1 = launched, 0 = shape left to elementwise.hip's
int dep_attn2_fwd(const float* out, const float* pre, float* ctx, float* alpha, int B, int T, int H, hipStream_t s);
int dep_attn2_bwd(const float* dctx, const float* out, const float* alpha, const float* pre, float* dout, float* dpre, int B,
                  int T, int H, hipStream_t s);
int dep_pack_cluster_fwd_split(const float* w_hh, float* out, int H, hipStream_t s);
int dep_pack_cluster_lstm_split(const float* w_hh, float* wp, float* wpT, int H, hipStream_t s);
int dep_launch_cluster16_fwd(const dep_sweep_args& a, void* xbuf, size_t xbuf_bytes);

int dep_gemm_internal(int transA, int transB, int M, int N, int K, const float* A, int lda,
                      const float* B, int ldb, float* C, int ldc, const float* bias, float beta,
                      int seq_T, int shiftB, void* ws, size_t ws_bytes, hipStream_t s);
