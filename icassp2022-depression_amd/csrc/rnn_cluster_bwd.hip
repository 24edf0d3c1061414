// Cluster-parallel GRU BPTT sweep, flag-published exchange (cdna_hip_programming.md G16 recipe R1).
//
// Same decomposition as rnn_cluster.hip (member c of a 16-utterance tile's cluster keeps the W_hh rows of
// hidden units [32c, 32c+32) in VGPRs and turns ITS 16x96 slice of dgh_t into a partial dh_{t-1} for all H
// columns), but the reduce-scatter of the partials moves 4096 values per member per step, too many for
// 8-byte tagged granules (each narrow write-through store is its own fabric write).  Here the partials are
// plain fp32 written in MFMA-fragment order with wave-contiguous 16-byte sc1 (write-through) stores, every
// wave drains (s_waitcnt vmcnt(0)), the workgroup barriers, and one lane publishes the step's epoch in the
// member's flag word; consumers poll the NC flags with relaxed agent-scope loads (>= epoch: flags only grow)
// and then read their two columns of every member's partial with sc1 loads, summing in member order
// (deterministic).  Payload buffers alternate by step parity; only the status/flag words are zeroed per launch.
#include "rnn_cluster_common.h"

namespace {
using namespace depc;

constexpr size_t EXCLUSIVE_LDS = 84 * 1024;
#define DEP_STAMP(slot) do { if (tr && t >= 100 && t < 104) tr[(t - 100) * 8 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

struct P2 {
    int B, T, H, nbtp, b0;      // b0: first utterance of this launch's batch chunk (B is the whole batch)
    const f32x4* wp;
    const float* y; int ldy;
    const float* dy; int lddy;
    float drop_p, drop_scale; uint64_t seed; uint32_t site;
    const float* dpooled; float pool_scale;
    const float* dh_n;
    const float* sv0; const float* sv1; const float* sv2; const float* sv3;
    float* dgi; int lddg;
    float* dghn; int lddghn;
    float* dbpart;
    unsigned* status; unsigned* flags; unsigned* hello; float* payload; unsigned payload_bytes;
    int nofast;              // DEP_CLUSTER_NOFAST=1: always use the write-through (placement-agnostic) stores
    long long* trace;        // debug: s_memtime stamps of workgroup 0 (DEP_TRACE=1), else nullptr
    int trall_off;           // debug: LDS offset (32-bit words) of the per-step stamp array
    int ntstream;            // non-temporal gate-gradient write-out (default on; DEP_BWD_NT=0: plain stores)
    int ntload;              // non-temporal one-touch input streams (default on; DEP_BWD_NTLD=0: plain loads)
    int wflags;              // DEP_BWD_WFLAGS: every compute wave raises its OWN epoch flag once its own payload stores are acknowledged (no workgroup barrier in front of the flag; the pollers watch 4 NC words)
    int dgpk;                // round 4: write the gate gradients as the PK image the bf16x3 GEMMs read without converting (gemm_bf16x3.hip FMT_PK): rows (t even, t+1) of an utterance hold the (hi, lo) bf16 pairs of both steps; burst kernel, 4H-wide layout, T even
};

struct StepIn { float2 r, z, n, hn, hp, dy; };

// Thread -> element map (matches the fragment order of the published partials, so the gather is coalesced):
//   jl = tid>>7 (which of the member's two 16-column tiles), lp = (tid>>1)&63 (MFMA lane id: row lp&15, quad lp>>4),
//   half = tid&1 -> columns (2c+jl)*16 + (lp>>4)*4 + 2*half + {0,1} of utterance row lp&15.
//
// SPLIT: dgates W_hh runs on the bf16 matrix cores with the 3-term split (see gru_fwd_cluster16): the member's 96 gate
// rows are exactly three 32-wide k-steps (r, z, n), 4 tiles x 3 k-steps x 3 products = 36 v_mfma_f32_16x16x32_bf16
// (576 cycles) replace 96 v_mfma_f32_16x16x4_f32 (3072 cycles, close to half of the step).  The thread that produces a
// gate gradient splits it once and writes the (hi, lo) bf16 planes the MFMA B fragments are read from.
//
// KB > 0: BURST STREAMS.  A CU returns vector loads in issue order ACROSS its waves (tools/micro/inorder.hip: a load that hits
// in L2 comes back after 200 cycles on a quiet CU, after 600-4000 when another wave of the CU has HBM loads or stores in
// flight), so every HBM stream request of a step sits in front of that step's flag polls and gather loads, whichever wave
// issues it: the sweep pays HBM time and exchange latency one after the other.  With KB > 0 four service waves (threads
// CT .. CT+255) own every stream and move KB steps at a time: on a tile's `dirty' step (every KB-th, the tiles of an XCD
// staggered) they request the inputs of KB steps (registers for KB-1 steps, then the LDS ring `ibuf') and write out the
// gate gradients of the last KB steps (LDS ring `obuf'); the other KB-1 steps of the tile run with nothing but the
// exchange in the CU's memory pipeline.  The compute waves read their inputs from ibuf and leave dr, dz, dn, dn*r in obuf.
constexpr int SVC_THREADS = 256;
constexpr int SROW = 36;                             // LDS row stride (floats) of ibuf / obuf: 16-byte aligned rows, conflict-free columns
constexpr int SARR = 16 * SROW;
constexpr int TRACE_F = 2048, IBUF_F = 2304;         // float offsets into the workgroup's LDS (planes: [0, 2048))
constexpr int obuf_slots(int KB) { return KB == 4 ? KB + 2 : KB + 1; }      // KB + 1 does for fp32 rows; the PK flush (KB = 4 only) works on step PAIRS and may lag one step
constexpr size_t burst_lds_bytes(int KB) { return (size_t)(IBUF_F + KB * 6 * SARR + obuf_slots(KB) * 4 * SARR) * sizeof(float); }
// AG (round 5): the all-gather form of the exchange.  Member c keeps the W_hh COLUMNS of its 32 hidden units (all 3H rows), every member
// publishes its own 16 x 96 gate gradients dgh_t = (dr, dz, dn*r) ONCE, as the (hi, lo) bf16 planes the MFMAs read, in MFMA B-fragment order
// (6 KB per member and step instead of 16 KB of fp32 partial dh; no sum over members at the consumer), and computes its own 32 columns of
// dh_{t-1} = dgh_t W_hh completely: K = 3H = 768 split into four quarters over the four compute waves (wave w = the 192 k of source members
// 2w, 2w+1: 6 k-steps x 2 output tiles x 3 products = the same 36 MFMAs per wave), the B fragments read STRAIGHT from the exchange buffer
// into registers (12 wave-contiguous 1 KB loads per wave; no LDS staging), the four K-quarter partials summed through a double-buffered
// LDS block `red' behind the step's only workgroup barrier.  LDS: red occupies [0, 4096) floats, so the trace / ring offsets move up.
constexpr int AG_TRACE_F = 4096, AG_IBUF_F = 4352, AG_MEMBER_BYTES = 6 * 1024;
constexpr size_t burst_lds_bytes_ag(int KB) { return (size_t)(AG_IBUF_F + KB * 6 * SARR + obuf_slots(KB) * 4 * SARR) * sizeof(float); }

// SV16: the saved gates r, z, n are 16-bit fixed point (rnn_cluster_common.h).  A template parameter, not a kernel argument: as a
// run-time switch the two load widths met in copies of the loaded registers and the service waves waited for every load they had
// just issued (both settings 5-20 % slower than the kernel without the switch, profiles/r04_ab_pairs.txt).
// BF (bf16-storage mode, dep_set_gemm_mode(3); implies SV16, burst kernel only): hn and the hidden sequence y are bf16 arrays (2-byte
// elements at the same positions), and the gate gradients go out as the PKH image (only the hi rows of the PK image).
template <int NTW, bool SPLIT, int KB, bool SV16 = false, bool BF = false, bool AG = false>      // output tiles per wave = H/64
__global__ __launch_bounds__(KB ? CT + SVC_THREADS : CT) void gru_bwd_cluster_r1(P2 p) {
    static_assert(!BF || (SV16 && KB > 0), "bf16 storage: 16-bit gates, burst kernel");
    static_assert(!AG || (SPLIT && KB == 4 && NTW == 4), "all-gather exchange: H = 256, split products, burst length 4");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 96, KCB = KS / 16, LDG = KS + LPAD;
    constexpr int LDGB = KS + 8;                      // bf16 elements per row of a split plane (208-byte rows)
    const int H = p.H, T = p.T, NC = H / 32, NTT = H / 16;
    const int c = blockIdx.x / p.nbtp, bt = blockIdx.x % p.nbtp;
    if (p.b0 + bt * BT >= p.B) return;
    if (ld_agent(p.status) != 0) return;           // an earlier sweep of this step gave up: the status word is sticky until the next dep_rnn_forward
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jl = tid >> 7, lp = (tid >> 1) & 63, half = tid & 1;
    const int j = lp & 15, ul = jl * 16 + (lp >> 4) * 4 + 2 * half;      // unit inside the member's 32
    const int col = 32 * c + ul;
    const int b = p.b0 + bt * BT + j;
    const bool valid = b < p.B;
    float* dgs = smem;                                                     // [16][LDG] fp32, or (SPLIT) two bf16 planes [16][LDGB]
    unsigned short* dg_hi = reinterpret_cast<unsigned short*>(smem);
    unsigned short* dg_lo = dg_hi + BT * LDGB;

    constexpr bool BURST = KB > 0;
    constexpr int KBX = BURST ? KB : 1;
    const bool svc = BURST && tid >= CT;              // wave-uniform
    float* ibuf = smem + (AG ? AG_IBUF_F : IBUF_F);   // [KB][6][16][SROW]: r, z, n, hn, h_{t-1}, dy of step k in slot k % KB
    constexpr int OSL = obuf_slots(KBX);
    float* obuf = ibuf + KBX * 6 * SARR;              // [OSL][4][16][SROW]: dr, dz, dn, dn*r of step k in slot k % OSL
    f32x4 wr[SPLIT ? 1 : NTW][SPLIT ? 1 : KCB];
    u32x4 wq[SPLIT ? NTW : 1][SPLIT ? 3 : 1][2];       // [tile][k-step = gate][hi, lo]   (AG: unused, wa below)
    u32x4 wa[AG ? 2 : 1][AG ? 6 : 1][2];               // AG: [own output tile][k-step = (source member 2w + ks/3, gate ks%3)][hi, lo]
    if (!svc) {
        if constexpr (AG) {
            // the SAME packed image as the reduce-scatter kernel's ([K member][output tile][gate][plane][lane]); this member takes the
            // two output tiles of its own 32 columns and, per wave, the K rows of source members 2w, 2w+1
            const u32x4* wpq = reinterpret_cast<const u32x4*>(p.wp);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        wa[i][ks][pl] = wpq[(size_t)((((2 * w + ks / 3) * NTT + 2 * c + i) * 3 + ks % 3) * 2 + pl) * 64 + lane];
        } else if constexpr (SPLIT) {
            const u32x4* wpq = reinterpret_cast<const u32x4*>(p.wp);
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        wq[i][ks][pl] = wpq[(size_t)(((c * NTT + w * NTW + i) * 3 + ks) * 2 + pl) * 64 + lane];
        } else {
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int k = 0; k < KCB; ++k)
                    wr[i][k] = p.wp[(size_t)((c * NTT + w * NTW + i) * KCB + k) * 64 + lane];
        }
    }
    float2 dhrec = (p.dh_n && valid && !svc) ? ld2(p.dh_n + (size_t)b * H + col) : f2(0.f, 0.f);
    float2 dpl = f2(0.f, 0.f);
    if (p.dpooled && valid && !svc) { dpl = ld2(p.dpooled + (size_t)b * H + col); dpl.x *= p.pool_scale; dpl.y *= p.pool_scale; }
    float2 dbr = f2(0.f, 0.f), dbz = dbr, dbn = dbr, dbh = dbr;
    const size_t pstride = (size_t)p.nbtp * NC * BT * H;                   // floats per parity buffer
    const size_t tile_base = (size_t)bt * NC * BT * H;                     // this tile's [NC][NTT][64][4] block
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.payload, 0, p.payload_bytes, 0x00020000);
    const bool wf = p.wflags != 0;                    // uniform
    unsigned* tflags = wf ? p.flags + bt * NC * 4 : p.flags + bt * NC;
    unsigned* myflag = wf ? tflags + c * 4 + (w & 3) : tflags + c;
    const int nflags = wf ? NC * 4 : NC;
    const int ml = lane & 15, mq = lane >> 4;
    const int sx = p.nofast ? 0 : cluster_same_xcd(p.hello + bt * NC, NC, c, p.status);
    if (sx < 0) return;
    const bool fast = sx == 1;

    auto load_step = [&](int t, StepIn& s) {
        s.r = s.z = s.n = s.hn = s.hp = s.dy = f2(0.f, 0.f);
        if (valid && t >= 0) {
            const size_t row = (size_t)b * T + t;
            const size_t so = row * H + col;
            if constexpr (SV16) {
                s.r = unpack_unorm2(*reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(p.sv0) + so));
                s.z = unpack_unorm2(*reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(p.sv1) + so));
                s.n = unpack_snorm2(*reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(p.sv2) + so));
            } else { s.r = ld2(p.sv0 + so); s.z = ld2(p.sv1 + so); s.n = ld2(p.sv2 + so); }
            s.hn = ld2(p.sv3 + so);
            if (t > 0) s.hp = ld2(p.y + (row - 1) * p.ldy + col);
            if (p.dy) s.dy = ld2(p.dy + row * p.lddy + col);
        }
    };
    StepIn cur, nxt;
    if constexpr (!BURST) load_step(T - 1, cur);
    // ---- service waves.  Step counter k = T-1-t.  Service thread st: piece idx = st + 256 i (i < 3) of a step's 768 input
    // pieces -> array idx / 128, utterance row (idx % 128) / 8, 16-byte piece idx % 8; likewise 512 write-out pieces (i < 2).
    const int st = tid - CT, sarr0 = (st >> 7) & 1, sr = (st >> 3) & 15, sp = st & 7;
    const int sb = p.b0 + bt * BT + sr;
    const bool svalid = sb < p.B;
    const int scol = 32 * c + sp * 4;
    const int phi = (bt >> 3) % KBX;                  // the tiles of an XCD (bt = xcd mod 8) take their dirty steps in turn
    f32x4 sreg[KBX][3];
    // wave-uniform array choice (waves 4, 5: arrays 0, 2, 4 = r, n, h_{t-1}; waves 6, 7: 1, 3, 5 = z, hn, dy), made scalar so
    // that the base pointers are selected in SGPRs (a per-lane choice makes hipcc index the kernel arguments in memory and
    // wait for that pointer load -- vmcnt(0) -- in front of every data load)
    const bool sodd = __builtin_amdgcn_readfirstlane(sarr0) != 0;
    // SV16: the wave pair 4, 5 (sodd = 0) streams the three fp32 arrays hn, h_{t-1}, dy (ring arrays 3, 4, 5), the pair 6, 7 the three
    // 16-bit arrays r, z, n (ring arrays 0, 1, 2; 8-byte pieces, decoded when they go into the ring): one load width per wave
    const float* sbase0 = SV16 ? (sodd ? p.sv0 : p.sv3) : (sodd ? p.sv1 : p.sv0);
    const float* sbase1 = SV16 ? (sodd ? p.sv1 : p.y) : (sodd ? p.sv3 : p.sv2);
    const float* sbase2 = SV16 ? (sodd ? p.sv2 : p.dy) : (sodd ? p.dy : p.y);
    const int sld2 = sodd ? p.lddy : p.ldy;
    // DEP_BWD_NT=1: the service waves' one-touch streams carry the non-temporal hint, so that they do not displace the
    // exchange payload (rewritten every other step) from this XCD's L2 and turn it into HBM write-backs
    // Round 4: the gate-gradient write-out goes out NON-TEMPORAL (buffer stores with the nt bit).  tools/micro/l2wb.hip + PMC:
    // one-touch stores that allocate in L2 are what evicts the exchange payload (1 MB per XCD, rewritten in place every other step)
    // before its next rewrite -- 0.8 GB of write-backs per launch; with nt stores the payload stays.  (Round 3's DEP_BWD_NT went
    // through __builtin_nontemporal_store, which hipcc compiled to PLAIN stores -- no nt bit in the ISA -- so it measured nothing.)
    const bool snt = p.ntstream != 0;
    // (resources are TILE-relative -- base = the tile's first row, extent = its 16 utterances -- so 32-bit offsets always suffice)
    const size_t trow0 = (size_t)(p.b0 + bt * BT) * T;
    float* const dgi_t = p.dgi + trow0 * p.lddg; float* const dghn_t = p.dghn + trow0 * p.lddghn;
    __amdgpu_buffer_rsrc_t rs_dgi = __builtin_amdgcn_make_buffer_rsrc((void*)dgi_t, 0, (unsigned)((size_t)BT * T * p.lddg * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_dghn = __builtin_amdgcn_make_buffer_rsrc((void*)dghn_t, 0, (unsigned)((size_t)BT * T * p.lddghn * 4), 0x00020000);
    // q inside the tile's rows of the array of `rs` (tile base `base`): 16-byte store, nt when enabled
    auto stnt2 = [&](__amdgpu_buffer_rsrc_t rs, const float* base, float* q, f32x4 v) {
        if (snt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)((const char*)q - (const char*)base), 0, 2 /* nt */);
        else *reinterpret_cast<f32x4*>(q) = v;
    };
    // ... and so do the service waves' one-touch READS (the saved gates, h_{t-1}, dy): p.ntload (DEP_BWD_NTLD, default on).  NT is a
    // compile-time tag of the service waves' code (svc_life below), not a select per load: two load forms meeting in one register
    // make hipcc wait for every load it has just issued.
    const int mld = H > p.ldy ? (H > p.lddy ? H : p.lddy) : (p.ldy > p.lddy ? p.ldy : p.lddy);
    const unsigned span = (unsigned)((size_t)(BT * T + 1) * mld * 4);        // bounds the tile's rows of every input array (+ the h_{t-1} row in front)
    // bases one row in front of the tile's first row (h_{t-1} of t = 0 is never read; the offset stays non-negative)
    // (the 16-bit arrays of the SV16 wave pair have 2-byte elements: their rows are H / 2 floats apart)
    const bool w16 = SV16 && sodd;
    // byte offset of the tile's first row in each of this wave's three input arrays (element size x row stride)
    const size_t es0 = (w16 || BF) ? 2 : 4, es1 = (w16 || BF) ? 2 : 4, es2 = w16 ? 2 : 4;
    const size_t rs1 = SV16 ? (sodd ? (size_t)H : (size_t)p.ldy) : (size_t)H, rs2 = SV16 ? (sodd ? (size_t)H : (size_t)p.lddy) : (size_t)sld2;
    const float* const in0_t = sbase0 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(sbase0) + trow0 * H * es0) : nullptr;
    const float* const in1_t = sbase1 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(sbase1) + trow0 * rs1 * es1) : nullptr;
    const float* const in2_t = sbase2 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(sbase2) + trow0 * rs2 * es2) : nullptr;
    __amdgpu_buffer_rsrc_t rs_in0 = __builtin_amdgcn_make_buffer_rsrc((void*)in0_t, 0, span, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_in1 = __builtin_amdgcn_make_buffer_rsrc((void*)in1_t, 0, span, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc((void*)in2_t, 0, span, 0x00020000);
    auto ldnt = [&](auto NT, int which, const float* q) -> f32x4 {      // 16 bytes at q, inside input array `which` of this wave
        if constexpr (decltype(NT)::value) {
            const float* base = which == 0 ? in0_t : (which == 1 ? in1_t : in2_t);
            const unsigned off = (unsigned)((const char*)q - (const char*)base);
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(which == 0 ? rs_in0 : (which == 1 ? rs_in1 : rs_in2), off, 0, 2 /* nt */);
            return __builtin_bit_cast(f32x4, v);
        } else {
            return ld4(q);
        }
    };
    auto ldnt8 = [&](auto NT, int which, const unsigned short* q) -> float2 {
        if constexpr (decltype(NT)::value) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const float* base = which == 0 ? in0_t : (which == 1 ? in1_t : in2_t);
            const unsigned off = (unsigned)((const char*)q - (const char*)base);
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(which == 0 ? rs_in0 : (which == 1 ? rs_in1 : rs_in2), off, 0, 2 /* nt */);
            return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
        } else {
            return *reinterpret_cast<const float2*>(q);
        }
    };
    // W16 (a std::bool_constant): this wave streams the 16-bit arrays
    auto svc_load1 = [&](auto W16, auto NT, int k, int i) -> f32x4 {     // input i of this wave for step k
        const int t = T - 1 - k;
        if (!svalid || t < 0) return zero4();
        const size_t row = (size_t)sb * T + t;
        if constexpr (SV16) {
            if constexpr (decltype(W16)::value) {             // four 16-bit values = 8 bytes
                const float* base = i == 0 ? sbase0 : (i == 1 ? sbase1 : sbase2);
                const float2 w = ldnt8(NT, i, reinterpret_cast<const unsigned short*>(base) + row * H + scol);
                const f32x4 r4 = {w.x, w.y, 0.f, 0.f};
                return r4;
            } else {
                if constexpr (BF) {                           // hn, h_{t-1}: bf16 arrays, four values = 8 bytes (decoded in svc_put)
                    if (i == 0) { const float2 w = ldnt8(NT, 0, reinterpret_cast<const unsigned short*>(sbase0) + row * H + scol); const f32x4 r4 = {w.x, w.y, 0.f, 0.f}; return r4; }
                    if (i == 1) {
                        if (t <= 0) return zero4();
                        const float2 w = ldnt8(NT, 1, reinterpret_cast<const unsigned short*>(sbase1) + (row - 1) * p.ldy + scol);
                        const f32x4 r4 = {w.x, w.y, 0.f, 0.f}; return r4;
                    }
                    return sbase2 ? ldnt(NT, 2, sbase2 + row * p.lddy + scol) : zero4();
                } else {
                    if (i == 0) return ldnt(NT, 0, sbase0 + row * H + scol);
                    if (i == 1) return t > 0 ? ldnt(NT, 1, sbase1 + (row - 1) * p.ldy + scol) : zero4();
                    return sbase2 ? ldnt(NT, 2, sbase2 + row * p.lddy + scol) : zero4();
                }
            }
        } else {
            if (i == 0) return ldnt(NT, 0, sbase0 + row * H + scol);
            if (i == 1) return ldnt(NT, 1, sbase1 + row * H + scol);
            if (sodd) return sbase2 ? ldnt(NT, 2, sbase2 + row * sld2 + scol) : zero4();
            return t > 0 ? ldnt(NT, 2, sbase2 + (row - 1) * sld2 + scol) : zero4();
        }
    };
    auto svc_issue = [&](auto W16, auto NT, int k0, int n) {   // inputs of steps k0 .. k0+n-1 -> registers
#pragma unroll
        for (int d = 0; d < KBX; ++d)
            if (d < n) {
#pragma unroll
                for (int i = 0; i < 3; ++i) sreg[d][i] = svc_load1(W16, NT, k0 + d, i);
            }
    };
    auto svc_put = [&](auto W16, int k0, int n, int dlo = 0) {     // registers -> ibuf slots of steps k0+dlo .. k0+n-1
#pragma unroll
        for (int d = 0; d < KBX; ++d)
            if (d >= dlo && d < n) {
                float* dst = ibuf + ((k0 + d) % KBX) * 6 * SARR + sr * SROW + sp * 4;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if constexpr (SV16) {
                        if constexpr (decltype(W16)::value) {
                            const unsigned w0 = __float_as_uint(sreg[d][i][0]), w1 = __float_as_uint(sreg[d][i][1]);
                            const float2 a = i < 2 ? unpack_unorm2(w0) : unpack_snorm2(w0), b = i < 2 ? unpack_unorm2(w1) : unpack_snorm2(w1);
                            const f32x4 v = {a.x, a.y, b.x, b.y};
                            *reinterpret_cast<f32x4*>(dst + i * SARR) = v;
                        } else if (BF && i < 2) {               // bf16 pairs -> fp32 (the value sits in the upper half)
                            const unsigned w0 = __float_as_uint(sreg[d][i][0]), w1 = __float_as_uint(sreg[d][i][1]);
                            const f32x4 v = {__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
                            *reinterpret_cast<f32x4*>(dst + (3 + i) * SARR) = v;
                        } else {
                            *reinterpret_cast<f32x4*>(dst + (3 + i) * SARR) = sreg[d][i];
                        }
                    } else {
                        *reinterpret_cast<f32x4*>(dst + (sarr0 + 2 * i) * SARR) = sreg[d][i];
                    }
                }
            }
    };
    auto svc_flush = [&](int k0, int k1) {            // gate gradients of steps k0 .. k1-1: obuf -> dgi (dr, dz, dn), dghn (dn * r)
        if (!svalid) return;
        for (int k = k0 < 0 ? 0 : k0; k < k1; ++k) {
            const size_t row = (size_t)sb * T + (T - 1 - k);
            const float* o = obuf + (k % OSL) * 4 * SARR + sr * SROW + sp * 4;
            // arrays sarr0 (dr / dz) and 2 + sarr0 (dn / dn*r)
            stnt2(rs_dgi, dgi_t, p.dgi + row * p.lddg + (sodd ? H : 0) + scol, ld4(o + (sodd ? SARR : 0)));
            if (sodd) stnt2(rs_dghn, dghn_t, p.dghn + row * p.lddghn + scol, ld4(o + 3 * SARR));
            else stnt2(rs_dgi, dgi_t, p.dgi + row * p.lddg + 2 * H + scol, ld4(o + 2 * SARR));
        }
    };
    // PK image (p.dgpk): steps (ka, ka+1), ka even, are rows t_even + 1, t_even (t_even = T-2-ka) of the utterance; physical row
    // t_even holds bf16hi(x[t_even]) | bf16hi(x[t_even+1]) << 16 per column, row t_even + 1 the residual (lo) pairs -- exactly the
    // (hi, lo) the GEMM's split4 would form (same v_cvt_pk_bf16_f32 roundings), so the contractions' bits do not change.
    auto svc_flush_pk = [&](int k0, int k1) {         // complete step pairs in [k0, k1): both even
        if (!svalid) return;
        for (int ka = k0 < 0 ? 0 : k0; ka + 1 < k1; ka += 2) {
            const size_t row = (size_t)sb * T + (T - 2 - ka);                    // the even row of the pair
            const float* oo = obuf + (ka % OSL) * 4 * SARR + sr * SROW + sp * 4;        // step ka   = row t_even + 1
            const float* oe = obuf + ((ka + 1) % OSL) * 4 * SARR + sr * SROW + sp * 4;  // step ka+1 = row t_even
#pragma unroll
            for (int q = 0; q < 2; ++q) {             // arrays sarr0 (dr / dz) and 2 + sarr0 (dn / dn*r)
                const int ao = (q ? 2 : 0) + (sodd ? 1 : 0);
                const f32x4 xe = ld4(oe + ao * SARR), xo = ld4(oo + ao * SARR);
                u32x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) { unsigned hh, ll; split_pair(xe[e], xo[e], hh, ll); h[e] = hh; l[e] = ll; }
                float* g = p.dgi + row * p.lddg + ao * H + scol;                 // 4H-wide rows [dr | dz | dn | dn*r]
                stnt2(rs_dgi, dgi_t, g, __builtin_bit_cast(f32x4, h));
                if constexpr (!BF) stnt2(rs_dgi, dgi_t, g + p.lddg, __builtin_bit_cast(f32x4, l));      // (PKH: the hi rows only)
            }
        }
    };
    if constexpr (BURST) {
        // the service waves' whole life (W16: this wave pair streams the 16-bit arrays)
        auto svc_life = [&](auto W16, auto NT) {
            // the service waves' whole life: same barrier sequence as the compute waves' loop below (two per step, one in the last)
            svc_issue(W16, NT, 0, KBX); svc_put(W16, 0, KBX);       // steps 0 .. KB-1 straight into the ring
            svc_issue(W16, NT, KBX, phi);                      // steps KB .. KB+phi-1: written at step phi-1, before the first dirty step (k = phi)
            __syncthreads();
            for (int k = 0; k < T; ++k) {
                const int jj = (k + KBX - phi) % KBX;                  // jj == 0: this tile's dirty step
                const int last = k - jj;                               // the dirty step this burst started with (< 0: the prologue)
                const bool trs = p.trace && blockIdx.x == 0 && st == 0 && k >= 100 && k < 104;
                if (trs) p.trace[32 + (k - 100) * 4 + 0] = (long long)__builtin_readcyclecounter();
                if (jj == 0) {
                    // everything the next KB steps need from HBM, and everything the last KB produced.  (The service waves run
                    // ahead of the compute waves -- nothing holds them after barrier #2 -- so the burst starts during the
                    // previous step's poll; holding it back until the compute waves reach the dirty step's gate phase
                    // measured the same launch time: the burst occupies the CU's memory pipeline for ~1.3 steps either way.)
                    svc_issue(W16, NT, k + KBX, KBX);
                    if (trs) p.trace[32 + (k - 100) * 4 + 1] = (long long)__builtin_readcyclecounter();
                    if (p.dgpk) svc_flush_pk(k - KBX - (phi & 1), k - (phi & 1));      // whole pairs: one step later for the tiles with an odd phase
                    else svc_flush(k - KBX, k);
                }
                if (trs) p.trace[32 + (k - 100) * 4 + 2] = (long long)__builtin_readcyclecounter();
                bar_lds();                           // #1
                if (trs) p.trace[32 + (k - 100) * 4 + 3] = (long long)__builtin_readcyclecounter();
                if constexpr (AG) {
                    // The all-gather step has ONE barrier, at its end, and the compute waves read ring slot (k+1) % KB right behind
                    // barrier(k): step s may be written only between barrier(s - KB) (its slot's previous occupant consumed) and
                    // barrier(s - 1).  For the burst requested at dirty step L = last (steps L+KB .. L+2KB-1) that is: the first
                    // KB-1 behind barrier(L + KB - 2), the last one behind barrier(L + KB - 1).  No race window at all.
                    if (jj == KBX - 2 && last >= 0) svc_put(W16, last + KBX, KBX - 1);
                    if (jj == KBX - 1) {
                        if (last >= 0) svc_put(W16, last + KBX, KBX, KBX - 1);
                        else svc_put(W16, KBX, phi);         // the prologue's steps KB .. KB+phi-1 (their slots were consumed by barrier(phi-1))
                    }
                    if (k == T - 1) break;
                    continue;
                }
                if (jj == KBX - 1) {
                    // last step of the burst: the ring slots of steps last .. k are consumed; the registers (requested at step
                    // `last', KB-1 steps ago) become steps last+KB .. k+KB
                    if (last >= 0) svc_put(W16, last + KBX, KBX);
                    else svc_put(W16, KBX, phi);
                }
                if (k == T - 1) break;
                if (!p.wflags) bar_lds();            // #2 (the compute waves' drain barrier; absent with per-wave flags)
            }
            const int jl2 = (T - 1 + KBX - phi) % KBX;
            if (p.dgpk) svc_flush_pk(T - 1 - jl2 - (phi & 1), T);      // T is even: the last pair is complete
            else svc_flush(T - 1 - jl2, T);           // the gate gradients since the last dirty step
        };
        if (svc) {
            const bool ntl = p.ntload != 0;             // uniform
            if constexpr (SV16) {
                if (sodd) { if (ntl) svc_life(std::true_type{}, std::true_type{}); else svc_life(std::true_type{}, std::false_type{}); }
                else { if (ntl) svc_life(std::false_type{}, std::true_type{}); else svc_life(std::false_type{}, std::false_type{}); }
            } else {
                if (ntl) svc_life(std::false_type{}, std::true_type{}); else svc_life(std::false_type{}, std::false_type{});
            }
            return;
        }
        __syncthreads();
    }
    // inter-layer dropout on the incoming dy: the Philox draw of step t-1 is made while step t waits for the other members
    // (it depends on nothing but the position), so it never sits on the step's critical path
    const bool masked = p.dy && p.drop_p > 0.f && valid;
    auto draw = [&](int t) {
        const size_t o = ((size_t)b * T + t) * p.lddy + col;
        const f32x4 m = dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
        return f2(half ? m[2] : m[0], half ? m[3] : m[1]);
    };
    float2 mk = masked ? draw(T - 1) : f2(1.f, 1.f);
    // debug stamps (DEP_TRACE=1, tools/trace_bwd.py): buffered in otherwise unused LDS, copied out after the sweep
    long long* trb = (p.trace && blockIdx.x == 0 && tid == 0) ? p.trace : nullptr;
    long long* trlb = reinterpret_cast<long long*>(smem + (AG ? AG_TRACE_F : BURST ? TRACE_F : 8192));
#define BSTAMP(slot) do { if (trb && t <= 199 && t > 195) trlb[(199 - t) * 8 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

    if (trb) trlb[7] = (long long)__builtin_readcyclecounter();
    unsigned* trall = reinterpret_cast<unsigned*>(smem) + p.trall_off;      // DEP_TRACE: low 32 bits of the tick counter at the top of every step
    if constexpr (AG) {
        // ---- all-gather sweep (see AG_MEMBER_BYTES above).  Exchange buffer: [parity][tile][member][gate][plane][64 lanes][16 B]; this
        // thread's pair (units ul, ul+1 of utterance j) is word `pw' of each of its member's six 1 KB blocks -- B-fragment order: lane
        // (k-group ul / 8, utterance j), word (ul % 8) / 2 -- and a wave's 64 words are 256 contiguous bytes.
        float* red = smem;                                  // [step parity][K quarter = wave][own tile][64 lanes][4]
        const unsigned pw = (unsigned)(half + 2 * ((lp >> 4) & 1) + 4 * j + 64 * (lp >> 5) + 128 * jl);
        const unsigned par_bytes = (unsigned)p.nbtp * NC * AG_MEMBER_BYTES;
        const unsigned pub0 = (unsigned)(bt * NC + c) * AG_MEMBER_BYTES + pw * 4;
        const unsigned ld0 = (unsigned)(bt * NC + 2 * w) * AG_MEMBER_BYTES + lane * 16;     // this wave's 12 blocks are contiguous: members 2w, 2w+1
        unsigned* srcflags = tflags + 8 * w;                // the eight per-wave flags of source members 2w, 2w+1
        const int rdo = (jl * 64 + (lp >> 4) * 16 + j) * 4 + 2 * half;      // own pair inside a K quarter's two accumulator tiles
        for (int t = T - 1; t >= 0; --t) {
            const int k = T - 1 - t;
            BSTAMP(0);
            if (trb && k < 360) trall[k] = (unsigned)__builtin_readcyclecounter();
            const float* ib = ibuf + (k % KBX) * 6 * SARR + j * SROW + ul;
            const float2 r = ld2(ib), z = ld2(ib + SARR), n = ld2(ib + 2 * SARR), hn = ld2(ib + 3 * SARR), hp = ld2(ib + 4 * SARR), dyi = ld2(ib + 5 * SARR);
            const float2 d = f2(dhrec.x + dpl.x + dyi.x * mk.x, dhrec.y + dpl.y + dyi.y * mk.y);
            float2 dn, dz, dr, dnr, dzt;
            dn.x = d.x * (1.0f - z.x) * (1.0f - n.x * n.x); dn.y = d.y * (1.0f - z.y) * (1.0f - n.y * n.y);
            dz.x = d.x * (hp.x - n.x) * z.x * (1.0f - z.x); dz.y = d.y * (hp.y - n.y) * z.y * (1.0f - z.y);
            dr.x = dn.x * hn.x * r.x * (1.0f - r.x); dr.y = dn.y * hn.y * r.y * (1.0f - r.y);
            dnr.x = dn.x * r.x; dnr.y = dn.y * r.y;
            dzt.x = d.x * z.x; dzt.y = d.y * z.y;
            const unsigned epoch = (unsigned)(k + 1);
            if (t > 0) {      // publish first: the (hi, lo) pair words of dr, dz, dn*r -- what every member's MFMAs read
                unsigned h0, l0, h1, l1, h2, l2;
                split_pair(dr.x, dr.y, h0, l0); split_pair(dz.x, dz.y, h1, l1); split_pair(dnr.x, dnr.y, h2, l2);
                const unsigned po = (unsigned)(t & 1) * par_bytes + pub0;
                const unsigned wds[6] = {h0, l0, h1, l1, h2, l2};
                if (fast) {           // same-XCD clusters: plain stores (that XCD's L2 is the coherence point)
#pragma unroll
                    for (int e = 0; e < 6; ++e) __builtin_amdgcn_raw_buffer_store_b32(wds[e], rsrc, po + e * 1024, 0, 0);
                } else {              // write-through
#pragma unroll
                    for (int e = 0; e < 6; ++e) __builtin_amdgcn_raw_buffer_store_b32(wds[e], rsrc, po + e * 1024, 0, 16);
                }
            }
            {
                float* ob = obuf + (k % OSL) * 4 * SARR + j * SROW + ul;
                st2(ob, dr); st2(ob + SARR, dz); st2(ob + 2 * SARR, dn); st2(ob + 3 * SARR, dnr);
            }
            dbr.x += dr.x; dbr.y += dr.y; dbz.x += dz.x; dbz.y += dz.y; dbn.x += dn.x; dbn.y += dn.y; dbh.x += dnr.x; dbh.y += dnr.y;
            BSTAMP(1);
            if (t == 0) { bar_lds(); break; }                // (the service waves' final flush reads obuf behind this barrier)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's six stores are acknowledged
            BSTAMP(2);
            if (lane == 0) { if (fast) st_local(myflag, epoch); else st_agent(myflag, epoch); }
            BSTAMP(3);
            // One vector poll of the eight per-wave flags of both source members, then all twelve requests; the Philox draw of the next
            // step's dropout mask (layer 0 only, ~150 VALU instructions) goes behind the requests, into the loads' shadow.  Measured and
            // dropped (profiles/r05_s2_allgather_poll_variants.txt): member-by-member poll + requests (the second poll returns behind the
            // first six loads: +2 %), polling through the scalar path (s_load ... glc: the flags arrive thousands of ticks late, 2x slower).
            const unsigned lo_ = (unsigned)(t & 1) * par_bytes + ld0;
            u32x4 gfr[6][2];
            if (!wait_flags(srcflags, 8, epoch, p.status, 3)) return;
            BSTAMP(4);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    gfr[ks][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lo_ + (unsigned)(ks * 2 + pl) * 1024, 0, 16 /* sc1: served by L2 */);
            __builtin_amdgcn_sched_barrier(0);               // all twelve requests first (hipcc otherwise sinks each load next to its MFMAs: twelve round trips in a row)
            if (masked) mk = draw(t - 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[2] = {zero4(), zero4()};
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const bf16x8 gh = __builtin_bit_cast(bf16x8, gfr[ks][0]), gl = __builtin_bit_cast(bf16x8, gfr[ks][1]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, wa[i][ks][0]), wl = __builtin_bit_cast(bf16x8, wa[i][ks][1]);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gl, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, gh, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gh, acc[i], 0, 0, 0);
                }
            }
            float* rw = red + ((k & 1) * 8 + w * 2) * 256 + lane * 4;
            *reinterpret_cast<f32x4*>(rw) = acc[0]; *reinterpret_cast<f32x4*>(rw + 256) = acc[1];
            BSTAMP(5);
            bar_lds();
            const float* rr = red + (k & 1) * 2048 + rdo;
            float2 s = f2(0.f, 0.f);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) { const float2 v = ld2(rr + q4 * 512); s.x += v.x; s.y += v.y; }      // fixed order: deterministic
            dhrec = f2(dzt.x + s.x, dzt.y + s.y);
            BSTAMP(6);
        }
    } else
    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)b * T + t;
        BSTAMP(0);
        if (trb && T - 1 - t < 360) trall[T - 1 - t] = (unsigned)__builtin_readcyclecounter();
        if constexpr (BURST) {
            const float* ib = ibuf + ((T - 1 - t) % KBX) * 6 * SARR + j * SROW + ul;
            cur.r = ld2(ib); cur.z = ld2(ib + SARR); cur.n = ld2(ib + 2 * SARR); cur.hn = ld2(ib + 3 * SARR);
            cur.hp = ld2(ib + 4 * SARR); cur.dy = ld2(ib + 5 * SARR);
        }
        const float2 dyv = f2(cur.dy.x * mk.x, cur.dy.y * mk.y);
        const float2 r = cur.r, z = cur.z, n = cur.n, hn = cur.hn, hp = cur.hp;
        const float2 d = f2(dhrec.x + dpl.x + dyv.x, dhrec.y + dpl.y + dyv.y);
        float2 dn, dz, dr, dnr, dzt;
        dn.x = d.x * (1.0f - z.x) * (1.0f - n.x * n.x); dn.y = d.y * (1.0f - z.y) * (1.0f - n.y * n.y);
        dz.x = d.x * (hp.x - n.x) * z.x * (1.0f - z.x); dz.y = d.y * (hp.y - n.y) * z.y * (1.0f - z.y);
        dr.x = dn.x * hn.x * r.x * (1.0f - r.x); dr.y = dn.y * hn.y * r.y * (1.0f - r.y);
        dnr.x = dn.x * r.x; dnr.y = dn.y * r.y;
        dzt.x = d.x * z.x; dzt.y = d.y * z.y;
        if constexpr (SPLIT) {                        // one (hi, lo) bf16 pair word per gate: units ul, ul+1
            unsigned h0, l0, h1, l1, h2, l2;
            split_pair(dr.x, dr.y, h0, l0); split_pair(dz.x, dz.y, h1, l1); split_pair(dnr.x, dnr.y, h2, l2);
            const int o = j * LDGB + ul;
            *reinterpret_cast<unsigned*>(dg_hi + o) = h0; *reinterpret_cast<unsigned*>(dg_lo + o) = l0;
            *reinterpret_cast<unsigned*>(dg_hi + o + 32) = h1; *reinterpret_cast<unsigned*>(dg_lo + o + 32) = l1;
            *reinterpret_cast<unsigned*>(dg_hi + o + 64) = h2; *reinterpret_cast<unsigned*>(dg_lo + o + 64) = l2;
        } else {
            st2(dgs + j * LDG + ul, dr); st2(dgs + j * LDG + 32 + ul, dz); st2(dgs + j * LDG + 64 + ul, dnr);
        }
        if constexpr (BURST) {
            float* ob = obuf + ((T - 1 - t) % OSL) * 4 * SARR + j * SROW + ul;
            st2(ob, dr); st2(ob + SARR, dz); st2(ob + 2 * SARR, dn); st2(ob + 3 * SARR, dnr);
        } else if (valid) {
            float* g = p.dgi + row * p.lddg;
            st2(g + col, dr); st2(g + H + col, dz); st2(g + 2 * H + col, dn);
            st2(p.dghn + row * p.lddghn + col, dnr);
        }
        dbr.x += dr.x; dbr.y += dr.y; dbz.x += dz.x; dbz.y += dz.y; dbn.x += dn.x; dbn.y += dn.y; dbh.x += dnr.x; dbh.y += dnr.y;
        bar_lds();                                   // LDS only: the dgi/dghn stores above stay in flight
        BSTAMP(1);
        if (t == 0) break;
        if constexpr (!BURST) load_step(t - 1, nxt); // independent of the recurrence: in flight under the MFMAs
        f32x4 acc[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) acc[i] = zero4();
        if constexpr (SPLIT) {
            const int go = ml * LDGB + mq * 8;        // lane (utterance ml, k-group mq): 8 consecutive k of each k-step
            bf16x8 gh[3], gl[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                gh[ks] = *reinterpret_cast<const bf16x8*>(dg_hi + go + ks * 32);
                gl[ks] = *reinterpret_cast<const bf16x8*>(dg_lo + go + ks * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[i][ks][0]), wl = __builtin_bit_cast(bf16x8, wq[i][ks][1]);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gl[ks], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, gh[ks], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gh[ks], acc[i], 0, 0, 0);
                }
        } else {
            const float* drow = dgs + ml * LDG + mq * 4;
            f32x4 hv[KCB];
#pragma unroll
            for (int k = 0; k < KCB; ++k) hv[k] = ld4(drow + k * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < KCB; ++k) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < NTW; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[i][k][e], hv[k][e], acc[i], 0, 0, 0);
            }
        }
        BSTAMP(2);
        // publish: payload[parity][tile][src c][out tile][lane][4]
        const unsigned epoch = (unsigned)(T - t);
        const size_t pbase = (size_t)(t & 1) * pstride + tile_base;
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const size_t fo = pbase + ((size_t)(c * NTT + w * NTW + i) * 64 + lane) * 4;
            u32x4 v;
            v.x = __float_as_uint(acc[i][0]); v.y = __float_as_uint(acc[i][1]);
            v.z = __float_as_uint(acc[i][2]); v.w = __float_as_uint(acc[i][3]);
            if (fast) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (unsigned)(fo * 4), 0, 0 /* plain: stays in this XCD's L2 */);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (unsigned)(fo * 4), 0, 16 /* sc1: write-through */);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
        BSTAMP(3);
        if (wf) {
            // per-wave flags: this wave's partials are acknowledged -> say so; nobody waits for the sibling waves here (their
            // flags are among the 4 NC words every wave polls below, which also orders this step's LDS reads before the next
            // step's LDS writes: a sibling raises its flag only after its MFMAs have read the gate-gradient planes)
            if (lane == 0) { if (fast) st_local(myflag, epoch); else st_agent(myflag, epoch); }
        } else {
            __builtin_amdgcn_s_barrier();            // every wave drained its payload stores (vmcnt(0) above)
            if (tid == 0) { if (fast) st_local(myflag, epoch); else st_agent(myflag, epoch); }
        }
        if (masked) mk = draw(t - 1);                // next step's mask, in the shadow of the wait below
        BSTAMP(4);
        // wait for every member's flag (one wave polls, relaxed; flags are monotonic)
        // every wave polls the flags itself (no verdict-broadcast barrier); a wave that gives up leaves, the hardware
        // barrier only counts live waves and the others give up too (status word raised)
        if (!wait_flags(tflags, nflags, epoch, p.status, 3)) return;
        BSTAMP(5);
        // gather this thread's two columns from the NC partials, sum in member order
        float2 s = f2(0.f, 0.f);
        const float* src = p.payload + pbase + ((size_t)(2 * c + jl) * 64 + lp) * 4 + 2 * half;
        constexpr int NCM = 2 * NTW;                  // members of a cluster = H / 32
        float2 part[NCM];
#pragma unroll
        for (int m = 0; m < NCM; ++m) part[m] = ld2_agent(src + (size_t)m * NTT * 256);
#pragma unroll
        for (int m = 0; m < NCM; ++m) { s.x += part[m].x; s.y += part[m].y; }
        dhrec = f2(dzt.x + s.x, dzt.y + s.y);
        if constexpr (!BURST) cur = nxt;
        BSTAMP(6);
    }
    if (trb) {
        trlb[15] = (long long)__builtin_readcyclecounter();
        for (int i = 0; i < 32; ++i) trb[i] = trlb[i];
        unsigned* o = reinterpret_cast<unsigned*>(trb + 64);
        for (int i = 0; i < T && i < 360; ++i) o[i] = trall[i];
    }
    // bias-gradient partials dbpart[bt][4][H]: sum over the 16 utterance rows = lanes that differ in bits 1..4
    float2 a[4] = {dbr, dbz, dbn, dbh};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int m = 2; m <= 16; m <<= 1) { a[k].x += __shfl_xor(a[k].x, m, 64); a[k].y += __shfl_xor(a[k].y, m, 64); }
    if (j == 0) {
        float* o = p.dbpart + (size_t)(p.b0 / BT + bt) * 4 * H;
        st2(o + col, a[0]); st2(o + H + col, a[1]); st2(o + 2 * H + col, a[2]); st2(o + 3 * H + col, a[3]);
    }
}

// =============================================================================== GRU forward, flag-published
// Same member/wave roles as gru_fwd_cluster (rnn_cluster.hip): wave w = (hidden tile jl = w>>1, K half kh = w&1).
// h_t travels as plain fp32 rows payload[parity][tile][16][H]: each lane stores its two values with one 8-byte
// sc1 store, the workgroup drains + barriers, one lane raises the member's flag; after the NC flags are seen the
// whole 16xH block is read once with 16-byte sc1 loads into LDS (no per-value polling traffic).
struct F2 {
    int B, T, H, nbtp, b0;      // b0: first utterance of this launch's batch chunk (B is the whole batch)
    const f32x4* wp; const float* b_hh;
    const float* gi; int ldgi;
    float* y; int ldy;
    float* ydrop; float drop_p, drop_scale; uint64_t seed; uint32_t site;
    float* pooled; float pool_scale;
    float* h_n;
    float* sv0; float* sv1; float* sv2; float* sv3;
    unsigned* status; unsigned* flags; unsigned* hello; float* payload; unsigned payload_bytes;
    int nofast;              // DEP_CLUSTER_NOFAST=1: always use the write-through (placement-agnostic) stores
    long long* trace;        // debug: s_memtime stamps of workgroup 0 (DEP_TRACE=1), else nullptr
    const unsigned* only_if; // run only if this word is set (fallback behind an exclusive forward kernel), or nullptr
    int sv16;                // saved gates r, z, n written as 16-bit fixed point
};

template <int KCH, bool SPLIT>      // SPLIT: 3-term bf16 split of the recurrent product, as in gru_fwd_cluster16
__global__ __launch_bounds__(CT) void gru_fwd_cluster_r1(F2 p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, LDH = H + LPAD, KC = H / 16, NC = H / 32;
    const int c = blockIdx.x / p.nbtp, bt = blockIdx.x % p.nbtp;
    if (p.b0 + bt * BT >= p.B) return;
    if (ld_agent(p.status) != 0) return;           // an earlier sweep of this step gave up: the status word is sticky until the next dep_rnn_forward
    if (p.only_if && ld_agent(const_cast<unsigned*>(p.only_if)) == 0) return;      // the exclusive kernel in front of us did the work
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4, jl = w >> 1, kh = w & 1;
    const int jt = c * 2 + jl;
    const int b = p.b0 + bt * BT + j;
    const bool valid = b < p.B;
    const int LDHB = H + 8;                           // bf16 elements per row of a split plane
    float* hs = smem;                                 // [16][LDH] fp32, or (SPLIT) two bf16 planes [16][LDHB]
    const int hs_floats = SPLIT ? BT * LDHB : BT * LDH;
    unsigned short* hs_hi = reinterpret_cast<unsigned short*>(smem);
    unsigned short* hs_lo = hs_hi + BT * LDHB;
    float* red = smem + hs_floats;                    // [4][3][64][4]
    for (int i = tid; i < hs_floats; i += CT) hs[i] = 0.f;

    constexpr int KS2 = KCH / 2;                      // 32-wide k-steps per wave (SPLIT)
    f32x4 wr[SPLIT ? 1 : 3][SPLIT ? 1 : KCH];
    u32x4 wq[SPLIT ? 3 : 1][SPLIT ? KS2 : 1][2];      // [gate][k-step][hi, lo]
    if constexpr (SPLIT) {
        const u32x4* wpq = reinterpret_cast<const u32x4*>(p.wp);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    wq[g][ks][pl] = wpq[(size_t)((((jt * 3 + g) * 2 + kh) * KS2 + ks) * 2 + pl) * 64 + lane];
    } else {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k = 0; k < KCH; ++k)
                wr[g][k] = p.wp[(size_t)((jt * 3 + g) * KC + kh * KCH + k) * 64 + lane];
    }
    const int col = jt * 16 + q * 4 + 2 * kh;
    float2 bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = ld2(p.b_hh + g * H + col);
    float2 hprev = f2(0.f, 0.f), pool = f2(0.f, 0.f);
    const size_t pstride = (size_t)p.nbtp * BT * H;
    const size_t tile_base = (size_t)bt * BT * H;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.payload, 0, p.payload_bytes, 0x00020000);
    unsigned* myflag = p.flags + bt * NC + c;
    unsigned* tflags = p.flags + bt * NC;
    const int hshift = __ffs(H) - 1;
    const int sx = p.nofast ? 0 : cluster_same_xcd(p.hello + bt * NC, NC, c, p.status);
    if (sx < 0) return;
    const bool fast = sx == 1;
    float2 gin[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
        gin[g] = valid ? ld2(p.gi + (size_t)b * T * p.ldgi + g * H + col) : f2(0.f, 0.f);
    __syncthreads();

    long long* tr = (p.trace && blockIdx.x == 0 && tid == 0) ? p.trace : nullptr;
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)b * T + t;
        DEP_STAMP(0);
        float2 gi[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) gi[g] = gin[g];
        if (valid && t + 1 < T) {
#pragma unroll
            for (int g = 0; g < 3; ++g) gin[g] = ld2(p.gi + (row + 1) * p.ldgi + g * H + col);
        }
        f32x4 acc[3] = {zero4(), zero4(), zero4()};
        if constexpr (SPLIT) {
            const int ho = j * LDHB + kh * KCH * 16 + q * 8;
            bf16x8 hh[KS2], hl[KS2];
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                hh[ks] = *reinterpret_cast<const bf16x8*>(hs_hi + ho + ks * 32);
                hl[ks] = *reinterpret_cast<const bf16x8*>(hs_lo + ho + ks * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[g][ks][0]), wl = __builtin_bit_cast(bf16x8, wq[g][ks][1]);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hl[ks], acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, hh[ks], acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hh[ks], acc[g], 0, 0, 0);
                }
        } else {
            const float* hrow = hs + j * LDH + kh * KCH * 16 + q * 4;
            f32x4 hv[KCH];                             // all B fragments first: one LDS latency, not KCH of them
#pragma unroll
            for (int k = 0; k < KCH; ++k) hv[k] = ld4(hrow + k * 16);
            __builtin_amdgcn_sched_barrier(0);         // keep the loads grouped: hipcc otherwise sinks each next to its MFMAs
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[g][k][e], hv[k][e], acc[g], 0, 0, 0);
            }
        }
        DEP_STAMP(1);
#pragma unroll
        for (int g = 0; g < 3; ++g) *reinterpret_cast<f32x4*>(red + ((w * 3 + g) * 64 + lane) * 4) = acc[g];
        bar_lds();
        DEP_STAMP(2);
        float2 tot[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float2 pv = ld2(red + (((w ^ 1) * 3 + g) * 64 + lane) * 4 + 2 * kh);
            tot[g].x = (kh ? acc[g][2] : acc[g][0]) + pv.x;
            tot[g].y = (kh ? acc[g][3] : acc[g][1]) + pv.y;
        }
        float2 r, z, hn, n, h;
        r.x = fast_sigmoid(gi[0].x + tot[0].x + bh[0].x); r.y = fast_sigmoid(gi[0].y + tot[0].y + bh[0].y);
        z.x = fast_sigmoid(gi[1].x + tot[1].x + bh[1].x); z.y = fast_sigmoid(gi[1].y + tot[1].y + bh[1].y);
        hn.x = tot[2].x + bh[2].x; hn.y = tot[2].y + bh[2].y;
        n.x = fast_tanh(gi[2].x + r.x * hn.x); n.y = fast_tanh(gi[2].y + r.y * hn.y);
        h.x = (1.0f - z.x) * n.x + z.x * hprev.x; h.y = (1.0f - z.y) * n.y + z.y * hprev.y;
        hprev = h; pool.x += h.x; pool.y += h.y;
        const unsigned epoch = (unsigned)t + 1u;
        const size_t pbase = (size_t)(t & 1) * pstride + tile_base;
        const bool more = t + 1 < T;
        DEP_STAMP(3);
        if (more) {       // publish first: it is on the critical path of the other members
            const u64 bits = SPLIT ? ((u64)split_word(h.x) | ((u64)split_word(h.y) << 32))
                                   : ((u64)__float_as_uint(h.x) | ((u64)__float_as_uint(h.y) << 32));
            if (fast) __hip_atomic_store((gu64*)(p.payload + pbase + (size_t)j * H + col), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store((gu64*)(p.payload + pbase + (size_t)j * H + col), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DEP_STAMP(4);
            __builtin_amdgcn_s_barrier();
            if (tid == 0) { if (fast) st_local(myflag, epoch); else st_agent(myflag, epoch); }
        }
        if (valid) {
            const size_t o = row * p.ldy + col;
            st2(p.y + o, h);
            if (p.ydrop) {
                const f32x4 m = dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
                st2(p.ydrop + o, f2(h.x * (kh ? m[2] : m[0]), h.y * (kh ? m[3] : m[1])));
            }
            if (p.sv0) {
                const size_t so = row * H + col;
                if (p.sv16) {
                    *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(p.sv0) + so) = pack_unorm2(r.x, r.y);
                    *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(p.sv1) + so) = pack_unorm2(z.x, z.y);
                    *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(p.sv2) + so) = pack_snorm2(n.x, n.y);
                } else { st2(p.sv0 + so, r); st2(p.sv1 + so, z); st2(p.sv2 + so, n); }
                st2(p.sv3 + so, hn);
            }
        }
        if (more) {
            if (!wait_flags(tflags, NC, epoch, p.status, 4)) return;      // every wave polls
            DEP_STAMP(5);
            DEP_STAMP(6);
            constexpr int PER = KCH / 2;              // 16-byte pieces per thread = 16*H/4/256
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i4 = (tid + CT * k) * 4;    // float index inside the 16xH block
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)((pbase + i4) * 4), 0, 16 /* sc1 */);
                if constexpr (SPLIT) {
                    const int o = (i4 >> hshift) * LDHB + (i4 & (H - 1));
                    uint2 hi2, lo2;
                    hi2.x = (v.x >> 16) | (v.y & 0xffff0000u); hi2.y = (v.z >> 16) | (v.w & 0xffff0000u);
                    lo2.x = (v.x & 0xffffu) | (v.y << 16);      lo2.y = (v.z & 0xffffu) | (v.w << 16);
                    *reinterpret_cast<uint2*>(hs_hi + o) = hi2; *reinterpret_cast<uint2*>(hs_lo + o) = lo2;
                } else {
                    f32x4 f;
                    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
                    *reinterpret_cast<f32x4*>(hs + (i4 >> hshift) * LDH + (i4 & (H - 1))) = f;
                }
            }
            bar_lds();
            DEP_STAMP(7);
        }
    }
    if (valid) {
        if (p.pooled) st2(p.pooled + (size_t)b * H + col, f2(pool.x * p.pool_scale, pool.y * p.pool_scale));
        if (p.h_n) st2(p.h_n + (size_t)b * H + col, hprev);
    }
}

// split-precision backward image (gru_bwd_cluster_r1<., true>): 16-byte piece
//   [(((c*(H/16) + jt)*3 + ks)*2 + plane)*64 + lane] = bf16 plane (0 hi, 1 lo) of
//   W[(ks*H + 32c + 8(lane>>4) + e) * H + jt*16 + (lane&15)],  e = 0..7     (k-step ks = gate, 32 units of member c)
__global__ void pack_cluster_bwd_split_kernel(const float* __restrict__ W, u32x4* __restrict__ out, int H) {
    const long n = (long)(H / 32) * (H / 16) * 3 * 64;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int lane = idx & 63; long r = idx >> 6;
    const int ks = r % 3; r /= 3;
    const int jt = r % (H / 16); const int c = r / (H / 16);
    const float* src = W + (size_t)(ks * H + 32 * c + 8 * (lane >> 4)) * H + jt * 16 + (lane & 15);
    u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned h, l;
        split_pair(src[(size_t)(2 * e) * H], src[(size_t)(2 * e + 1) * H], h, l);
        hi[e] = h; lo[e] = l;
    }
    out[(idx - lane) * 2 + lane] = hi;
    out[(idx - lane) * 2 + 64 + lane] = lo;
}

// split-precision forward image of the 32-unit-member kernel (gru_fwd_cluster_r1<., true>): 16-byte piece
//   [((((jt*3 + g)*2 + kh)*KS2 + ks)*2 + plane)*64 + lane] = bf16 plane of W[(g*H + jt*16 + (lane&15))*H + kh*(H/2) + 32ks + 8(lane>>4) + 0..7]
__global__ void pack_cluster_fwd_split_kernel(const float* __restrict__ W, u32x4* __restrict__ out, int H) {
    const int KS2 = H / 64;
    const long n = (long)(H / 16) * 3 * 2 * KS2 * 64;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int lane = idx & 63; long r = idx >> 6;
    const int ks = r % KS2; r /= KS2;
    const int kh = r % 2; r /= 2;
    const int g = r % 3; const int jt = r / 3;
    const float* src = W + (size_t)(g * H + jt * 16 + (lane & 15)) * H + kh * (H / 2) + 32 * ks + 8 * (lane >> 4);
    u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned h, l; split_pair(src[2 * e], src[2 * e + 1], h, l); hi[e] = h; lo[e] = l; }
    out[(idx - lane) * 2 + lane] = hi;
    out[(idx - lane) * 2 + 64 + lane] = lo;
}

// all split-precision images of a step in ONE launch (blockIdx.y = job): five ~4.5 us launches per training step otherwise
struct PackJobs { const float* src[8]; u32x4* dst[8]; int bwd[8]; int n; };
__global__ void pack_cluster_split_multi_kernel(PackJobs j, int H) {
    const int k = blockIdx.y;
    if (k >= j.n) return;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const float* W = j.src[k]; u32x4* out = j.dst[k];
    if (j.bwd[k]) {
        const long n = (long)(H / 32) * (H / 16) * 3 * 64;
        if (idx >= n) return;
        const int lane = idx & 63; long r = idx >> 6;
        const int ks = r % 3; r /= 3;
        const int jt = r % (H / 16); const int c = r / (H / 16);
        const float* src = W + (size_t)(ks * H + 32 * c + 8 * (lane >> 4)) * H + jt * 16 + (lane & 15);
        u32x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { unsigned h, l; split_pair(src[(size_t)(2 * e) * H], src[(size_t)(2 * e + 1) * H], h, l); hi[e] = h; lo[e] = l; }
        out[(idx - lane) * 2 + lane] = hi;
        out[(idx - lane) * 2 + 64 + lane] = lo;
    } else {
        const int KS2 = H / 64;
        const long n = (long)(H / 16) * 3 * 2 * KS2 * 64;
        if (idx >= n) return;
        const int lane = idx & 63; long r = idx >> 6;
        const int ks = r % KS2; r /= KS2;
        const int kh = r % 2; r /= 2;
        const int g = r % 3; const int jt = r / 3;
        const float* src = W + (size_t)(g * H + jt * 16 + (lane & 15)) * H + kh * (H / 2) + 32 * ks + 8 * (lane >> 4);
        u32x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { unsigned h, l; split_pair(src[2 * e], src[2 * e + 1], h, l); hi[e] = h; lo[e] = l; }
        out[(idx - lane) * 2 + lane] = hi;
        out[(idx - lane) * 2 + 64 + lane] = lo;
    }
}

}  // namespace

// (H x 3H recurrent / input weight) -> forward (bwd[k] = 0) or backward (1) split image, up to 8 jobs in one launch
int dep_pack_cluster_split_multi(int n, const float* const* src, float* const* dst, const int* bwd, int H, hipStream_t s) {
    DEP_CHECK_ARG(n > 0 && n <= 8 && src && dst && bwd);
    PackJobs j{};
    j.n = n;
    for (int k = 0; k < n; ++k) { DEP_CHECK_ARG(src[k] && dst[k]); j.src[k] = src[k]; j.dst[k] = (u32x4*)dst[k]; j.bwd[k] = bwd[k]; }
    const long nf = (long)(H / 16) * 3 * 2 * (H / 64) * 64, nb = (long)(H / 32) * (H / 16) * 3 * 64;
    DEP_LAUNCH(pack_cluster_split_multi_kernel, dim3(dep_cdiv(nf > nb ? nf : nb, 256), n), dim3(256), 0, s, j, H);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

int dep_pack_cluster_fwd_split(const float* w_hh, float* out, int H, hipStream_t s) {
    const long n = (long)(H / 16) * 3 * 2 * (H / 64) * 64;
    DEP_LAUNCH(pack_cluster_fwd_split_kernel, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, w_hh, (u32x4*)out, H);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

int dep_pack_cluster_bwd_split(const float* w_hh, float* out, int H, hipStream_t s) {
    const long n = (long)(H / 32) * (H / 16) * 3 * 64;
    DEP_LAUNCH(pack_cluster_bwd_split_kernel, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, w_hh, (u32x4*)out, H);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

int dep_launch_cluster_fwd(const dep_sweep_args& a, void* xbuf, size_t xbuf_bytes) {
    DEP_CHECK_ARG(dep_cluster_ok(a.cell, a.H, a.B, a.dirs) && xbuf && xbuf_bytes >= dep_cluster_xbuf_bytes(a.cell, a.H, a.B, a.dirs));
    const int NC = a.H / 32, CH = dep_cluster_chunk(NC, 1, 256);      // one workgroup per CU: 256 / NC tiles per launch, larger batches in chunks
    const int nbtp_max = (dep_cdiv(a.B < CH ? a.B : CH, BT) + 7) / 8 * 8;
    F2 p{};
    p.B = a.B; p.T = a.T; p.H = a.H;
    p.wp = (const f32x4*)a.wp[0]; p.b_hh = a.b_hh[0];
    p.gi = a.gi; p.ldgi = 3 * a.H; p.y = a.y; p.ldy = a.ldy;
    p.ydrop = (a.drop_p > 0.f) ? a.ydrop : nullptr;
    p.drop_p = a.drop_p; p.drop_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f; p.seed = a.seed; p.site = a.site;
    p.pooled = a.pooled; p.pool_scale = a.pool_scale; p.h_n = a.h_n;
    p.sv0 = a.training ? a.sv0 : nullptr; p.sv1 = a.sv1; p.sv2 = a.sv2; p.sv3 = a.sv3;
    const size_t pay = (size_t)2 * nbtp_max * BT * a.H * sizeof(float);
    DEP_CHECK_ARG(PAYLOAD_OFF + pay <= xbuf_bytes && (size_t)nbtp_max * NC <= 256);
    p.status = (unsigned*)xbuf; p.flags = (unsigned*)(hdr_base(xbuf, a.hdr_slot) + FLAG_OFF); p.hello = (unsigned*)(hdr_base(xbuf, a.hdr_slot) + HELLO_OFF);
    p.nofast = nofast_env();
    p.payload = (float*)((char*)xbuf + PAYLOAD_OFF); p.payload_bytes = (unsigned)pay;
    p.trace = trace_env() ? (long long*)(hdr_base(xbuf, a.hdr_slot) + TRACE_OFF) : nullptr;
    p.only_if = a.only_if; p.sv16 = a.training ? a.sv16 : 0;
    DepProfScope prof(DEP_PROF_GRU_FWD, a.stream, a.only_if == nullptr);      // a conditional fallback launch is not a sweep of the step
    // Ask for more than half of the CU's 160 KiB LDS: the dispatcher can then never co-locate two members on one
    // CU (they would share the four matrix pipes and stretch every step of BOTH clusters).
    const size_t lds = EXCLUSIVE_LDS;
    static bool attr_f = false;
    if (!attr_f) {
#define DEP_FWD_ATTR(K) (void)hipFuncSetAttribute((const void*)gru_fwd_cluster_r1<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                        (void)hipFuncSetAttribute((const void*)gru_fwd_cluster_r1<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
        DEP_FWD_ATTR(2); DEP_FWD_ATTR(4); DEP_FWD_ATTR(8); DEP_FWD_ATTR(16);
#undef DEP_FWD_ATTR
        attr_f = true;
    }
    for (int b0 = 0; b0 < a.B; b0 += CH) {
        const int cb = a.B - b0 < CH ? a.B - b0 : CH;
        p.b0 = b0; p.nbtp = (dep_cdiv(cb, BT) + 7) / 8 * 8;
        // flags / hello words only: the status word is sticky over every sweep of a step (cleared by dep_rnn_forward)
        { const int rc_h = hdr_prepare(xbuf, a.hdr_slot, a.hdr_clean && b0 == 0, a.stream); if (rc_h) return rc_h; }
        dim3 grid(NC * p.nbtp);
#define DEP_FWD_LAUNCH(K)                                                                                                 \
        do { if (a.split) DEP_LAUNCH((gru_fwd_cluster_r1<K, true>), grid, dim3(CT), lds, a.stream, p);            \
             else DEP_LAUNCH((gru_fwd_cluster_r1<K, false>), grid, dim3(CT), lds, a.stream, p); } while (0)
        switch (a.H) {                                // KCH = H / 32
            case 64: DEP_FWD_LAUNCH(2); break;
            case 128: DEP_FWD_LAUNCH(4); break;
            case 256: DEP_FWD_LAUNCH(8); break;
            default: DEP_FWD_LAUNCH(16); break;       // 512
        }
#undef DEP_FWD_LAUNCH
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}

// May dep_launch_cluster_bwd write the gate gradients as the PK image (dep_sweep_bwd_args.dg_pk)?  Needs the burst-stream kernel
// (its service waves' flush forms the pairs): H <= 256, bursts not switched off, not the two-per-CU placement experiment.
bool dep_cluster_bwd_pk_ok(int H, int T) { return H <= 256 && T % 2 == 0; }

int dep_launch_cluster_bwd(const dep_sweep_bwd_args& a, void* xbuf, size_t xbuf_bytes) {
    DEP_CHECK_ARG(dep_cluster_ok(a.cell, a.H, a.B, a.dirs) && xbuf && xbuf_bytes >= dep_cluster_xbuf_bytes(a.cell, a.H, a.B, a.dirs));
    const int NC = a.H / 32, CH = dep_cluster_chunk(NC, 1, 256), nbt = dep_cdiv(a.B, BT);
    const int nbtp_max = (dep_cdiv(a.B < CH ? a.B : CH, BT) + 7) / 8 * 8;
    P2 p{};
    p.B = a.B; p.T = a.T; p.H = a.H;
    p.wp = (const f32x4*)a.wpT[0];
    p.y = a.y; p.ldy = a.ldy; p.dy = a.dy; p.lddy = a.lddy;
    p.drop_p = a.dy ? a.drop_p : 0.f; p.drop_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    p.seed = a.seed; p.site = a.site;
    p.dpooled = a.dpooled; p.pool_scale = a.pool_scale; p.dh_n = a.dh_n;
    p.sv0 = a.sv0; p.sv1 = a.sv1; p.sv2 = a.sv2; p.sv3 = a.sv3;
    p.dgi = a.dgi; p.lddg = a.lddg ? a.lddg : 3 * a.H; p.dghn = a.dghn; p.lddghn = a.lddghn ? a.lddghn : a.H; p.dbpart = a.dbpart;
    p.dgpk = a.dg_pk;
    DEP_CHECK_ARG(!a.sv16 || a.split);               // the 16-bit saved gates exist in split-precision mode only
    DEP_CHECK_ARG(!a.bf16st || (a.H == 256 && a.split && a.sv16 && a.dg_pk));      // bf16-storage mode: H = 256, burst kernel (checked below via dg_pk)
    DEP_CHECK_ARG(a.dbpart_rows >= nbt);
    const size_t pay = (size_t)2 * nbtp_max * NC * BT * a.H * sizeof(float);
    DEP_CHECK_ARG(PAYLOAD_OFF + pay <= xbuf_bytes && (size_t)nbtp_max * NC <= 256);
    p.status = (unsigned*)xbuf; p.flags = (unsigned*)(hdr_base(xbuf, a.hdr_slot) + FLAG_OFF); p.hello = (unsigned*)(hdr_base(xbuf, a.hdr_slot) + HELLO_OFF);
    p.nofast = nofast_env();
    p.payload = (float*)((char*)xbuf + PAYLOAD_OFF); p.payload_bytes = (unsigned)pay;
    p.trace = trace_env() ? (long long*)(hdr_base(xbuf, a.hdr_slot) + TRACE_OFF) : nullptr;
    DepProfScope prof(DEP_PROF_GRU_BWD, a.stream);
    // burst length 4 (the round-1 schedule KB = 0 serves H = 512 only), per-wave epoch flags, non-temporal one-touch streams:
    // the measured winners of rounds 2-5 (profiles/r04_ab_pairs.txt, r05_final_ab_switches.txt); the losers live in the git history
    p.wflags = 1; p.ntstream = 1; p.ntload = 1;
    // (one tile's rows of the widest array must fit a 32-bit buffer offset)
    { const int mxl = p.lddg > p.lddy ? p.lddg : p.lddy; DEP_CHECK_ARG((size_t)(BT * a.T + 1) * (mxl > p.ldy ? mxl : p.ldy) * 4 < 0xffffffffull); }
    const int kb = a.H >= 512 ? 0 : 4;                // H = 512: 192 weight registers per compute wave leave no room for a second wave per SIMD
    // H = 256, split products: the all-gather exchange of the members' gate gradients (round 5); every other shape / the exact-fp32 mode: the
    // reduce-scatter of fp32 partial dh
    const bool ag = a.H == 256 && a.split && kb == 4;
    const size_t lds = ag ? burst_lds_bytes_ag(4) + 2048
                          : (kb ? (burst_lds_bytes(kb) > EXCLUSIVE_LDS ? burst_lds_bytes(kb) : EXCLUSIVE_LDS) : EXCLUSIVE_LDS) + 2048;
    p.trall_off = (int)((lds - 2048) / 4);
    static bool attr_b = false;
    if (!attr_b) {
#define DEP_BWD_ATTR1(N, S, V, X) (void)hipFuncSetAttribute((const void*)gru_bwd_cluster_r1<N, S, V, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((V ? burst_lds_bytes(V) > EXCLUSIVE_LDS ? burst_lds_bytes(V) : EXCLUSIVE_LDS : EXCLUSIVE_LDS) + 2048))
#define DEP_BWD_ATTR(N, S, V) do { DEP_BWD_ATTR1(N, S, V, false); if (S) DEP_BWD_ATTR1(N, true, V, true); } while (0)
        DEP_BWD_ATTR(1, false, 4); DEP_BWD_ATTR(1, true, 4);
        DEP_BWD_ATTR(2, false, 4); DEP_BWD_ATTR(2, true, 4); DEP_BWD_ATTR1(4, false, 4, false);
        DEP_BWD_ATTR(8, false, 0); DEP_BWD_ATTR(8, true, 0);
        (void)hipFuncSetAttribute((const void*)gru_bwd_cluster_r1<4, true, 4, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(burst_lds_bytes_ag(4) + 2048));
        (void)hipFuncSetAttribute((const void*)gru_bwd_cluster_r1<4, true, 4, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(burst_lds_bytes_ag(4) + 2048));
        (void)hipFuncSetAttribute((const void*)gru_bwd_cluster_r1<4, true, 4, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(burst_lds_bytes_ag(4) + 2048));
#undef DEP_BWD_ATTR1
#undef DEP_BWD_ATTR
        attr_b = true;
    }
    if (p.dgpk) {       // the PK image needs the burst kernel's flush, the 4H-wide rows and whole step pairs (dep_cluster_bwd_pk_ok)
        DEP_CHECK_ARG(kb == 4 && a.split && a.T % 2 == 0 && a.lddg == 4 * a.H && a.lddghn == 4 * a.H && a.dghn == a.dgi + 3 * a.H);
    }
    for (int b0 = 0; b0 < a.B; b0 += CH) {
        const int cb = a.B - b0 < CH ? a.B - b0 : CH;
        p.b0 = b0; p.nbtp = (dep_cdiv(cb, BT) + 7) / 8 * 8;
        // flags / hello words only: the status word is sticky over every sweep of a step (cleared by dep_rnn_forward)
        { const int rc_h = hdr_prepare(xbuf, a.hdr_slot, a.hdr_clean && b0 == 0, a.stream); if (rc_h) return rc_h; }
        dim3 grid(NC * p.nbtp);
        const dim3 block(kb ? CT + SVC_THREADS : CT);
#define DEP_BWD_LAUNCH1(N, S, X) DEP_LAUNCH((gru_bwd_cluster_r1<N, S, 4, X>), grid, block, lds, a.stream, p)
#define DEP_BWD_LAUNCH(N, S) do { if (S && a.sv16) DEP_BWD_LAUNCH1(N, true, true); else DEP_BWD_LAUNCH1(N, S, false); } while (0)
        switch (a.H) {                                // NTW = H / 64
            case 64: if (a.split) DEP_BWD_LAUNCH(1, true); else DEP_BWD_LAUNCH(1, false); break;
            case 128: if (a.split) DEP_BWD_LAUNCH(2, true); else DEP_BWD_LAUNCH(2, false); break;
            case 256:
                if (ag && a.bf16st) DEP_LAUNCH((gru_bwd_cluster_r1<4, true, 4, true, true, true>), grid, block, lds, a.stream, p);
                else if (ag && a.sv16) DEP_LAUNCH((gru_bwd_cluster_r1<4, true, 4, true, false, true>), grid, block, lds, a.stream, p);
                else if (ag) DEP_LAUNCH((gru_bwd_cluster_r1<4, true, 4, false, false, true>), grid, block, lds, a.stream, p);
                else DEP_BWD_LAUNCH(4, false);        // exact-fp32 mode
                break;
            default:                                  // 512: round-1 schedule only (kb == 0)
                if (a.split && a.sv16) DEP_LAUNCH((gru_bwd_cluster_r1<8, true, 0, true>), grid, block, lds, a.stream, p);
                else if (a.split) DEP_LAUNCH((gru_bwd_cluster_r1<8, true, 0>), grid, block, lds, a.stream, p);
                else DEP_LAUNCH((gru_bwd_cluster_r1<8, false, 0>), grid, block, lds, a.stream, p);
                break;
        }
#undef DEP_BWD_LAUNCH1
#undef DEP_BWD_LAUNCH
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}
