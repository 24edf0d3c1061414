// Two-layer GRU BPTT as ONE cluster-parallel launch (the backward twin of rnn_fused2.hip), round 5: ALL-GATHER form.
//
// Per 16-utterance tile a cluster of NC = 8 workgroups (one per CU, co-resident); member c owns hidden units [32c, 32c+32) of both
// layers.  Twelve waves, three groups of four, each group keeping the 32 COLUMNS of one weight matrix that belong to the member's units
// (all 3H = 768 rows, (hi, lo) bf16 planes, 96 VGPRs per wave: wave gw = the 192 rows of source members 2gw, 2gw+1 = six 32-wide k-steps):
//   group 0 : [dr, dz, dn*r](l1, t)   x W_hh(l1)[:, own 32]  ->  dh1_{t-1}, own columns, COMPLETE     + layer 1's gate gradients
//   group 1 : [dr, dz, dn  ](l1, t)   x W_ih(l1)[:, own 32]  ->  d(dropout(y0))_t = the gradient entering layer 0 at the member's OWN units
//                                                                (the former dX GEMM of layer 1 and its HBM round trip) + every HBM stream
//   group 2 : [dr, dz, dn*r](l0, t')  x W_hh(l0)[:, own 32]  ->  dh0_{t'-1}                            + layer 0's gate gradients
// What travels between members is each member's own 16 x 32 gate gradients, ONCE, as the (hi, lo) bf16 words the MFMAs read, in
// B-fragment order (layer 1: dr, dz, dn*r, dn = 8 KB, layer 0: 6 KB per member and step; the reduce-scatter form of rounds 2-4 published
// three 16 KB blocks of fp32 partials per member and step and lost to the two per-layer sweeps).  Consumers read the fragments straight
// from the exchange buffer into registers (rnn_cluster_bwd.hip, AG); the four K-quarter partials of a product meet in LDS (`red').
// Group 1 works ONE STEP BEHIND: in fused step v it multiplies the layer-1 gate gradients of step v-1 -- complete and visible since the
// barrier that ended step v-1, so it never polls -- and layer 0 runs TWO steps behind layer 1 (t0 = T+1-v).  That slack is what lets the
// group own the member's HBM streams: its fragment requests go out first, the streams behind them, and everything it has in flight has
// landed before the step's barrier -- no HBM access is ever in front of a wave that the cluster is waiting for.
// Fused step v (0..T+1):  [groups 0 / 2: gate gradients (2 elements per thread) -> publish words -> acknowledged -> own flag (per wave)
//   -> poll the 8 flags of the two source members -> 12 fragment loads (two rounds of six: registers) + 36 MFMAs -> partial to red]
//   [group 1: fragments of step v-1 + 36 MFMAs -> red ; inputs of step v+1 HBM -> registers -> LDS, gate gradients of step v-1 LDS -> HBM]
//   -> ONE barrier -> [K-quarter sums: dh1_rec (group 0), dh0_rec and the masked dy0 (group 2)].
// Layer 1's exchange buffer is triple-buffered (group 1 reads step v-1 while step v+1 may already be written by a fast member), layer
// 0's double-buffered.  Exchange protocol, same-XCD fast path, bounded spins, sticky status: rnn_cluster_common.h.
#include "rnn_cluster_common.h"

namespace {
using namespace depc;

constexpr int BH = 256, BNC = 8, BTHREADS = 768;
constexpr int N_IBUF = 11, N_OBUF = 8;        // input arrays: l1 r,z,n,hn,h_{t-1},[dy] ; l0 r,z,n,hn,h_{t-1}  (10 without dy: five per wave half)
constexpr unsigned L1_MEMBER = 8 * 1024, L0_MEMBER = 6 * 1024;      // bytes: [gate][plane][64 lanes][16 B] -- l1: dr, dz, dn*r, dn ; l0: dr, dz, dn*r
constexpr int N_RED = 3 * 4 * 2 * 256;        // floats of one step parity: [group][K quarter = wave][own tile][64 lanes][4]
constexpr int IARR = BT * 32;                 // an fp32 INPUT array in LDS: [16 utterances][32 units] unpadded -- written by LDS-DMA (a wave instruction fills 1 KB
                                              // contiguously), 16-byte pieces XOR-swizzled by the row so that the gate threads' column reads spread over the banks
// SV16: the saved gates r, z, n arrive as 16-bit fixed point (rnn_cluster_common.h): half-size arrays, ONE DMA instruction each, decoded by the gate threads.
// PK: the gate gradients leave as the 4H-wide PK image of gemm_bf16x3.hip ((hi, lo) bf16 pairs of two consecutive steps in two rows): a third write-out
// slot, flushed every other step.
// BF (bf16-storage mode, dep_set_gemm_mode(3); implies SV16 and PK): hn and the hidden sequences are bf16 arrays too (half-size in LDS, one DMA instruction
// each, decoded by the gate threads), and only the hi rows of the PK image are written (PKH).
constexpr int gate_arr(bool sv16) { return sv16 ? IARR / 2 : IARR; }
constexpr int hid_arr(bool bf) { return bf ? IARR / 2 : IARR; }
constexpr int ipar_floats(bool sv16, bool hasdy, bool bf = false) { return 2 * (3 * gate_arr(sv16) + 2 * hid_arr(bf)) + (hasdy ? IARR : 0); }     // floats of one step slot of ibuf
// float offset of input array a (l1: r, z, n, hn, hp, [dy] ; l0: r, z, n, hn, hp) inside a slot
constexpr int iarr_off(int a, bool sv16, bool hasdy, bool bf = false) {
    const int n1 = hasdy ? 6 : 5, l1sz = 3 * gate_arr(sv16) + 2 * hid_arr(bf) + (hasdy ? IARR : 0);
    const int k = a < n1 ? a : a - n1;
    return (a < n1 ? 0 : l1sz) + (k < 3 ? k * gate_arr(sv16) : 3 * gate_arr(sv16) + (k - 3 < 2 ? (k - 3) * hid_arr(bf) : 2 * hid_arr(bf)));
}
// Prefetch distance of the input streams (fused steps): 2 where the LDS has room for a third input slot (16-bit gates, no external dy) -- the DMA
// requests then go out AFTER the step's critical fragment loads were issued and land during the NEXT step's gate phase, when nothing latency-critical
// uses the CU's memory pipeline (a CU returns loads in issue order across its waves: an HBM request in flight holds every later L2 hit back).
constexpr int pf_dist(bool sv16, bool hasdy) { return (sv16 && !hasdy) ? 2 : 1; }
constexpr size_t fb_lds_bytes(bool sv16, bool hasdy, bool pk, bool bf = false) {
    return (size_t)(2 * N_RED + (pf_dist(sv16, hasdy) + 1) * ipar_floats(sv16, hasdy, bf) + ((pk ? 3 : 2) * N_OBUF + 8 + 2) * IARR + 64 + 192) * sizeof(float);      // (+192: DEP_TRACE stamps)
}

struct FB {
    int B, T, nbtp, b0;
    const u32x4* wh1; const u32x4* wi1; const u32x4* wh0;          // pack_cluster_bwd_split images ([K member][out tile][gate][plane][lane])
    const float* y1; const float* y0;                               // forward hidden sequences (h_{t-1})
    const float* sv1; const float* sv0; unsigned svstride;          // saved r | z | n | hn, svstride floats apart
    const float* dy; const float* dpooled; float pool_scale; const float* dhn1; const float* dhn0;
    float drop_p, drop_scale; uint64_t seed; uint32_t site;
    float* dgi1; float* dghn1; float* dgi0; float* dghn0;           // (B*T, 3H) / (B*T, H); PK: dgi1 / dgi0 are the (B*T, 4H) images, dghn unused
    int lddg, lddghn;                                               // row strides of the fp32 arrays (3H / H, or 4H / 4H)
    float* dbpart1; float* dbpart0;                                 // [batch tile][4][H] bias-gradient partials
    unsigned* status; unsigned* flags1; unsigned* flags0; unsigned* hello;      // per-wave epoch flags of the layer-1 / layer-0 publishers: [tile][member][4]
    float* payload; unsigned payload_bytes; unsigned l0_off;        // layer 1's three buffers, then (at l0_off bytes) layer 0's two
    int nofast;
    // input streams as buffer resources (LDS-DMA): per layer ONE base below its y / saved-gate arrays, the arrays as byte offsets from it
    const char* sb1; const char* sb0; unsigned sbytes1, sbytes0;
    unsigned o_sv1, o_y1, o_sv0, o_y0;                              // saved gates r (then z, n, hn at + k svstride floats), forward sequence
    long long* trace;                                               // DEP_TRACE=1: stamps of workgroup 0 (tools/trace_fbwd.py), else nullptr
};

// Registers: 3 waves per SIMD -> 168 VGPRs, 96 of them weights.  Per-thread indices are re-derived every step from a laundered
// threadIdx.x (see rnn_fused2.hip), the fragments come in two rounds of six (24 registers), and the per-role persistent state shares
// six vector registers:
//   groups 0 / 2: st[0] = (dh_rec.xy, dpool.xy)  st[1] = (db_r.xy, db_z.xy)  st[2] = (db_n.xy, db_hn.xy)
//                 st[3] = layer 0: (dy0.xy = masked gradient from layer 1, mask.xy) ; layer 1: (d*z .xy, -, -)
//   group 1     : nothing -- its input streams go HBM -> LDS directly (buffer_load ... lds), no staging registers (the first all-gather build
//                 staged them in 20-24 VGPRs: 31-52 VGPR spills, weight fragments reloaded from scratch inside the MFMA chain)
// The bias-gradient accumulators (8 more registers) live in LDS (dbl): one read-modify-write of four float2 per thread and step.
#define BSTMP(slot) do { if (TRACE && trl && v >= 100 && v < 104) trl[(v - 100) * 8 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)
template <bool DROP, bool HASDY, bool SV16, bool PK, bool TRACE = false, bool BF = false>
__global__ __launch_bounds__(BTHREADS) void gru2_bwd_fused(FB p) {
    static_assert(!PK || SV16, "the PK write-out's third slot needs the LDS the 16-bit gates free");
    static_assert(!BF || (SV16 && PK), "bf16 storage: 16-bit gates, PKH image");
    constexpr int IPAR = ipar_floats(SV16, HASDY, BF), OSL = PK ? 3 : 2, PF = pf_dist(SV16, HASDY), ISL = PF + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T;
    const int c = blockIdx.x / p.nbtp, bt = blockIdx.x % p.nbtp;
    if (p.b0 + bt * BT >= p.B) return;
    if (ld_agent(p.status) != 0) return;           // an earlier sweep of this step gave up (sticky status)
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w >> 2, gw = w & 3;               // 0: layer-1 recurrent, 1: layer-1 input + HBM streams, 2: layer-0 recurrent
    const int shalf = gw >> 1;
    float* red = smem;                                // [2 step parities][3 groups][4 K quarters][2 tiles][64][4]
    float* ibuf = smem + 2 * N_RED;                   // [2 step parities][11][16][32, swizzled]: l1 r,z,n,hn,hp,dy ; l0 r,z,n,hn,hp
    float* obuf = ibuf + ISL * IPAR;                  // [OSL step slots][8][16][32, swizzled] : l1 dr,dz,dn,dn*r ; l0 dr,dz,dn,dn*r
    float* dbl = obuf + OSL * N_OBUF * IARR;          // [2 layers][4][16][32, swizzled]: bias-gradient accumulators of the gate threads
    float* mbuf = dbl + 8 * IARR;                     // [2 step parities][16][32, swizzled]: dropout mask values of the dy0 a step ends with (drawn by group 1)
    unsigned* sig = reinterpret_cast<unsigned*>(mbuf + 2 * IARR);      // count of critical waves that have issued their last fragment requests (monotonic)
    const int b0t = p.b0 + bt * BT;

    u32x4 wq[2][6][2];                                // [own output tile][k-step = (source member 2gw + ks/3, gate ks%3)][hi, lo]
    {
        const u32x4* wimg = grp == 0 ? p.wh1 : (grp == 1 ? p.wi1 : p.wh0);
        const int lane = tid & 63;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    wq[i][ks][pl] = wimg[(size_t)((((2 * gw + ks / 3) * 16 + 2 * c + i) * 3 + ks % 3) * 2 + pl) * 64 + lane];
    }
    f32x4 st[4];                                      // (st[1], st[2] unused: the accumulators moved to LDS)
#pragma unroll
    for (int i = 0; i < 4; ++i) st[i] = zero4();
    for (int i = tid; i < 8 * IARR; i += BTHREADS) dbl[i] = 0.f;
    if (tid == 0) *sig = 0u;
    if (TRACE && tid < 192) reinterpret_cast<float*>(sig + 64)[tid] = 0.f;

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.payload, 0, p.payload_bytes, 0x00020000);
    const unsigned par1 = (unsigned)p.nbtp * BNC * L1_MEMBER, par0 = (unsigned)p.nbtp * BNC * L0_MEMBER;      // bytes of one buffer
    unsigned* tflags1 = p.flags1 + bt * BNC * 4;
    unsigned* tflags0 = p.flags0 + bt * BNC * 4;
    unsigned* myflag = (grp == 0 ? tflags1 : tflags0) + c * 4 + gw;
    const int sx = p.nofast ? 0 : cluster_same_xcd(p.hello + bt * BNC, BNC, c, p.status);
    if (sx < 0) return;
    const bool fast = sx == 1;

    // ---- group 1's input streams: HBM -> LDS by DMA.  Inputs of fused step uu: layer 1 at t1 = T-1-uu (r, z, n, hn, h_{t1-1}, dy), layer 0
    // at t0 = T+1-uu (r, z, n, hn, h_{t0-1}).  One wave instruction moves 8 utterance rows x 128 bytes of one array (1 KB, contiguous in LDS):
    // instruction id q -> array q / 2, row half q % 2; wave gw issues q = gw, gw + 4, ...  Lane -> (row lane / 8, LDS piece lane % 8), and it
    // fetches the GLOBAL piece (lane % 8) ^ (row % 8): the swizzle the readers undo.  Rows past the batch re-read the last utterance (finite
    // values in rows whose results are never stored); a step outside the sequence is written as zeros.
    __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.sb1, 0, p.sbytes1, 0x00020000);
    __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.sb0, 0, p.sbytes0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, HASDY ? 0xfffffff0u : 0u, 0x00020000);
    typedef __attribute__((address_space(3))) void* ldsp;
    // per-lane parts of the DMA addresses (row of the utterance's step 0, the member's columns, the swizzled piece): computed BEFORE the issue
    // signal is awaited -- behind it only the requests themselves remain.  [0]: 16-bit arrays (lane -> row lane / 4), [1], [2]: the two row
    // halves of an fp32 array (lane -> row 8 hh + lane / 8).  The step's time index travels in the scalar offset.
    auto stage_prep = [&](int tv, unsigned (&vo)[3]) {
        const int ln = tv & 63;
        { const int row = ln >> 2; int b = b0t + row; b = b < p.B ? b : p.B - 1;
          vo[0] = ((unsigned)b * (unsigned)T * BH + (unsigned)c * 32u) * 2u + (unsigned)(((ln & 3) ^ ((row >> 1) & 3)) * 16); }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int row = hh * 8 + (ln >> 3); int b = b0t + row; b = b < p.B ? b : p.B - 1;
            vo[1 + hh] = ((unsigned)b * (unsigned)T * BH + (unsigned)c * 32u) * 4u + (unsigned)(((ln & 7) ^ (row & 7)) * 16);
        }
    };
    // (GW = this wave's index as a compile-time value: the instruction list of a wave is then straight-line code.  Walking all 14-16 candidate
    // slots with a wave-uniform test each cost ~400 scalar instructions per step -- 1200 ticks on a SIMD that three MFMA-issuing waves share --
    // and this group is the one the step's barrier waits for.)
    auto stage_w = [&](auto GW, int tv, int uu, const unsigned (&vo)[3]) {
        constexpr int gwc = decltype(GW)::value;
        constexpr int N1 = HASDY ? 6 : 5, NA = N1 + 5;
        const int ln = tv & 63;
        const int t1 = T - 1 - uu, t0 = T + 1 - uu;
        int q = 0;                                        // instruction counter: compile-time after unrolling; wave gw takes q % 4 == gw
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const bool l1 = a < N1;
            const int k = l1 ? a : a - N1;                // 0..3 saved gates, 4 h_{t-1}, 5 dy (layer 1 only)
            const bool g16 = (SV16 && k < 3) || (BF && (k == 3 || k == 4));      // a 16-bit array: 16 rows x 64 bytes = ONE instruction (lane -> row lane / 4, piece lane % 4)
            const int t = (l1 ? t1 : t0) - (k == 4 ? 1 : 0);
            const bool on = (l1 ? t1 >= 0 : (uu >= 2 && t0 >= 0)) && t >= 0;
#pragma unroll
            for (int hh = 0; hh < (g16 ? 1 : 2); ++hh, ++q) {
                if ((q & 3) != gwc) continue;             // compile-time
                float* dst = ibuf + (uu % ISL) * IPAR + iarr_off(a, SV16, HASDY, BF) + hh * 256;
                if (on) {
                    const unsigned so = (k < 4 ? (l1 ? p.o_sv1 : p.o_sv0) + (unsigned)k * p.svstride * 4u : (k == 4 ? (l1 ? p.o_y1 : p.o_y0) : 0u))
                                        + (unsigned)t * (g16 ? BH * 2u : BH * 4u);
                    const unsigned v_ = g16 ? vo[0] : vo[1 + hh];
                    // aux 2 = nt: one-touch streams must not displace the exchange buffer from this XCD's L2
                    if (k == 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (ldsp)dst, 16, v_, so, 0, 2);
                    else if (l1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (ldsp)dst, 16, v_, so, 0, 2);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (ldsp)dst, 16, v_, so, 0, 2);
                } else {
                    *reinterpret_cast<f32x4*>(dst + ln * 4) = zero4();
                }
            }
        }
    };
    auto stage = [&](int tv, int uu, const unsigned (&vo)[3]) {
        switch (gw) {
            case 0: stage_w(std::integral_constant<int, 0>{}, tv, uu, vo); break;
            case 1: stage_w(std::integral_constant<int, 1>{}, tv, uu, vo); break;
            case 2: stage_w(std::integral_constant<int, 2>{}, tv, uu, vo); break;
            default: stage_w(std::integral_constant<int, 3>{}, tv, uu, vo); break;
        }
    };
    // an input pair of the gate threads: utterance row j, units (ul, ul+1) -> swizzled float offset inside an array
    auto iswz = [](int j, int ul) { return j * 32 + (((ul >> 2) ^ (j & 7)) << 2) + (ul & 3); };
    // ... and of a 16-bit array (one word = the pair): 16 words per row, pieces of four words swizzled by row / 2
    auto iswz16 = [](int j, int ul) { return j * 16 + (((ul >> 3) ^ ((j >> 1) & 3)) << 2) + ((ul & 7) >> 1); };
    auto flush = [&](int tv, int u) {                 // gate gradients of fused step u: LDS -> HBM
        const int rem = tv & 127, su = rem >> 3, sqd = rem & 7;
        if (b0t + su >= p.B) return;
        const unsigned row0 = (unsigned)(b0t + su) * T;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int a = pr * 2 + shalf;                 // 0..3 layer 1 (dr, dz, dn, dn*r), 4..7 layer 0
            const bool l1 = a < 4;
            const int k = a & 3;
            const int t = l1 ? T - 1 - u : T + 1 - u;
            const bool on = l1 ? (u <= T - 1) : (u >= 2);
            float* base = k < 3 ? (l1 ? p.dgi1 : p.dgi0) : (l1 ? p.dghn1 : p.dghn0);
            const size_t off = k < 3 ? (size_t)(row0 + t) * p.lddg + k * BH : (size_t)(row0 + t) * p.lddghn;
            if (on) *reinterpret_cast<f32x4*>(base + off + c * 32 + sqd * 4) = ld4(obuf + ((u % OSL) * N_OBUF + a) * IARR + su * 32 + ((sqd ^ (su & 7)) << 2));
        }
    };
    // PK: the step pair (ua, ua + 1), ua even, of both layers = rows (t_even + 1, t_even): row t_even takes bf16hi(x[t_even]) | bf16hi(x[t_even+1]) << 16
    // per column, row t_even + 1 the residual pairs -- exactly what the GEMM staging's split would form (rnn_cluster_bwd.hip svc_flush_pk)
    __amdgpu_buffer_rsrc_t rso1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.dgi1, 0, (unsigned)((size_t)p.B * T * 4 * BH * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rso0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.dgi0, 0, (unsigned)((size_t)p.B * T * 4 * BH * 4), 0x00020000);
    struct PkOut { u32x4 h[4], l[4]; unsigned go[4]; unsigned on; };
    auto flush_pk_prep = [&](int tv, int ua, PkOut& o) {      // LDS reads + the (hi, lo) split: no memory traffic, done while the issue signal is awaited
        const int rem = tv & 127, su = rem >> 3, sqd = rem & 7;
        o.on = 0u;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int a = pr * 2 + shalf;                 // 0..3 layer 1 (dr, dz, dn, dn*r), 4..7 layer 0
            const bool l1 = a < 4;
            const int k = a & 3;
            const bool on = (b0t + su < p.B) && (l1 ? (ua + 1 <= T - 1) : (ua >= 2));
            const int te = l1 ? T - 2 - ua : T - ua;      // the even row of the pair = the LATER step's time index
            const int so_ = su * 32 + ((sqd ^ (su & 7)) << 2);
            const f32x4 xo = ld4(obuf + ((ua % 3) * N_OBUF + a) * IARR + so_);            // step ua     = row te + 1
            const f32x4 xe = ld4(obuf + (((ua + 1) % 3) * N_OBUF + a) * IARR + so_);      // step ua + 1 = row te
#pragma unroll
            for (int e = 0; e < 4; ++e) { unsigned hh, ll; split_pair(xe[e], xo[e], hh, ll); o.h[pr][e] = hh; o.l[pr][e] = ll; }
            o.go[pr] = (((unsigned)(b0t + su) * (unsigned)T + (unsigned)te) * 4u * BH + (unsigned)k * BH + (unsigned)c * 32u + (unsigned)sqd * 4u) * 4u;
            o.on |= on ? (1u << pr) : 0u;
            __builtin_amdgcn_sched_barrier(0);            // one array at a time: all eight LDS reads in flight at once cost 32 registers this kernel does not have
        }
    };
    auto flush_pk_issue = [&](const PkOut& o) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            if (!((o.on >> pr) & 1u)) continue;
            if (pr * 2 + shalf < 4) { __builtin_amdgcn_raw_buffer_store_b128(o.h[pr], rso1, o.go[pr], 0, 2 /* nt */); if constexpr (!BF) __builtin_amdgcn_raw_buffer_store_b128(o.l[pr], rso1, o.go[pr] + 4u * BH * 4u, 0, 2); }
            else { __builtin_amdgcn_raw_buffer_store_b128(o.h[pr], rso0, o.go[pr], 0, 2 /* nt */); if constexpr (!BF) __builtin_amdgcn_raw_buffer_store_b128(o.l[pr], rso0, o.go[pr] + 4u * BH * 4u, 0, 2); }
        }
    };
    {   // initial recurrent gradient (dh_n) and the pooling gradient of the top layer
        const int lt = tid & 255, lp = (lt >> 1) & 63, half = lt & 1;
        const int b = b0t + (lp & 15), col = c * 32 + (lt >> 7) * 16 + (lp >> 4) * 4 + 2 * half;
        if (grp != 1 && b < p.B) {
            const float* dhn = grp == 0 ? p.dhn1 : p.dhn0;
            if (dhn) { const float2 v = ld2(dhn + (size_t)b * BH + col); st[0][0] = v.x; st[0][1] = v.y; }
            if (grp == 0 && p.dpooled) { const float2 v = ld2(p.dpooled + (size_t)b * BH + col); st[0][2] = v.x * p.pool_scale; st[0][3] = v.y * p.pool_scale; }
        }
    }
    if (grp == 1) { unsigned vo0[3]; stage_prep(tid, vo0); stage(tid, 0, vo0); if (PF == 2) stage(tid, 1, vo0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();
    // debug stamps (DEP_TRACE=1): thread 0 (group 0), 256 (group 1), 512 (group 2) of workgroup 0, fused steps 100..103, buffered behind sig
    long long* trl = nullptr;
    if (TRACE && p.trace && blockIdx.x == 0 && (tid & 255) == 0) trl = reinterpret_cast<long long*>(sig + 64) + grp * 32;

    for (int v = 0; v <= T + 1; ++v) {
        int tv = tid;
        asm volatile("" : "+v"(tv));                  // launder: indices derived from tv are recomputed per step, not hoisted
        const int lane = tv & 63;
        const int lt = tv & 255, jl = lt >> 7, lp = (lt >> 1) & 63, half = lt & 1;
        const int j = lp & 15, ul = jl * 16 + (lp >> 4) * 4 + 2 * half;       // utterance row, unit pair (ul, ul+1) of the member's 32
        const bool act = grp == 0 ? (v <= T - 1) : (grp == 2 ? (v >= 2) : false);      // gate-gradient role: layer 1 at t = T-1-v, layer 0 at t = T+1-v
        const unsigned epoch = (unsigned)v + 1u;
        BSTMP(0);
        // ---- gate gradients (groups 0 and 2; identical code, role-dependent LDS bases), published at once
        if (act) {
            constexpr int GA = gate_arr(SV16);
            const float* il = ibuf + (v % ISL) * IPAR + (grp == 0 ? 0 : iarr_off(HASDY ? 6 : 5, SV16, HASDY, BF));      // this layer's arrays: r, z, n | hn, hp, [dy]
            float2 r, z, n;
            if constexpr (SV16) {
                const unsigned* iw = reinterpret_cast<const unsigned*>(il) + iswz16(j, ul);
                r = unpack_unorm2(iw[0]); z = unpack_unorm2(iw[GA]); n = unpack_snorm2(iw[2 * GA]);
            } else {
                const float* ig = il + iswz(j, ul);
                r = ld2(ig); z = ld2(ig + GA); n = ld2(ig + 2 * GA);
            }
            float2 hn, hp;
            if constexpr (BF) {                           // bf16 pairs: the value sits in the upper half of its fp32
                const unsigned* ih = reinterpret_cast<const unsigned*>(il + 3 * GA) + iswz16(j, ul);
                const unsigned w0 = ih[0], w1 = ih[hid_arr(true)];
                hn = f2(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)); hp = f2(__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u));
            } else {
                const float* ib = il + 3 * GA + iswz(j, ul);
                hn = ld2(ib); hp = ld2(ib + IARR);
            }
            float2 dyv = f2(0.f, 0.f);
            if (grp == 2) dyv = f2(st[3][0], st[3][1]);   // layer 0: masked gradient from layer 1
            if (HASDY && grp == 0) dyv = ld2(il + 3 * GA + 2 * hid_arr(BF) + iswz(j, ul));
            const float2 d = f2(st[0][0] + st[0][2] + dyv.x, st[0][1] + st[0][3] + dyv.y);
            float2 dn, dz, dr, dnr;
            dn.x = d.x * (1.0f - z.x) * (1.0f - n.x * n.x); dn.y = d.y * (1.0f - z.y) * (1.0f - n.y * n.y);
            dz.x = d.x * (hp.x - n.x) * z.x * (1.0f - z.x); dz.y = d.y * (hp.y - n.y) * z.y * (1.0f - z.y);
            dr.x = dn.x * hn.x * r.x * (1.0f - r.x); dr.y = dn.y * hn.y * r.y * (1.0f - r.y);
            dnr.x = dn.x * r.x; dnr.y = dn.y * r.y;
            // d*z (the part of dh_{t-1} that bypasses the gates) waits for the K-quarter sum: layer 1 parks it in st[3].xy (unused
            // there), layer 0 in st[0].zw (its pooling-gradient slot, which must read zero again at the next gate phase)
            if (grp == 0) { st[3][0] = d.x * z.x; st[3][1] = d.y * z.y; } else { st[0][2] = d.x * z.x; st[0][3] = d.y * z.y; }
            // layer 1 always publishes (the gradient entering layer 0 needs every step's gates); layer 0 not at its last step (t = 0: v = T+1)
            if (grp == 0 || v <= T) {
                unsigned wd[8];
                split_pair(dr.x, dr.y, wd[0], wd[1]); split_pair(dz.x, dz.y, wd[2], wd[3]); split_pair(dnr.x, dnr.y, wd[4], wd[5]);
                split_pair(dn.x, dn.y, wd[6], wd[7]);
                // this pair's word in each of the member's 1 KB blocks (B-fragment order: lane (k-group ul / 8, utterance j), word (ul % 8) / 2)
                const unsigned pw = (unsigned)(half + 2 * ((lp >> 4) & 1) + 4 * j + 64 * (lp >> 5) + 128 * jl) * 4u;
                const unsigned po = grp == 0 ? (unsigned)(v % 3) * par1 + (unsigned)(bt * BNC + c) * L1_MEMBER + pw
                                             : p.l0_off + (unsigned)(v & 1) * par0 + (unsigned)(bt * BNC + c) * L0_MEMBER + pw;
                if (fast) {
#pragma unroll
                    for (int e = 0; e < 6; ++e) __builtin_amdgcn_raw_buffer_store_b32(wd[e], rsrc, po + e * 1024, 0, 0);
                    if (grp == 0) { __builtin_amdgcn_raw_buffer_store_b32(wd[6], rsrc, po + 6 * 1024, 0, 0); __builtin_amdgcn_raw_buffer_store_b32(wd[7], rsrc, po + 7 * 1024, 0, 0); }
                } else {
#pragma unroll
                    for (int e = 0; e < 6; ++e) __builtin_amdgcn_raw_buffer_store_b32(wd[e], rsrc, po + e * 1024, 0, 16);
                    if (grp == 0) { __builtin_amdgcn_raw_buffer_store_b32(wd[6], rsrc, po + 6 * 1024, 0, 16); __builtin_amdgcn_raw_buffer_store_b32(wd[7], rsrc, po + 7 * 1024, 0, 16); }
                }
            }
            float* ob = obuf + ((v % OSL) * N_OBUF + (grp == 0 ? 0 : 4)) * IARR + iswz(j, ul);
            st2(ob, dr); st2(ob + IARR, dz); st2(ob + 2 * IARR, dn); st2(ob + 3 * IARR, dnr);
            {   // bias-gradient accumulators (this thread's own four float2 slots)
                float* da = dbl + (grp == 0 ? 0 : 4) * IARR + iswz(j, ul);
                const float2 a0 = ld2(da), a1 = ld2(da + IARR), a2 = ld2(da + 2 * IARR), a3 = ld2(da + 3 * IARR);
                st2(da, f2(a0.x + dr.x, a0.y + dr.y)); st2(da + IARR, f2(a1.x + dz.x, a1.y + dz.y));
                st2(da + 2 * IARR, f2(a2.x + dn.x, a2.y + dn.y)); st2(da + 3 * IARR, f2(a3.x + dnr.x, a3.y + dnr.y));
            }
        }
        if (v == T + 1) { bar_lds(); break; }             // layer 0's last step (t = 0): nothing left to exchange
        BSTMP(1);
        if (grp != 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's words are acknowledged
            BSTMP(2);
            if (act && lane == 0) { if (fast) st_local(myflag, epoch); else st_agent(myflag, epoch); }
        }
        // (prefetch distance 2: the inputs of step v+1 and the last write-out were requested late in step v-1 and had this group's whole idle time to land)
        if (PF == 2 && grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (PK) {
            // The pair write-out (32 KB per member every other step) goes out HERE, at the top of the step, not behind the issue signal: the
            // critical waves are in their gate phase (LDS only) for the next ~1300 ticks, and this group's own fragment requests queue behind its
            // stores for a round trip it can afford (its product is due at the step's barrier, ~5000 ticks away).  Behind the signal the request
            // queue pushed back for ~2000 ticks on every even step while the critical waves stood at the barrier (profiles/r05_final_trace_bwd.txt).
            if (grp == 1 && v >= 2 && !(v & 1)) { PkOut po; flush_pk_prep(tv, v - 2, po); flush_pk_issue(po); }
        }
        // (The DMA REQUESTS at the top of the step as well -- distance 1, no signal -- measured 1.06 -> 1.27 ms: HBM loads in the CU's queue hold back the
        // critical waves' polls and fragment loads for their whole round trip; posted stores do not.)
        // ---- the group's product: K quarter gw (source members 2gw, 2gw+1) x own 32 columns
        //   group 0: layer 1, step v (needed while a layer-1 step follows) ; group 1: layer 1, step v-1 ; group 2: layer 0, step v
        const bool mact = grp == 0 ? (v <= T - 2) : (grp == 1 ? (v >= 1 && v <= T) : (v >= 2 && v <= T));
        if (mact) {
            // (group 1's data is one step old and normally visible already -- its member's group 0 gathered the same step before the last
            // barrier -- but group 0 does not gather at layer 1's LAST step: group 1 always checks the flags itself, one round trip off the
            // critical path)
            if (!wait_flags((grp == 2 ? tflags0 : tflags1) + 8 * gw, 8, grp == 1 ? epoch - 1u : epoch, p.status, 7)) return;
            BSTMP(3);
            const unsigned mb = grp == 2 ? L0_MEMBER : L1_MEMBER;
            const unsigned src0 = (grp == 2 ? p.l0_off + (unsigned)(v & 1) * par0 : (unsigned)((grp == 0 ? v : v - 1) % 3) * par1)
                                  + (unsigned)(bt * BNC + 2 * gw) * mb + (unsigned)lane * 16u;
            const unsigned g2 = grp == 1 ? 3u * 2048u : 2u * 2048u;      // third k-step of a member: dn for the input path, dn*r for the recurrent ones
            f32x4 acc[2] = {zero4(), zero4()};
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {          // one source member per round: six 1 KB requests, then its 18 MFMAs
                u32x4 gfr[3][2];
                const unsigned so = src0 + (unsigned)rd * mb;
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        gfr[g][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, so + (g == 2 ? g2 : (unsigned)g * 2048u) + (unsigned)pl * 1024u, 0, 16 /* sc1: served by L2 */);
                if (rd == 1 && grp != 1 && lane == 0) sig_raise(sig);       // this critical wave's last requests are in the CU's queue (raised after the FIRST round instead: +6 %, the second round then queues behind the streams)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const bf16x8 gh = __builtin_bit_cast(bf16x8, gfr[g][0]), gl = __builtin_bit_cast(bf16x8, gfr[g][1]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[i][rd * 3 + g][0]), wl = __builtin_bit_cast(bf16x8, wq[i][rd * 3 + g][1]);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gl, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, gh, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gh, acc[i], 0, 0, 0);
                    }
                }
            }
            float* rw = red + ((((v & 1) * 3 + grp) * 4 + gw) * 2) * 256 + lane * 4;
            *reinterpret_cast<f32x4*>(rw) = acc[0]; *reinterpret_cast<f32x4*>(rw + 256) = acc[1];
            BSTMP(4);
        }
        else if (grp != 1 && lane == 0) sig_raise(sig);               // (no product this step: nothing of this wave will be in the queue)
        if (grp == 1) {
            // group 1's HBM streams: next inputs by DMA, then the finished gate gradients.  They go out once the eight critical waves of this
            // member have ISSUED their last fragment requests (an LDS counter: no memory traffic to watch it): an HBM access in the CU's queue
            // holds back every load issued behind it for its whole round trip.  With prefetch distance 2 they then land in the next step's gate
            // phase; with distance 1 they must land before this step's barrier.  Everything that can be prepared -- the mask draw, the write-out's
            // LDS reads and (hi, lo) split, the DMA addresses -- happens BEFORE the signal is awaited (the phase trace showed this group as the
            // step's long pole: 1400-2600 ticks of issue work behind a signal that came 3500-4000 ticks into the step).
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DROP) {
                // the dropout mask of the dy0 this step ends with (t = T - v; same Philox draw as the forward's mask of y0): ~150 VALU
                // instructions that have no business on the gate threads' chain -- two of this group's waves draw the member's 128 blocks
                if (shalf == 1 && v >= 1 && v <= T) {
                    const int rem = tv & 127, su = rem >> 3, sqd = rem & 7;
                    const size_t o = ((size_t)(b0t + su) * T + (T - v)) * BH + c * 32 + sqd * 4;
                    *reinterpret_cast<f32x4*>(mbuf + (v & 1) * IARR + su * 32 + ((sqd ^ (su & 7)) << 2)) = dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
                }
            }
            unsigned vo[3];
            stage_prep(tv, vo);
            auto wait_signal = [&]() {
                const unsigned want = 8u * ((unsigned)v + 1u);
                for (int spin = 0; spin < 20000 && sig_read(sig) < want; ++spin) __builtin_amdgcn_s_sleep(1);
            };
            __builtin_amdgcn_sched_barrier(0);
            wait_signal();
            BSTMP(5);
            stage(tv, v + PF, vo);
            if constexpr (!PK) { if (v >= 1) flush(tv, v - 1); }
        }
        if (PF == 1 && grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // distance 1: the DMA'd inputs of step v+1 must be in LDS at this barrier
        BSTMP(6);
        bar_lds();                                        // the step's ONE barrier: partials in red, next step's inputs in ibuf, this step's write-out read
        BSTMP(7);
        // ---- K-quarter sums (fixed order: deterministic)
        if (grp != 1) {
            const float* rr = red + (v & 1) * N_RED + (jl * 64 + (lp >> 4) * 16 + j) * 4 + 2 * half;      // own pair inside a wave's two accumulator tiles
            if (grp == 0) {                               // dh1_{t-1}
                if (v <= T - 2) {
                    float2 s = f2(0.f, 0.f);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) { const float2 x = ld2(rr + q4 * 512); s.x += x.x; s.y += x.y; }
                    st[0][0] = st[3][0] + s.x; st[0][1] = st[3][1] + s.y;
                }
            } else {
                if (v >= 2 && v <= T) {                   // dh0_{t-1}
                    float2 s = f2(0.f, 0.f);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) { const float2 x = ld2(rr + 2 * 2048 + q4 * 512); s.x += x.x; s.y += x.y; }
                    st[0][0] = st[0][2] + s.x; st[0][1] = st[0][3] + s.y; st[0][2] = 0.f; st[0][3] = 0.f;
                }
                if (v >= 1 && v <= T) {                   // the gradient entering layer 0 at t = T - v (group 1's product), masked
                    float2 s = f2(0.f, 0.f);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) { const float2 x = ld2(rr + 2048 + q4 * 512); s.x += x.x; s.y += x.y; }
                    if constexpr (DROP) { const float2 m = ld2(mbuf + (v & 1) * IARR + iswz(j, ul)); s.x *= m.x; s.y *= m.y; }
                    st[3][0] = s.x; st[3][1] = s.y;
                }
            }
        }
    }
    if (TRACE && trl) { long long* o = p.trace + grp * 32; for (int i = 0; i < 32; ++i) o[i] = trl[i]; }
    if (grp == 1) {                                       // layer 0's last two steps (t = 1, 0) are still in LDS
        if constexpr (PK) { PkOut po; flush_pk_prep(tid, T, po); flush_pk_issue(po); }
        else { flush(tid, T); flush(tid, T + 1); }
    }
    if (grp != 1) {
        // bias-gradient partials [batch tile][4][H]: sum over the 16 utterance rows = lanes that differ in bits 1..4
        float2 a[4];
        {
            const int lt0 = tid & 255, lp0 = (lt0 >> 1) & 63, j0 = lp0 & 15, ul0 = (lt0 >> 7) * 16 + (lp0 >> 4) * 4 + 2 * (lt0 & 1);
            const float* da = dbl + (grp == 0 ? 0 : 4) * IARR + j0 * 32 + (((ul0 >> 2) ^ (j0 & 7)) << 2) + (ul0 & 3);
            const bool rowok = b0t + j0 < p.B;            // (rows past the batch carried a copy of the last utterance: not part of the sums)
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = rowok ? ld2(da + k * IARR) : f2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 2; m <= 16; m <<= 1) { a[k].x += __shfl_xor(a[k].x, m, 64); a[k].y += __shfl_xor(a[k].y, m, 64); }
        const int lt = tid & 255, lp = (lt >> 1) & 63, half = lt & 1;
        const int col = c * 32 + (lt >> 7) * 16 + (lp >> 4) * 4 + 2 * half;
        if ((lp & 15) == 0) {
            float* o = (grp == 0 ? p.dbpart1 : p.dbpart0) + (size_t)(p.b0 / BT + bt) * 4 * BH;
            st2(o + col, a[0]); st2(o + BH + col, a[1]); st2(o + 2 * BH + col, a[2]); st2(o + 3 * BH + col, a[3]);
        }
    }
}

}  // namespace

// The fused backward addresses a layer's input arrays (y, dropped y, r, z, n, hn: one stride apart in the reserve) and its 4H-wide gate-gradient image
// through 32-bit buffer offsets: larger batches take the per-layer sweeps (tile-relative resources) instead.
bool dep_fused2_bwd_fits(int B, int T) { return (size_t)B * T * BH * sizeof(float) * 8 < 0xfffffff0ull; }

size_t dep_fused2_bwd_xbuf_bytes(int B) {
    const int CH = dep_cluster_chunk(BNC, 1, 256);
    const int nbtp = (dep_cdiv(B < CH ? B : CH, BT) + 7) / 8 * 8;
    return PAYLOAD_OFF + (size_t)nbtp * BNC * (3 * L1_MEMBER + 2 * L0_MEMBER) + 4096;
}

int dep_launch_fused2_bwd(const dep_fused2_bwd_args& a, void* xbuf, size_t xbuf_bytes) {
    const int CH = dep_cluster_chunk(BNC, 1, 256), nbt = dep_cdiv(a.B, BT);
    const int nbtp_max = (dep_cdiv(a.B < CH ? a.B : CH, BT) + 7) / 8 * 8;
    FB p{};
    p.B = a.B; p.T = a.T;
    p.wh1 = (const u32x4*)a.wh1; p.wi1 = (const u32x4*)a.wi1; p.wh0 = (const u32x4*)a.wh0;
    p.y1 = a.y1; p.y0 = a.y0; p.sv1 = a.sv1; p.sv0 = a.sv0; p.svstride = (unsigned)a.svstride;
    p.dy = a.dy; p.dpooled = a.dpooled; p.pool_scale = a.pool_scale; p.dhn1 = a.dhn1; p.dhn0 = a.dhn0;
    const bool drop = a.drop_p > 0.f;
    p.drop_p = a.drop_p; p.drop_scale = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f; p.seed = a.seed; p.site = a.site;
    p.dgi1 = a.dgi1; p.dghn1 = a.dghn1; p.dgi0 = a.dgi0; p.dghn0 = a.dghn0; p.dbpart1 = a.dbpart1; p.dbpart0 = a.dbpart0;
    p.lddg = a.lddg ? a.lddg : 3 * BH; p.lddghn = a.lddghn ? a.lddghn : BH;
    DEP_CHECK_ARG(!a.dg_pk || (p.lddg == 4 * BH && a.dghn1 == a.dgi1 + 3 * BH && a.dghn0 == a.dgi0 + 3 * BH));
    DEP_CHECK_ARG(a.dbpart_rows >= nbt && a.wh1 && a.wi1 && a.wh0 && a.dgi1 && a.dgi0 && a.dghn1 && a.dghn0);
    static_assert(DEP_HDR_SLOTS >= 2, "the fused backward keeps layer 0's flags in header slot 1");
    p.status = (unsigned*)xbuf; p.flags1 = (unsigned*)(hdr_base(xbuf, 0) + FLAG_OFF); p.flags0 = (unsigned*)(hdr_base(xbuf, 1) + FLAG_OFF);
    p.hello = (unsigned*)(hdr_base(xbuf, 0) + HELLO_OFF);
    p.payload = (float*)((char*)xbuf + PAYLOAD_OFF); p.nofast = nofast_env();
    p.trace = trace_env() ? (long long*)(hdr_base(xbuf, 0) + TRACE_OFF) : nullptr;
    {   // the input streams' buffer resources: per layer one base below its arrays, 32-bit offsets
        const size_t arr = (size_t)a.B * a.T * BH * sizeof(float);
        auto span = [&](const float* y, const float* sv, const char*& base, unsigned& bytes, unsigned& oy, unsigned& osv) {
            const char* cy = (const char*)y; const char* cs = (const char*)sv;
            base = cy < cs ? cy : cs;
            const char* e1 = cy + arr; const char* e2 = cs + (size_t)3 * a.svstride * sizeof(float) + arr;
            const size_t n = (size_t)((e1 > e2 ? e1 : e2) - base);
            oy = (unsigned)(cy - base); osv = (unsigned)(cs - base); bytes = (unsigned)n;
            return n < 0xfffffff0ull;
        };
        DEP_CHECK_ARG(span(a.y1, a.sv1, p.sb1, p.sbytes1, p.o_y1, p.o_sv1) && span(a.y0, a.sv0, p.sb0, p.sbytes0, p.o_y0, p.o_sv0));
        DEP_CHECK_ARG(arr < 0xfffffff0ull);           // (an external dy is addressed from its own base)
    }
    const bool sv16 = a.sv16 != 0, pk = a.dg_pk != 0, bf = a.bf16st != 0;
    DEP_CHECK_ARG(!bf || (sv16 && pk));               // bf16-storage mode: 16-bit gates, PKH image
    DEP_CHECK_ARG(!pk || (sv16 && a.T % 2 == 0 && (size_t)a.B * a.T * 4 * BH * 4 < 0xfffffff0ull));      // PK: 16-bit gates (LDS), whole step pairs, 32-bit offsets into the image
    static bool attr = false;
    if (!attr) {
#define FB_ATTR(D, Y, S, P) (void)hipFuncSetAttribute((const void*)gru2_bwd_fused<D, Y, S, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds_bytes(S, Y, P))
#define FB_ATTR4(S, P) FB_ATTR(true, true, S, P); FB_ATTR(true, false, S, P); FB_ATTR(false, true, S, P); FB_ATTR(false, false, S, P)
        FB_ATTR4(false, false); FB_ATTR4(true, false); FB_ATTR4(true, true);
#define FB_ATTRB(D, Y) (void)hipFuncSetAttribute((const void*)gru2_bwd_fused<D, Y, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds_bytes(true, Y, true, true))
        FB_ATTRB(true, true); FB_ATTRB(true, false); FB_ATTRB(false, true); FB_ATTRB(false, false);
#undef FB_ATTRB
        (void)hipFuncSetAttribute((const void*)gru2_bwd_fused<true, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds_bytes(true, false, true));
        (void)hipFuncSetAttribute((const void*)gru2_bwd_fused<false, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds_bytes(true, false, true));
#undef FB_ATTR4
#undef FB_ATTR
        attr = true;
    }
    DepProfScope prof(DEP_PROF_GRU_BWD, a.stream);
    for (int b0 = 0; b0 < a.B; b0 += CH) {
        const int cb = a.B - b0 < CH ? a.B - b0 : CH;
        p.b0 = b0; p.nbtp = (dep_cdiv(cb, BT) + 7) / 8 * 8;
        const size_t pay = (size_t)p.nbtp * BNC * (3 * L1_MEMBER + 2 * L0_MEMBER);
        DEP_CHECK_ARG(xbuf && PAYLOAD_OFF + pay <= xbuf_bytes && (size_t)nbtp_max * BNC <= 256 && pay < (1ull << 32));
        p.payload_bytes = (unsigned)pay; p.l0_off = (unsigned)((size_t)3 * p.nbtp * BNC * L1_MEMBER);
        // flags / hello words only (both layers' slots): the status word is sticky over every sweep of a step (cleared by dep_rnn_forward)
        { const int rc_h = hdr_prepare(xbuf, 0, false, a.stream); if (rc_h) return rc_h; }
        { const int rc_h = hdr_prepare(xbuf, 1, false, a.stream); if (rc_h) return rc_h; }
        const dim3 grid(BNC * p.nbtp), blk(BTHREADS);
#define FB_GO(D, Y, S, P) DEP_LAUNCH((gru2_bwd_fused<D, Y, S, P>), grid, blk, fb_lds_bytes(S, Y, P), a.stream, p)
#define FB_GO4(S, P) do { if (drop) { if (a.dy) FB_GO(true, true, S, P); else FB_GO(true, false, S, P); } \
                          else      { if (a.dy) FB_GO(false, true, S, P); else FB_GO(false, false, S, P); } } while (0)
        if (bf) {                                     // bf16-storage mode (never traced)
#define FB_GOB(D, Y) DEP_LAUNCH((gru2_bwd_fused<D, Y, true, true, false, true>), grid, blk, fb_lds_bytes(true, Y, true, true), a.stream, p)
            if (drop) { if (a.dy) FB_GOB(true, true); else FB_GOB(true, false); } else { if (a.dy) FB_GOB(false, true); else FB_GOB(false, false); }
#undef FB_GOB
        }
        else if (p.trace && pk && !a.dy) {            // DEP_TRACE=1: the stamped variant (tools/trace_fbwd.py)
            if (drop) DEP_LAUNCH((gru2_bwd_fused<true, false, true, true, true>), grid, blk, fb_lds_bytes(true, false, true), a.stream, p);
            else DEP_LAUNCH((gru2_bwd_fused<false, false, true, true, true>), grid, blk, fb_lds_bytes(true, false, true), a.stream, p);
        }
        else if (pk) FB_GO4(true, true); else if (sv16) FB_GO4(true, false); else FB_GO4(false, false);
#undef FB_GO4
#undef FB_GO
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}
