// Two-layer GRU forward as ONE cluster-parallel launch: layer 1 runs two steps behind layer 0 (wavefront over the layers),
// so a forward pass costs T + 1 dependent hand-offs instead of 2 T, and layer 1's input projection never exists in HBM.
//
// Per 16-utterance tile a cluster of NC = 8 workgroups (one per CU, all co-resident); member c owns hidden units
// [32c, 32c + 32) of BOTH layers.  Twelve waves, three groups of four (wave = (16-unit column tile jl, K half kh) as in
// gru_fwd_cluster_r1), each group keeping one 96 x 256 weight slice register-resident as (hi, lo) bf16 planes (96 VGPRs):
//   group 0 : W_hh of layer 0     x h0_{s-1}              -> layer-0 step s
//   group 1 : W_hh of layer 1     x h1_{s-3}              -> layer-1 step s-2
//   group 2 : W_ih of layer 1     x dropout(h0_{s-1})     -> input projection of layer-1 step s-1 (consumed one fused step
//             later).  These waves have no gate math: their MFMAs run while groups 0 / 1 do theirs (the matrix pipe is idle
//             then), and they own every HBM stream of the member (prefetch of layer 0's input projection, write-out of h,
//             dropout(h) and the saved gates of both layers) -- the waves on the critical path never touch HBM.
// Fused step s (0..T+1): [groups 0/1: 36 MFMAs per wave | group 2: write-out of step s-1] -> barrier -> [gate math of both
// layers | group 2: its MFMAs] -> payload stores (h0_s, dropout(h0_s), h1_{s-2} as bf16 (hi << 16 | lo) words), drain,
// barrier, ONE flag per member -> [poll the cluster's 8 flags, gather the three 16 x 256 blocks into LDS planes | group 2:
// partial sums to LDS, input-projection prefetch] -> barrier.  Exchange protocol, same-XCD fast path, parity
// double-buffered payload, bounded spins and sticky status: rnn_cluster_common.h.  Products use the 3-term bf16 split
// (w_hi h_lo + w_lo h_hi + w_hi h_hi, fp32 accumulate), everything elementwise is fp32.
// Matrix-pipe time per step and SIMD: 2 x 36 v_mfma_f32_16x16x32_bf16 on the critical path (1152 cycles) + 36 beside the gate math.
#include "rnn_cluster_common.h"

namespace {
using namespace depc;

constexpr int FH = 256, FNC = 8, FKS2 = 4, FTHREADS = 768;
constexpr int FLDHB = FH + 8;                 // bf16 elements per row of a split plane (528-byte rows: conflict-free b128 reads)
constexpr int FPLANE = BT * FLDHB;            // bf16 elements per plane
constexpr int F_RED = 12 * 3 * 256;           // floats: [wave][gate][half of the lane's four][lane][2] -- the gate threads read ONE half of a lane's
                                              // fragment: as [lane][4] their 8-byte reads had a 16-byte stride, a 2-way bank conflict on 6-12 reads per
                                              // thread at the start of the gate phase, when all eight gate waves read at once (-4 % on the launch)
constexpr int OROW = 36;                      // floats per utterance row of gbuf / obuf (32 + 4: an unpadded row puts all 16 rows on one bank)
constexpr int OARR = BT * OROW;               // one [16 utterances][32 units] array
constexpr int F_GBUF = 3 * OARR;              // [gate][16 utterances][32 units]
constexpr int F_OBUF = 12 * OARR;             // per layer 6 slots: h, r, z, n, hn, dropout(h) (layer 0 only)
constexpr int F_REGION = BT * FH;             // words of one payload block (one tile, one tensor)
constexpr int F_BIAS = 3 * 3 * 32;            // b_hh l0, b_hh l1, b_ih l1 slices of the member: [vector][gate][32 units]
constexpr int F_TRACE = 3 * 4 * 8 * 2;        // debug stamps (DEP_TRACE=1): [role 0 / 2 / 1][4 steps][8 slots] 64-bit
constexpr int F_MBUF = OARR;                  // next step's dropout mask values of the member's [16 utterances][32 units] (drawn by group 2)
constexpr size_t F_LDS_BYTES = (size_t)(3 * FPLANE + F_RED + F_GBUF + F_OBUF + F_BIAS + 4 + F_TRACE + 16 + F_MBUF) * sizeof(float);

struct FF {
    int B, T, nbtp, b0;
    const u32x4* wp0; const u32x4* wp1; const u32x4* wpi;        // W_hh l0, W_hh l1, W_ih l1 images (pack_cluster_fwd_split format)
    const float* b_hh0; const float* b_ih1; const float* b_hh1;
    const float* gi; int ldgi;                                    // layer-0 input projection incl. b_ih (B*T, 3H)
    float* y0; float* y1;                                         // (B,T,H) outputs; the other per-layer arrays follow each at
    unsigned ostride; int training;                               // +k*ostride floats: [y0, dropout(y0) (DROP only), r, z, n, hn] / [y1, r, z, n, hn]
    float drop_p, drop_scale; uint64_t seed; uint32_t site;
    float* pooled; float pool_scale; float* hn0; float* hn1;
    unsigned* status; unsigned* flags; unsigned* hello; float* payload; unsigned payload_bytes; int nofast;
    long long* trace;
    unsigned* soft;          // fallback flag (rnn_cluster_common.h), or nullptr: a failed hello then raises the status word
    int force_soft;          // test hook (DEP_FORCE_SOFT_FALLBACK=1): behave as if the hello had timed out
    int ntstore;             // write-out with non-temporal buffer stores (DEP_FWD_NT=1; A/B: no effect, default off)
};

#define FSTAMP(slot) do { if (TRACE && trl && s >= 100 && s < 104) trl[(s - 100) * 8 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ float2 add2(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

// Register budget: 12 waves -> 3 per SIMD -> 168 VGPRs, 96 of them the weight slice.  Every per-thread index (utterance,
// unit pair, LDS / payload offsets) is therefore RE-DERIVED inside the loop from a laundered copy of threadIdx.x (the
// compiler would otherwise hoist ~40 loop-invariant address registers of all three roles out of the loop and spill -- and a
// scratch reload in the streaming waves waits on vmcnt, i.e. on their outstanding HBM stores: 10k cycles per step measured).
// SV16 (saved gates r, z, n as 16-bit fixed point) is a template parameter: as a run-time switch the write-out's two store widths
// cost the launch 4 % (profiles/r04_ab_pairs.txt).
// BF (bf16-storage mode, implies SV16): h, dropout(h) and hn are written as bf16 too (2-byte elements at the same positions of their
// arrays); the recurrence itself is unchanged -- the exchanged h words keep both split planes.
// SX (round 5, after lstm_fwd_cluster's DF = 3): the exchanged words are their own flag.  A step's hand-off used to be: payload stores
// acknowledged -> workgroup barrier -> flag store -> every wave polls the eight flags -> gather loads -- three L2 round trips and a flag's
// flight in a row.  Now a member publishes into one of FOUR slots (step s -> slot s % 4) whose words hold a SENTINEL until they are written
// (0xffffffff: split_word() of a finite h never is; a NaN is published as 0x7fc07fc0; the mask byte's sentinel is 0xff, its values are 0..3)
// and the gather simply loads its pieces until none of their words is the sentinel: no acknowledgement wait, no flag, no poll.  With step
// s's words a member re-arms its words of slot (s + 2) % 4 (they held step s-2: every member finished gathering that before it published
// step s-1, which this member needed to gate step s); that store is acknowledged before the member's next publish is issued (the gather
// loop's vmcnt(0)), and nobody loads slot (s + 2) % 4 for step s+2 before having gathered this member's step s+1 -- a gather never meets
// a slot's previous tenant.  The slots are armed in the prologue, in front of the hello rendezvous (which SX therefore always runs).
constexpr unsigned FX_SENT = 0xffffffffu;
template <bool DROP, bool TRACE, bool SV16, bool BF = false>
__global__ __launch_bounds__(FTHREADS) void gru2_fwd_fused(FF p) {
    static_assert(!BF || SV16, "bf16 storage implies 16-bit gates");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T;
    const int c = blockIdx.x / p.nbtp, bt = blockIdx.x % p.nbtp;
    if (p.b0 + bt * BT >= p.B) return;
    if (ld_agent(p.status) != 0) return;           // an earlier sweep of this step gave up (sticky status)
    if (p.soft && ld_agent(p.soft) != 0) return;   // dispatched after the clusters gave this launch up: the fallback kernels redo it
    if (p.soft && p.force_soft == 1) { if (threadIdx.x == 0) st_agent(p.soft, 1); return; }
    // test hooks for the hello race: 2 = member FNC-1 of tile 0 arrives ~25 ms late (the others time out softly and take
    // their hello words back; the late one must leave on the soft word), 3 = it vanishes right AFTER a complete hello (the others
    // are already past it and must leave quietly at their first flag wait)
    if (p.soft && p.force_soft == 2 && c == FNC - 1 && bt == 0)
        for (int i = 0; i < 8000; ++i) __builtin_amdgcn_s_sleep(127);
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform: role tests become scalar branches
    const int grp = w >> 2, gw = w & 3, jl = gw >> 1, kh = gw & 1;
    const int shalf = gw >> 1;                        // group 2: which array / gate of a pair this wave streams
    const int jt = c * 2 + jl;
    unsigned short* hs0 = reinterpret_cast<unsigned short*>(smem);      // h0 planes: hi at +0, lo at +FPLANE
    unsigned short* hs0d = hs0 + 2 * FPLANE;                              // dropout(h0)
    unsigned short* hs1 = hs0 + 4 * FPLANE;                               // h1
    float* red = smem + 3 * FPLANE;
    float* gbuf = red + F_RED;
    float* obuf = gbuf + F_GBUF;
    float* bias_l = obuf + F_OBUF;                    // biases live in LDS, not in registers (the weight slice needs those)
    float* zpair = bias_l + F_BIAS;                   // two zeros (branch-free gate math), then the debug stamps
    float* mbuf = zpair + 4 + F_TRACE + 16;
    for (int i = tid; i < 3 * FPLANE; i += FTHREADS) smem[i] = 0.f;
    if (tid < 4) zpair[tid] = 0.f;
    if (tid < F_BIAS) {
        const int v = tid / 96, g = (tid / 32) % 3, u = tid & 31;
        const float* src = v == 0 ? p.b_hh0 : (v == 1 ? p.b_hh1 : p.b_ih1);
        bias_l[tid] = src[g * FH + c * 32 + u];
    }
    // the group's weight slice, register-resident for the whole sweep
    u32x4 wq[3][FKS2][2];
    {
        const u32x4* wimg = grp == 0 ? p.wp0 : (grp == 1 ? p.wp1 : p.wpi);
        const int lane = tid & 63;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < FKS2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    wq[g][ks][pl] = wimg[(size_t)((((jt * 3 + g) * 2 + kh) * FKS2 + ks) * 2 + pl) * 64 + lane];
    }
    const unsigned short* bsrc = grp == 0 ? hs0 : (grp == 1 ? hs1 : (DROP ? hs0d : hs0));
    // per-role persistent state shares two vector registers (the roles are wave-uniform, the compiler cannot know):
    //   groups 0 / 1: st0 = (h_prev.x, h_prev.y, pool.x, pool.y), st1 = (mask.x, mask.y, -, -)
    //   group 2     : st0, st1 = the prefetched input-projection pieces
    f32x4 st0 = zero4(), st1 = zero4();

    const unsigned pstride = (unsigned)p.nbtp * 3 * F_REGION;           // payload words (fits 32 bits: <= 2 x 32 x 3 x 4096)
    const unsigned tile_base = (unsigned)bt * 3 * F_REGION;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.payload, 0, p.payload_bytes, 0x00020000);
    {
        // arm this member's words of all four slots (write-through: the placement is not known yet); the hello below is the rendezvous
        if (grp < 2) {
            const int l0 = tid & 63, j0 = l0 & 15, ul0 = jl * 16 + (l0 >> 4) * 4 + 2 * kh;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const unsigned pb = (unsigned)sl * pstride + tile_base;
                __hip_atomic_store((gu64*)(p.payload + (pb + (unsigned)(grp == 0 ? 0 : 2) * F_REGION + j0 * FH + c * 32 + ul0)), ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (DROP && grp == 0) {
                    typedef __attribute__((address_space(1))) unsigned char gu8;
                    __hip_atomic_store((gu8*)(reinterpret_cast<unsigned char*>(p.payload + (pb + (unsigned)F_REGION)) + j0 * (FH / 2) + c * 16 + (ul0 >> 1)), (unsigned char)0xff,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // (with a fallback the hello also runs under DEP_CLUSTER_NOFAST: it is what proves that every member is resident)
    const int sxh = cluster_same_xcd(p.hello + bt * FNC, FNC, c, p.status, p.soft);
    const int sx = (p.nofast && sxh >= 0) ? 0 : sxh;
    if (sx < 0) return;
    if (p.soft && p.force_soft == 3 && c == FNC - 1 && bt == 0) { if (threadIdx.x == 0) st_agent(p.soft, 1); return; }
    const bool fast = sx == 1;
    const int b0t = p.b0 + bt * BT;                   // first utterance of the tile

    // ---- streaming role of group 2: thread -> (utterance su, 16-byte piece sqd of the member's 32 units)
    auto issue_gi = [&](int tv, int t) {
        const int rem = tv & 127, su = rem >> 3, sqd = rem & 7;
        const bool on = b0t + su < p.B && t < T;
        const float* src = p.gi + ((size_t)(b0t + su) * T + t) * p.ldgi + c * 32 + sqd * 4;
        st0 = on ? ld4(src + shalf * FH) : zero4();
        if (shalf == 0) st1 = on ? ld4(src + 2 * FH) : zero4();
    };
    auto write_gbuf = [&](int tv) {
        const int rem = tv & 127, ro = (rem >> 3) * OROW + (rem & 7) * 4;
        *reinterpret_cast<f32x4*>(gbuf + shalf * OARR + ro) = st0;
        if (shalf == 0) *reinterpret_cast<f32x4*>(gbuf + 2 * OARR + ro) = st1;
    };
    // part: 0 = everything, 1 = layer 1's arrays only, 2 = layer 0's only (the steady state splits the write-out over slots Y and Z)
    auto flush = [&](int tv, int s, int part) {       // results of fused step s: LDS -> HBM (16-byte stores)
        const int rem = tv & 127, su = rem >> 3, sqd = rem & 7;
        if (b0t + su >= p.B) return;
        const unsigned so = ((unsigned)(b0t + su) * T) * FH + c * 32 + sqd * 4;       // < 2^28: one array is B*T*H floats
        // obuf slot k of layer l -> array index in the reserve's order (see FF): layer 0 with dropout h,hd,r,z,n,hn
        constexpr int dslot0[6] = {0, DROP ? 2 : 1, DROP ? 3 : 2, DROP ? 4 : 3, DROP ? 5 : 4, 1};
        // Round 4: the steady state (both layers active, training) as straight-line code.  The general loop below decides per array
        // whether / where to store with a dozen wave-uniform branches each (and reloads spilled scalars with v_readlane): the ISA
        // showed ~40 instructions per store, 11 stores per step -- most of this group's ~3000-tick slot Z, which is what groups 0 / 1
        // wait for at barrier #1.
        const bool steady = s >= 2 && s < T && p.training != 0 && !p.ntstore;
        if (!steady && part == 1) return;                 // (outside the steady state slot Z writes everything)
        if (steady) {
            const float* ob = obuf + su * OROW;
            const unsigned e0 = so + (unsigned)s * FH, e1 = so + (unsigned)(s - 2) * FH;
            const size_t os = p.ostride;
            auto stg = [&](float* arr, unsigned e, int a) {       // a gate array: packed 16-bit words (SV16) or fp32
                if constexpr (SV16) *reinterpret_cast<float2*>(reinterpret_cast<unsigned short*>(arr) + e) = ld2(ob + a * OARR + sqd * 2);
                else *reinterpret_cast<f32x4*>(arr + e) = ld4(ob + a * OARR + sqd * 4);
            };
            auto st16 = [&](float* arr, unsigned e, int a) {      // h, dropout(h), hn: fp32, or (BF) packed bf16 pairs like the gates
                if constexpr (BF) stg(arr, e, a);
                else *reinterpret_cast<f32x4*>(arr + e) = ld4(ob + a * OARR + sqd * 4);
            };
            if (shalf == 0) {                         // obuf arrays 0 h0, 2 z0, 4 hn0, 6 h1, 8 z1, 10 hn1
                if (part != 1) { st16(p.y0, e0, 0); stg(p.y0 + dslot0[2] * os, e0, 2); st16(p.y0 + dslot0[4] * os, e0, 4); }
                if (part != 2) { st16(p.y1, e1, 6); stg(p.y1 + 2 * os, e1, 8); st16(p.y1 + 4 * os, e1, 10); }
            } else {                                  // 1 r0, 3 n0, 5 dropout(h0), 7 r1, 9 n1
                if (part != 1) {
                    stg(p.y0 + dslot0[1] * os, e0, 1); stg(p.y0 + dslot0[3] * os, e0, 3);
                    if constexpr (DROP) st16(p.y0 + dslot0[5] * os, e0, 5);
                }
                if (part != 2) { stg(p.y1 + 1 * os, e1, 7); stg(p.y1 + 3 * os, e1, 9); }
            }
            return;
        }
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
            const int a = pr * 2 + shalf;             // obuf array (wave-uniform): 0..5 layer 0, 6..10 layer 1
            const bool l0 = a < 6;
            const int k = l0 ? a : a - 6;
            const bool on = (l0 ? (s < T) : (s >= 2)) && a < 11 && (k == 0 || (k == 5 ? DROP : p.training != 0));
            const int t = l0 ? s : s - 2;
            float* base = l0 ? p.y0 : p.y1;
            const unsigned slot = l0 ? dslot0[k] : k;
            if (on && p.ntstore) {
                // non-temporal: one-touch write-out that allocates in L2 evicts the exchange payload (tools/micro/l2wb.hip)
                __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xfffffff0u, 0x00020000);
                const unsigned eo = (unsigned)slot * p.ostride + so + (unsigned)t * FH;       // element offset inside the layer's arrays (< 2^30: host check)
                if (BF || (SV16 && k >= 1 && k <= 3)) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const float2 q = ld2(obuf + a * OARR + su * OROW + sqd * 2);
                    const u32x2 qv = {__float_as_uint(q.x), __float_as_uint(q.y)};
                    __builtin_amdgcn_raw_buffer_store_b64(qv, rso, (unsigned)slot * p.ostride * 4u + (so + (unsigned)t * FH) * 2u, 0, 2 /* nt */);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ld4(obuf + a * OARR + su * OROW + sqd * 4)), rso, eo * 4u, 0, 2 /* nt */);
                }
            } else if (on) {
                float* arr = base + (size_t)slot * p.ostride;
                if (BF || (SV16 && k >= 1 && k <= 3)) {
                    // r, z (unorm16), n (snorm16) [BF: every array, as bf16]: the gate threads left them PACKED in obuf (two units per word, 16 words per
                    // utterance row) -- four values = one 8-byte copy, no arithmetic in this group (it is the busiest one)
                    const float2 q = ld2(obuf + a * OARR + su * OROW + sqd * 2);
                    *reinterpret_cast<float2*>(reinterpret_cast<unsigned short*>(arr) + (so + (unsigned)t * FH)) = q;
                } else {
                    *reinterpret_cast<f32x4*>(arr + (so + (unsigned)t * FH)) = ld4(obuf + a * OARR + su * OROW + sqd * 4);
                }
            }
        }
    };
    // inter-layer dropout: the mask of step s+1 is drawn while step s waits for the other members
    auto draw = [&](int tv, int t) {
        const int l = tv & 63, j = l & 15, ul = jl * 16 + (l >> 4) * 4 + 2 * kh;
        const size_t o = ((size_t)(b0t + j) * T + t) * FH + c * 32 + ul;
        const f32x4 m = dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
        return f2(kh ? m[2] : m[0], kh ? m[3] : m[1]);
    };
    if (grp == 2) { issue_gi(tid, 0); write_gbuf(tid); issue_gi(tid, 1); }
    if (grp < 2) { st1[0] = 1.f; st1[1] = 1.f; }
    if (DROP && grp == 0) { const float2 m0 = draw(tid, 0); st1[0] = m0.x; st1[1] = m0.y; }
    __syncthreads();

    long long* trl = nullptr;                         // workgroup 0, thread 0 (group 0), thread 512 (group 2), thread 256 (group 1)
    if (TRACE && p.trace && blockIdx.x == 0 && (tid == 0 || tid == 512 || tid == 256)) trl = reinterpret_cast<long long*>(zpair + 4) + (tid == 0 ? 0 : (tid == 512 ? 32 : 64));
    // Group 2 runs ONE BARRIER out of phase with groups 0 / 1 (it passes one extra barrier here and one fewer at the end): its
    // slot X (the MFMAs) then coincides with their slot Y (gate math, matrix pipe idle), its slot Y (partial sums to LDS, staged
    // projection, its own poll + gather of the dropout block) with their slot Z (poll + gather), its slot Z (write-out of the
    // step, projection prefetch) with their next slot X.
    // One loop body, one MFMA code site for all three roles -- two sites cost 16 VGPRs of spills.
    if (grp == 2) bar_lds();
    for (int s = 0; s <= T + 1; ++s) {
        if (grp == 2 && s == T + 1) break;
        FSTAMP(0);
        int tv = tid;
        asm volatile("" : "+v"(tv));                  // launder: everything derived from tv is recomputed per step, not hoisted
        const int lane = tv & 63, j = lane & 15, q = lane >> 4;
        const int ul = jl * 16 + q * 4 + 2 * kh;      // this lane's pair of units (ul, ul+1) inside the member's 32
        // group 0: layer-0 step s ; group 1: layer-1 step s-2 ; group 2: input projection of layer-1 step s-1 (phase 2)
        const bool act = grp == 0 ? (s < T) : (grp == 1 ? (s >= 2) : (s >= 1 && s <= T));
        f32x4 acc[3] = {zero4(), zero4(), zero4()};
        auto matvec = [&]() {
            // the accumulators of the K-half-0 wave start from the group's bias (b_hh l0 / b_hh l1 / b_ih l1): the bias add is free
            if (kh == 0 && !(DROP && grp == 2)) {
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = ld4(bias_l + grp * 96 + g * 32 + jl * 16 + q * 4);
            }
            const int ho = j * FLDHB + kh * 128 + q * 8;
            bf16x8 hh[2], hl[2];                          // two k-steps in flight (registers are the scarce resource here)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                hh[ks] = *reinterpret_cast<const bf16x8*>(bsrc + ho + ks * 32);
                hl[ks] = *reinterpret_cast<const bf16x8*>(bsrc + FPLANE + ho + ks * 32);
            }
#pragma unroll
            for (int ks = 0; ks < FKS2; ++ks) {
                const bf16x8 ch = hh[ks & 1], cl = hl[ks & 1];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[g][ks][0]), wl = __builtin_bit_cast(bf16x8, wq[g][ks][1]);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, cl, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ch, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ch, acc[g], 0, 0, 0);
                }
                if (ks + 2 < FKS2) {
                    hh[ks & 1] = *reinterpret_cast<const bf16x8*>(bsrc + ho + (ks + 2) * 32);
                    hl[ks & 1] = *reinterpret_cast<const bf16x8*>(bsrc + FPLANE + ho + (ks + 2) * 32);
                }
            }
        };
        auto put_red = [&]() {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float* rb = red + (w * 3 + g) * 256 + lane * 2;
                st2(rb, f2(acc[g][0], acc[g][1])); st2(rb + 128, f2(acc[g][2], acc[g][3]));
            }
        };
        // ---- slot X
        if (act) matvec();
        if (grp < 2 && act) {
            put_red();
        }
        FSTAMP(1);
        bar_lds();                                        // groups 0/1: #1 of step s (partial sums in LDS) | group 2: #2 of step s
        FSTAMP(2);
        const unsigned pbase = (unsigned)(s & 3) * pstride + tile_base;
        if (grp == 2) {                                   // ---- slot Y of group 2 (the others are polling / gathering); placed before the gate block so that
                                                          // the accumulators' live range does not span it
            if (act) {
                if constexpr (DROP) {
                    // Round 4: this group multiplied W_ih(l1) with the MASKED but UNSCALED planes of h0 (below): the dropout scale goes onto
                    // the sums, then b_ih (K half 0): W (m * s * h) + b = s * (W (m * h)) + b
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const f32x4 bb = kh == 0 ? ld4(bias_l + 2 * 96 + g * 32 + jl * 16 + q * 4) : zero4();
                        acc[g] = acc[g] * p.drop_scale + bb;
                    }
                }
                put_red();                                // group 1 reads these partial sums in the next step's gate phase
            }
            write_gbuf(tv);                               // layer-0 input projection of step s+1
            flush(tv, s, 1);                              // layer 1's arrays of step s go out here (obuf is complete: the others wrote it before this barrier) ...
            if constexpr (DROP) {
                // layer 0's dropout mask of step s+1 (read by group 0 at the start of its next gate block): the Philox draw is ~150 VALU
                // instructions; inside group 0's gate block it sat on every SIMD's VALU beside both groups' gate math (a third of that
                // phase's instructions).  Here two of this group's waves (the ones with less to write out) draw the member's 128 blocks
                // once, while the others poll and gather.
                if (shalf == 1 && s + 1 < T) {
                    const int rem = tv & 127, su = rem >> 3, sqd = rem & 7;
                    const size_t o = ((size_t)(b0t + su) * T + (s + 1)) * FH + c * 32 + sqd * 4;
                    *reinterpret_cast<f32x4*>(mbuf + su * OROW + sqd * 4) = dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
                }
            }
            // (Until round 4 this group polled the flags and gathered a third payload block, dropout(h0_s), here -- ~2000 ticks that made it
            // the long pole of the step.  Now group 0 publishes the two mask BITS of its unit pair as one byte and the waves that gather h0_s
            // write the masked planes too.)
        }
        if (grp < 2 && act) {
            // branch-free over the two roles: the input-projection term is gi = A + B with
            //   layer 0: A = prefetched projection (gbuf), B = a zero pair ; layer 1: A, B = group 2's two K halves (b_ih inside)
            const int e2 = kh * 128 + lane * 2;           // this lane's pair inside a [2][64][2] fragment block
            if (DROP && grp == 0 && s >= 1) { const float2 mn = ld2(mbuf + j * OROW + ul); st1[0] = mn.x; st1[1] = mn.y; }      // this step's mask (group 2 drew it)
            // both K halves of the lane's pair come back from LDS (the own half too: selecting acc[g][2 kh + i] in registers
            // compiles to a dynamic-index select tree of ~150 instructions)
            const float* po = red + (w * 3) * 256 + e2;
            const float* pp = red + ((w ^ 1) * 3) * 256 + e2;
            const float* pa = grp == 0 ? gbuf + j * OROW + ul : red + ((8 + jl * 2) * 3) * 256 + e2;
            const float* pb = grp == 0 ? zpair : red + ((9 + jl * 2) * 3) * 256 + e2;
            const int sa = grp == 0 ? OARR : 256, sb = grp == 0 ? 0 : 256;
            float2 ov[3], pv[3], va[3], vb[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) { ov[g] = ld2(po + g * 256); pv[g] = ld2(pp + g * 256); va[g] = ld2(pa + g * sa); vb[g] = ld2(pb + g * sb); }
            float2 tot[3], gi[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) { tot[g] = add2(ov[g], pv[g]); gi[g] = add2(va[g], vb[g]); }
            float2 r, z, hn, n, h;
            r.x = fast_sigmoid(gi[0].x + tot[0].x); r.y = fast_sigmoid(gi[0].y + tot[0].y);
            z.x = fast_sigmoid(gi[1].x + tot[1].x); z.y = fast_sigmoid(gi[1].y + tot[1].y);
            hn = tot[2];
            n.x = fast_tanh(gi[2].x + r.x * hn.x); n.y = fast_tanh(gi[2].y + r.y * hn.y);
            h.x = (1.0f - z.x) * n.x + z.x * st0[0]; h.y = (1.0f - z.y) * n.y + z.y * st0[1];
            st0[0] = h.x; st0[1] = h.y; st0[2] += h.x; st0[3] += h.y;       // (the running sum is only read by layer 1)
            const float2 hd = f2(h.x * st1[0], h.y * st1[1]);
            if (s <= T) {                                 // publish first: it is on the other members' critical path
                gu64* dst = (gu64*)(p.payload + (pbase + (unsigned)(grp == 0 ? 0 : 2) * F_REGION + j * FH + c * 32 + ul));
                unsigned w0 = split_word(h.x), w1 = split_word(h.y);
                if (w0 == FX_SENT) w0 = 0x7fc07fc0u;      // (NaN inputs only) never the sentinel
                if (w1 == FX_SENT) w1 = 0x7fc07fc0u;
                const u64 bits = (u64)w0 | ((u64)w1 << 32);
                if (fast) __hip_atomic_store(dst, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(dst, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                {                                         // re-arm the slot two steps ahead
                    const unsigned rb = (unsigned)((s + 2) & 3) * pstride + tile_base;
                    gu64* rdst = (gu64*)(p.payload + (rb + (unsigned)(grp == 0 ? 0 : 2) * F_REGION + j * FH + c * 32 + ul));
                    if (fast) __hip_atomic_store(rdst, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store(rdst, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (DROP && grp == 0) {
                        typedef __attribute__((address_space(1))) unsigned char gu8;
                        gu8* rm = (gu8*)(reinterpret_cast<unsigned char*>(p.payload + (rb + (unsigned)F_REGION)) + j * (FH / 2) + c * 16 + (ul >> 1));
                        if (fast) __hip_atomic_store(rm, (unsigned char)0xff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else __hip_atomic_store(rm, (unsigned char)0xff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (DROP && grp == 0) {
                    // the mask bits of this lane's unit pair: byte [utterance j][pair (32 c + ul) / 2] of the second payload block
                    typedef __attribute__((address_space(1))) unsigned char gu8;
                    const unsigned char mb = (unsigned char)((st1[0] != 0.f ? 1u : 0u) | (st1[1] != 0.f ? 2u : 0u));
                    gu8* mdst = (gu8*)(reinterpret_cast<unsigned char*>(p.payload + (pbase + (unsigned)F_REGION)) + j * (FH / 2) + c * 16 + (ul >> 1));
                    if (fast) __hip_atomic_store(mdst, mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store(mdst, mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            float* ob = obuf + grp * (6 * OARR) + j * OROW + ul;            // results for the streaming waves
            if constexpr (BF) {                           // bf16 pairs, one word per unit pair
                typedef float f2v __attribute__((ext_vector_type(2)));
                typedef __bf16 b2v __attribute__((ext_vector_type(2)));
                float* o16 = obuf + grp * (6 * OARR) + j * OROW + (ul >> 1);
                const f2v hv = {h.x, h.y}, nv = {hn.x, hn.y};
                o16[0] = __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector(hv, b2v)));
                o16[4 * OARR] = __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector(nv, b2v)));
            } else { st2(ob, h); st2(ob + 4 * OARR, hn); }
            if constexpr (SV16) {                         // 16-bit fixed point, one word per unit pair (rnn_cluster_common.h)
                float* o16 = obuf + grp * (6 * OARR) + j * OROW + (ul >> 1);
                o16[OARR] = __uint_as_float(pack_unorm2(r.x, r.y)); o16[2 * OARR] = __uint_as_float(pack_unorm2(z.x, z.y));
                o16[3 * OARR] = __uint_as_float(pack_snorm2(n.x, n.y));
            } else {
                st2(ob + OARR, r); st2(ob + 2 * OARR, z); st2(ob + 3 * OARR, n);
            }
            if (DROP && grp == 0) {
                if constexpr (BF) {
                    typedef float f2v __attribute__((ext_vector_type(2)));
                    typedef __bf16 b2v __attribute__((ext_vector_type(2)));
                    const f2v dv = {hd.x, hd.y};
                    obuf[j * OROW + (ul >> 1) + 5 * OARR] = __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector(dv, b2v)));
                } else st2(ob + 5 * OARR, hd);
            }
        }
        if (grp < 2 && s == T + 1) break;
        FSTAMP(3);
        FSTAMP(4);
        bar_lds();                                        // groups 0/1: #2 (every publishing wave drained) | group 2: #3
        if (grp == 2) {
            // slot Z of group 2 (groups 0 / 1 are in their MFMAs: nothing latency-critical uses the CU's memory pipeline now, and
            // a CU returns loads in issue order across its waves -- DESIGN 4.1c): the member's HBM streams.  Write-out of step s,
            // then the prefetch of the input projection two steps ahead (issued beside the others' flag polls / gathers
            // instead, the loads queue in front of them: +0.10 ms per forward).  Schedules measured for this group, 2-layer forward
            // incl. the projection GEMM: prefetch, gather, (write-out in slot Y) 1.49 ms; gather, prefetch 1.44; gather, write-out,
            // prefetch all here 1.66; poll + gather + write-out in slot Y 1.58; poll + gather in slot Y, write-out + prefetch
            // here 1.39 (this one).  The group is close to being the critical resource: ~1400 ticks of MFMAs, ~1800 of
            // poll + gather, ~2700 to push 31 KB through the CU's memory pipeline, out of a 6800-tick step.
            flush(tv, s, 2);                              // ... layer 0's (everything outside the steady state) here (obuf is next written in the others' slot Y of step s+1)
            issue_gi(tv, s + 2);
        } else {
            FSTAMP(5);
            FSTAMP(6);
            // gather: h0_s (next layer-0 step; also layer 1's input when there is no dropout) and h1_{s-2} (next layer-1 step)
            const bool need0 = s < T, need2 = s >= 2;    // (h0_{T-1} is still layer 1's last input)
            u32x4 v[2][2];
            unsigned mk[2] = {0u, 0u};                    // DROP: the four mask bits of the piece's units (two bytes: pairs col0 / 2, col0 / 2 + 1)
            for (unsigned spins = 0;; ++spins) {          // until no word is the sentinel
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
                if (rg == 0 ? need0 : need2) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        v[rg][k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (pbase + (unsigned)(rg * 2) * F_REGION + (unsigned)(tv + 512 * k) * 4) * 4, 0, 16 /* sc1 */);
                        if (DROP && rg == 0) {
                            const int i4 = (tv + 512 * k) * 4;
                            mk[k] = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (pbase + (unsigned)F_REGION) * 4 + (unsigned)((i4 >> 8) * (FH / 2) + ((i4 & (FH - 1)) >> 1)), 0, 16 /* sc1 */);
                        }
                    }
                }
            }
            {
                unsigned mx = 0u;                         // max over the requested words: all-ones <=> one of them is still the sentinel
                if (need0) { mx = max(max(max(v[0][0].x, v[0][0].y), max(v[0][0].z, v[0][0].w)), max(max(v[0][1].x, v[0][1].y), max(v[0][1].z, v[0][1].w)));
                             if (DROP && ((mk[0] | mk[1]) & 0xfcfcu)) mx = FX_SENT; }
                if (need2) mx = max(mx, max(max(max(v[1][0].x, v[1][0].y), max(v[1][0].z, v[1][0].w)), max(max(v[1][1].x, v[1][1].y), max(v[1][1].z, v[1][1].w))));
                if (!__any(mx == FX_SENT)) break;
                if (spins > SPIN_LIMIT) { st_agent(p.status, 6); return; }
                if ((spins & 63) == 63 && (ld_agent(p.status) != 0 || (p.soft && ld_agent(p.soft) != 0))) return;
            }
            }
            // The re-arm store of slot (s + 2) % 4 (issued with this step's publish) must be acknowledged before the member's NEXT publish is
            // issued.  The gather loads' own wait already covers it on gfx9 (loads and stores share vmcnt), so this costs nothing; it is spelled
            // out so that the protocol does not hang on the compiler's placement of that wait.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
                if (rg == 0 ? need0 : need2) {
                    unsigned short* hi = rg == 0 ? hs0 : hs1;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int i4 = (tv + 512 * k) * 4;
                        const int o = (i4 >> 8) * FLDHB + (i4 & (FH - 1));
                        const u32x4 x = v[rg][k];
                        uint2 hi2, lo2;
                        hi2.x = (x.x >> 16) | (x.y & 0xffff0000u); hi2.y = (x.z >> 16) | (x.w & 0xffff0000u);
                        lo2.x = (x.x & 0xffffu) | (x.y << 16);      lo2.y = (x.z & 0xffffu) | (x.w << 16);
                        *reinterpret_cast<uint2*>(hi + o) = hi2; *reinterpret_cast<uint2*>(hi + FPLANE + o) = lo2;
                        if (DROP && rg == 0) {
                            // dropout(h0) for group 2: the same planes with the dropped units zeroed (the scale is applied to its sums)
                            const unsigned w = mk[k];
                            const unsigned m01 = ((0u - (w & 1u)) & 0xffffu) | ((0u - ((w >> 1) & 1u)) & 0xffff0000u);
                            const unsigned m23 = ((0u - ((w >> 8) & 1u)) & 0xffffu) | ((0u - ((w >> 9) & 1u)) & 0xffff0000u);
                            uint2 dh, dl;
                            dh.x = hi2.x & m01; dh.y = hi2.y & m23; dl.x = lo2.x & m01; dl.y = lo2.y & m23;
                            *reinterpret_cast<uint2*>(hs0d + o) = dh; *reinterpret_cast<uint2*>(hs0d + FPLANE + o) = dl;
                        }
                    }
                }
            }
        }
        FSTAMP(7);
        bar_lds();                                        // groups 0/1: #3 (planes of step s+1 complete) | group 2: #1 of step s+1
    }
    bar_lds();                                            // the last layer-1 step's results are in obuf
    if (TRACE && trl) { long long* o = p.trace + (tid == 0 ? 0 : (tid == 512 ? 32 : 64)); for (int i = 0; i < 32; ++i) o[i] = trl[i]; }
    if (grp == 2) flush(tid, T + 1, 0);
    {
        const int lane = tid & 63, j = lane & 15, ul = jl * 16 + (lane >> 4) * 4 + 2 * kh;
        const int b = b0t + j, col = c * 32 + ul;
        if (b < p.B) {
            if (grp == 0 && p.hn0) st2(p.hn0 + (size_t)b * FH + col, f2(st0[0], st0[1]));
            if (grp == 1) {
                if (p.pooled) st2(p.pooled + (size_t)b * FH + col, f2(st0[2] * p.pool_scale, st0[3] * p.pool_scale));
                if (p.hn1) st2(p.hn1 + (size_t)b * FH + col, f2(st0[0], st0[1]));
            }
        }
    }
}


// =====================================================================================================================================
}  // namespace

bool dep_fused2_ok(int cell, int H, int L, int dirs) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("DEP_FUSED2"); off = (e && e[0] == '0') ? 1 : 0; }
    return !off && cell == DEP_CELL_GRU && H == FH && L == 2 && dirs == 1;
}

size_t dep_fused2_xbuf_bytes(int B) {
    const int CH = dep_cluster_chunk(FNC, 1, 256);
    const int nbtp = (dep_cdiv(B < CH ? B : CH, BT) + 7) / 8 * 8;
    return PAYLOAD_OFF + (size_t)4 * nbtp * 3 * F_REGION * sizeof(float) + 4096;      // four sentinel-armed slots
}

int dep_launch_fused2_fwd(const dep_fused2_args& a, void* xbuf, size_t xbuf_bytes) {
    const int CH = dep_cluster_chunk(FNC, 1, 256);
    const int nbtp_max = (dep_cdiv(a.B < CH ? a.B : CH, BT) + 7) / 8 * 8;
    FF p{};
    p.B = a.B; p.T = a.T;
    p.wp0 = (const u32x4*)a.wp0; p.wp1 = (const u32x4*)a.wp1; p.wpi = (const u32x4*)a.wpi;
    p.b_hh0 = a.b_hh0; p.b_ih1 = a.b_ih1; p.b_hh1 = a.b_hh1;
    p.gi = a.gi; p.ldgi = 3 * FH;
    const bool drop = a.drop_p > 0.f;
    p.y0 = a.y0; p.y1 = a.y1; p.ostride = (unsigned)a.ostride; p.training = a.training;
    p.drop_p = a.drop_p; p.drop_scale = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f; p.seed = a.seed; p.site = a.site;
    p.pooled = a.pooled; p.pool_scale = a.pool_scale; p.hn0 = a.hn0; p.hn1 = a.hn1;
    // the kernel addresses every per-layer output array as y_l + k * ostride: check the caller's layout really is that
    DEP_CHECK_ARG(!drop || a.y0d == a.y0 + a.ostride);
    if (a.training) for (int k = 0; k < 4; ++k)
        DEP_CHECK_ARG(a.sv[0][k] == a.y0 + (size_t)(k + (drop ? 2 : 1)) * a.ostride && a.sv[1][k] == a.y1 + (size_t)(k + 1) * a.ostride);
    const size_t pay = (size_t)4 * nbtp_max * 3 * F_REGION * sizeof(float);      // four sentinel-armed slots
    DEP_CHECK_ARG(xbuf && PAYLOAD_OFF + pay <= xbuf_bytes && (size_t)nbtp_max * FNC <= 256);
    DEP_CHECK_ARG(!drop || a.y0d);
    p.status = (unsigned*)xbuf; p.flags = (unsigned*)(hdr_base(xbuf, 0) + FLAG_OFF); p.hello = (unsigned*)(hdr_base(xbuf, 0) + HELLO_OFF);
    p.payload = (float*)((char*)xbuf + PAYLOAD_OFF); p.payload_bytes = (unsigned)pay; p.nofast = nofast_env();
    p.trace = trace_env() ? (long long*)(hdr_base(xbuf, 0) + TRACE_OFF) : nullptr;
    p.soft = a.soft_fallback ? (unsigned*)xbuf + 1 : nullptr;
    const bool sv16 = a.training && a.sv16;
    const bool bf = a.training && a.bf16st;
    DEP_CHECK_ARG(!bf || sv16);
    p.ntstore = 0;                                    // (non-temporal write-out measured: no effect on this launch -- its payload, 0.4 MB per XCD, survives anyway)
    { static int fs = -1; if (fs < 0) { const char* e = getenv("DEP_FORCE_SOFT_FALLBACK"); fs = (e && e[0] >= '1' && e[0] <= '3') ? e[0] - '0' : 0; } p.force_soft = fs; }
    static bool attr = false;
    if (!attr) {
#define F2_ATTR(D, TR, X) (void)hipFuncSetAttribute((const void*)gru2_fwd_fused<D, TR, X, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES)
        (void)hipFuncSetAttribute((const void*)gru2_fwd_fused<true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gru2_fwd_fused<false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES);
        F2_ATTR(true, false, false); F2_ATTR(false, false, false); F2_ATTR(true, true, false); F2_ATTR(false, true, false);
        F2_ATTR(true, false, true); F2_ATTR(false, false, true); F2_ATTR(true, true, true); F2_ATTR(false, true, true);
#undef F2_ATTR
        attr = true;
    }
    DepProfScope prof(DEP_PROF_GRU_FWD, a.stream);
    for (int b0 = 0; b0 < a.B; b0 += CH) {
        const int cb = a.B - b0 < CH ? a.B - b0 : CH;
        p.b0 = b0; p.nbtp = (dep_cdiv(cb, BT) + 7) / 8 * 8;
        // flags / hello words only: the status word is sticky over every sweep of a step (cleared by dep_rnn_forward)
        { const int rc_h = hdr_prepare(xbuf, 0, a.hdr_clean && b0 == 0, a.stream); if (rc_h) return rc_h; }
        const dim3 grid(FNC * p.nbtp), blk(FTHREADS);
#define F2_LAUNCH(D, TR) do { if (sv16) DEP_LAUNCH((gru2_fwd_fused<D, TR, true>), grid, blk, F_LDS_BYTES, a.stream, p); \
                              else DEP_LAUNCH((gru2_fwd_fused<D, TR, false>), grid, blk, F_LDS_BYTES, a.stream, p); } while (0)
        if (bf) {                                         // bf16-storage mode (never traced)
            if (drop) DEP_LAUNCH((gru2_fwd_fused<true, false, true, true>), grid, blk, F_LDS_BYTES, a.stream, p);
            else DEP_LAUNCH((gru2_fwd_fused<false, false, true, true>), grid, blk, F_LDS_BYTES, a.stream, p);
        } else if (p.trace) {                             // DEP_TRACE=1: the stamped variant (tools/trace_fused.py)
            if (drop) F2_LAUNCH(true, true); else F2_LAUNCH(false, true);
        } else {
            if (drop) F2_LAUNCH(true, false); else F2_LAUNCH(false, false);
        }
#undef F2_LAUNCH
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}
