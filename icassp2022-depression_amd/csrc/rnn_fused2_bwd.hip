// Two-layer GRU BPTT as ONE cluster-parallel launch (the backward twin of rnn_fused2.hip): layer 0 runs one step behind
// layer 1, so the backward pass costs T + 1 dependent hand-offs instead of 2 T, and the gradient that flows from layer 1
// into layer 0 (dX of layer 1: B x T x H, the former NN GEMM and its HBM round trip) never exists in HBM.
//
// Per 16-utterance tile a cluster of NC = 8 workgroups (one per CU, co-resident); member c owns hidden units [32c, 32c+32) of
// both layers.  Twelve waves, three groups of four, each keeping the TRANSPOSED weight slice of the member's 96 gate rows
// register-resident (96 VGPRs, (hi, lo) bf16 planes; wave = 4 of the 16 output column tiles, K = 96 = three 32-wide k-steps):
//   group 0 : W_hh(l1)^T x [dr, dz, dn*r](l1, t)     -> partial dh1_{t-1}            (16 x 256, all columns)
//   group 1 : W_ih(l1)^T x [dr, dz, dn  ](l1, t)     -> partial d(dropout(y0))_t     (what used to be dX of layer 1)
//   group 2 : W_hh(l0)^T x [dr, dz, dn*r](l0, t+1)   -> partial dh0_t
// Fused step u (0..T; layer 1 at t = T-1-u, layer 0 at t+1): [gate gradients of both layers on groups 0 / 2 (2 elements
// per thread), written as bf16 planes] -> barrier -> [36 MFMAs per wave] -> the three partial blocks are published in
// MFMA-fragment order (16-byte stores), drain, barrier, ONE flag per member -> [groups 0 / 2: poll the 8 flags, gather the own
// 32 columns of every member's partials (reduce-scatter, summed in member order: deterministic) | group 1: owns every HBM
// stream of the member: saved gates / h_{t-1} / dy of the next steps into LDS, dgi / dghn of this step out] -> barrier.
// Exchange protocol, same-XCD fast path, parity double-buffered payload, bounded spins, sticky status: rnn_cluster_common.h.
#include "rnn_cluster_common.h"

namespace {
using namespace depc;

constexpr int BH = 256, BNC = 8, BTHREADS = 768;
constexpr int LDGB = 128 + 8;                 // bf16 elements per row of a gate-gradient plane: [dr | dz | dn*r | dn] + pad (272-byte rows)
constexpr int GPLANE = BT * LDGB;             // one plane (hi or lo)
constexpr int OROW = 36, OARR = BT * OROW;    // fp32 [16 utterances][32 units] arrays, rows padded (bank conflicts)
constexpr int N_IBUF = 11, N_OBUF = 8;       // input arrays: l1 r,z,n,hn,h_{t-1},[dy] ; l0 r,z,n,hn,h_{t-1}  (10 without dy: five per wave half)
constexpr int B_BLOCK = BNC * BT * BH;        // floats of one partial block of one tile: [src member][out tile][lane][4]
constexpr int N_PBLK = 4 * 4 * 256;           // group 1's partial block on its way to the publishing waves: [wave][tile][lane][4]
constexpr size_t B_LDS_BYTES = (size_t)(2 * GPLANE + (2 * N_IBUF + 2 * N_OBUF) * OARR + N_PBLK + 64) * sizeof(float);   // 2 layers x (hi+lo) planes = 4 GPLANE bf16 = 2 GPLANE floats

struct FB {
    int B, T, nbtp, b0;
    const u32x4* wh1; const u32x4* wi1; const u32x4* wh0;          // transposed-slice images (pack_cluster_bwd_split format)
    const float* y1; const float* y0;                               // forward hidden sequences (h_{t-1})
    const float* sv1; const float* sv0; unsigned svstride;          // saved r | z | n | hn, svstride floats apart
    const float* dy; const float* dpooled; float pool_scale; const float* dhn1; const float* dhn0;
    float drop_p, drop_scale; uint64_t seed; uint32_t site;
    float* dgi1; float* dghn1; float* dgi0; float* dghn0;           // (B*T, 3H) / (B*T, H)
    float* dbpart1; float* dbpart0;                                 // [batch tile][4][H] bias-gradient partials
    unsigned* status; unsigned* flags; unsigned* hello; float* payload; unsigned payload_bytes; int nofast;
};

// Registers: 3 waves per SIMD -> 168 VGPRs, 96 of them weights.  Per-thread indices are re-derived every step from a laundered
// threadIdx.x (see rnn_fused2.hip), and the per-role persistent state shares six vector registers:
//   groups 0 / 2: st[0] = (dh_rec.xy, dpool.xy)  st[1] = (db_r.xy, db_z.xy)  st[2] = (db_n.xy, db_hn.xy)
//                 st[3] = layer 0: (dy0.xy = masked gradient from layer 1, mask.xy) ; layer 1: (d*z .xy, -, -)
//   group 1     : st[0..4] (st[0..5] with an external dy) = the prefetched input pieces of the step after next
template <bool DROP, bool HASDY>
__global__ __launch_bounds__(BTHREADS) void gru2_bwd_fused(FB p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T;
    const int c = blockIdx.x / p.nbtp, bt = blockIdx.x % p.nbtp;
    if (p.b0 + bt * BT >= p.B) return;
    if (ld_agent(p.status) != 0) return;           // an earlier sweep of this step gave up (sticky status)
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w >> 2, gw = w & 3;               // 0: layer-1 recurrent, 1: layer-1 input + HBM streams, 2: layer-0 recurrent
    const int shalf = gw >> 1;
    unsigned short* dg1 = reinterpret_cast<unsigned short*>(smem);      // layer-1 planes: hi at +0, lo at +GPLANE
    unsigned short* dg0 = dg1 + 2 * GPLANE;
    float* ibuf = smem + 2 * GPLANE;                  // [2 step parities][11][16][36]: l1 r,z,n,hn,hp,dy ; l0 r,z,n,hn,hp
    float* obuf = ibuf + 2 * N_IBUF * OARR;               // [2 step parities][8][16][36] : l1 dr,dz,dn,dn*r ; l0 dr,dz,dn,dn*r
    float* pblk = obuf + 2 * N_OBUF * OARR;
    const int b0t = p.b0 + bt * BT;

    u32x4 wq[4][3][2];                                // [out tile of the wave][k-step = gate][hi, lo]
    {
        const u32x4* wimg = grp == 0 ? p.wh1 : (grp == 1 ? p.wi1 : p.wh0);
        const int lane = tid & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    wq[i][ks][pl] = wimg[(size_t)(((c * 16 + gw * 4 + i) * 3 + ks) * 2 + pl) * 64 + lane];
    }
    f32x4 st[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) st[i] = zero4();

    const unsigned pstride = (unsigned)p.nbtp * 3 * B_BLOCK;
    const unsigned tile_base = (unsigned)bt * 3 * B_BLOCK;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.payload, 0, p.payload_bytes, 0x00020000);
    unsigned* myflag = p.flags + bt * BNC + c;
    unsigned* tflags = p.flags + bt * BNC;
    const int sx = p.nofast ? 0 : cluster_same_xcd(p.hello + bt * BNC, BNC, c, p.status);
    if (sx < 0) return;
    const bool fast = sx == 1;

    // ---- group 1's streams.  Thread -> (utterance su, 16-byte piece sqd of the member's 32 units); wave half -> array of a pair.
    // Inputs of fused step uu: layer 1 at t1 = T-1-uu (r, z, n, hn, h_{t1-1}, dy), layer 0 at t0 = T-uu (r, z, n, hn, h_{t0-1}).
    auto stage = [&](int tv, int uu) {
        const int rem = tv & 127, su = rem >> 3, sqd = rem & 7;
        const bool uv = b0t + su < p.B;
        const unsigned ro = ((unsigned)(b0t + su) * T) * BH + c * 32 + sqd * 4;
        const int t1 = T - 1 - uu, t0 = T - uu;
        const bool a1 = uv && t1 >= 0, a0 = uv && uu >= 1 && t0 >= 0;
        constexpr int N1 = HASDY ? 6 : 5, NA = N1 + 5, NP = (NA + 1) / 2;    // arrays of layer 1, all arrays, pairs
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            const int a = pr * 2 + shalf;                 // wave-uniform array id
            const bool l1 = a < N1;
            const int k = l1 ? a : a - N1;                // 0..3 saved gates, 4 h_{t-1}, 5 dy (layer 1 only)
            const int t = l1 ? t1 : t0;
            bool on = (l1 ? a1 : a0) && a < NA;
            const float* src;
            if (k < 4) src = (l1 ? p.sv1 : p.sv0) + (size_t)k * p.svstride + (ro + (unsigned)t * BH);
            else if (k == 4) { src = (l1 ? p.y1 : p.y0) + (ro + (unsigned)(t - 1) * BH); on = on && t >= 1; }
            else { src = p.dy + (ro + (unsigned)t * BH); on = on && HASDY; }
            st[pr] = on ? ld4(src) : zero4();
        }
    };
    auto put_ibuf = [&](int tv, int uu) {             // staged inputs of fused step uu -> ibuf[uu & 1]
        const int rem = tv & 127, ro = (uu & 1) * N_IBUF * OARR + (rem >> 3) * OROW + (rem & 7) * 4;
        constexpr int NA = (HASDY ? 6 : 5) + 5, NP = (NA + 1) / 2;
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            const int a = pr * 2 + shalf;
            if (a < NA) *reinterpret_cast<f32x4*>(ibuf + a * OARR + ro) = st[pr];
        }
    };
    auto flush = [&](int tv, int u) {                 // gate gradients of fused step u: LDS -> HBM
        const int rem = tv & 127, su = rem >> 3, sqd = rem & 7;
        if (b0t + su >= p.B) return;
        const unsigned row0 = (unsigned)(b0t + su) * T;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int a = pr * 2 + shalf;                 // 0..3 layer 1 (dr, dz, dn, dn*r), 4..7 layer 0
            const bool l1 = a < 4;
            const int k = a & 3;
            const int t = l1 ? T - 1 - u : T - u;
            const bool on = l1 ? (u <= T - 1) : (u >= 1);
            float* base = k < 3 ? (l1 ? p.dgi1 : p.dgi0) : (l1 ? p.dghn1 : p.dghn0);
            const size_t off = k < 3 ? (size_t)(row0 + t) * (3 * BH) + k * BH : (size_t)(row0 + t) * BH;
            if (on) *reinterpret_cast<f32x4*>(base + off + c * 32 + sqd * 4) = ld4(obuf + ((u & 1) * N_OBUF + a) * OARR + su * OROW + sqd * 4);
        }
    };
    // inter-layer dropout on the gradient entering layer 0: same Philox draw as the forward's mask of y0 at (b, t, col)
    auto draw = [&](int tv, int t) {
        const int lt = tv & 255, lp = (lt >> 1) & 63, half = lt & 1;
        const int col = c * 32 + (lt >> 7) * 16 + (lp >> 4) * 4 + 2 * half;
        const size_t o = ((size_t)(b0t + (lp & 15)) * T + t) * BH + col;
        const f32x4 m = dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
        return f2(half ? m[2] : m[0], half ? m[3] : m[1]);
    };
    {   // initial recurrent gradient (dh_n) and the pooling gradient of the top layer
        const int lt = tid & 255, lp = (lt >> 1) & 63, half = lt & 1;
        const int b = b0t + (lp & 15), col = c * 32 + (lt >> 7) * 16 + (lp >> 4) * 4 + 2 * half;
        if (grp != 1 && b < p.B) {
            const float* dhn = grp == 0 ? p.dhn1 : p.dhn0;
            if (dhn) { const float2 v = ld2(dhn + (size_t)b * BH + col); st[0][0] = v.x; st[0][1] = v.y; }
            if (grp == 0 && p.dpooled) { const float2 v = ld2(p.dpooled + (size_t)b * BH + col); st[0][2] = v.x * p.pool_scale; st[0][3] = v.y * p.pool_scale; }
        }
        if (grp == 2) { st[3][2] = 1.f; st[3][3] = 1.f; }
    }
    if (grp == 1) { stage(tid, 0); put_ibuf(tid, 0); stage(tid, 1); }
    __syncthreads();

    for (int u = 0; u <= T; ++u) {
        int tv = tid;
        asm volatile("" : "+v"(tv));                  // launder: indices derived from tv are recomputed per step, not hoisted
        const int lane = tv & 63;
        const int lt = tv & 255, jl = lt >> 7, lp = (lt >> 1) & 63, half = lt & 1;
        const int j = lp & 15, ul = jl * 16 + (lp >> 4) * 4 + 2 * half;       // utterance row, unit pair (ul, ul+1) of the member's 32
        const bool act = grp == 2 ? (u >= 1) : (u <= T - 1);
        // ---- group 1's HBM streams go HERE, while nothing latency-critical uses the CU's memory pipeline (the others are in
        // their gate math, then everybody in the MFMAs): the next step's inputs first, then the previous step's write-out (from
        // the other obuf parity).  36 KB per step and CU is ~3400 cycles of the CU's ~10.7 B/clk share of HBM: issued beside
        // the payload stores / flag polls / gather loads instead, these streams queue in front of them.  (Measured: without the
        // streams this launch takes 1.07 ms, with the loads alone 1.47, the stores alone 1.31, both 1.9-2.1 -- wherever they are
        // issued -- against 1.64 + 0.25 ms for the two per-layer sweeps plus the dX GEMM this launch removes: not adopted, opt-in.)
        if (grp == 1) {
            put_ibuf(tv, u + 1);                          // loads issued at the top of the PREVIOUS step: a whole step to land
            stage(tv, u + 2);
            if (u >= 1) flush(tv, u - 1);
        }
        // ---- gate gradients (groups 0 and 2; identical code, role-dependent LDS bases)
        if (grp != 1 && act) {
            const float* ib = ibuf + ((u & 1) * N_IBUF + (grp == 0 ? 0 : (HASDY ? 6 : 5))) * OARR + j * OROW + ul;
            const float2 r = ld2(ib), z = ld2(ib + OARR), n = ld2(ib + 2 * OARR), hn = ld2(ib + 3 * OARR), hp = ld2(ib + 4 * OARR);
            float2 dyv = f2(0.f, 0.f);
            if (grp == 2) dyv = f2(st[3][0], st[3][1]);   // layer 0: masked gradient from layer 1
            if (HASDY && grp == 0) dyv = ld2(ib + 5 * OARR);
            const float2 d = f2(st[0][0] + st[0][2] + dyv.x, st[0][1] + st[0][3] + dyv.y);
            float2 dn, dz, dr, dnr;
            dn.x = d.x * (1.0f - z.x) * (1.0f - n.x * n.x); dn.y = d.y * (1.0f - z.y) * (1.0f - n.y * n.y);
            dz.x = d.x * (hp.x - n.x) * z.x * (1.0f - z.x); dz.y = d.y * (hp.y - n.y) * z.y * (1.0f - z.y);
            dr.x = dn.x * hn.x * r.x * (1.0f - r.x); dr.y = dn.y * hn.y * r.y * (1.0f - r.y);
            dnr.x = dn.x * r.x; dnr.y = dn.y * r.y;
            // d*z (the part of dh_{t-1} that bypasses the gates) waits for the gathered sum: layer 1 parks it in st[3].xy (unused
            // there), layer 0 in st[0].zw (its pooling-gradient slot, which must read zero again at the next gate phase)
            if (grp == 0) { st[3][0] = d.x * z.x; st[3][1] = d.y * z.y; } else { st[0][2] = d.x * z.x; st[0][3] = d.y * z.y; }
            unsigned short* gh = (grp == 0 ? dg1 : dg0) + j * LDGB + ul;
            unsigned h0, l0, h1, l1, h2, l2, h3, l3;
            split_pair(dr.x, dr.y, h0, l0); split_pair(dz.x, dz.y, h1, l1); split_pair(dnr.x, dnr.y, h2, l2); split_pair(dn.x, dn.y, h3, l3);
            *reinterpret_cast<unsigned*>(gh) = h0; *reinterpret_cast<unsigned*>(gh + GPLANE) = l0;
            *reinterpret_cast<unsigned*>(gh + 32) = h1; *reinterpret_cast<unsigned*>(gh + GPLANE + 32) = l1;
            *reinterpret_cast<unsigned*>(gh + 64) = h2; *reinterpret_cast<unsigned*>(gh + GPLANE + 64) = l2;
            if (grp == 0) { *reinterpret_cast<unsigned*>(gh + 96) = h3; *reinterpret_cast<unsigned*>(gh + GPLANE + 96) = l3; }
            float* ob = obuf + ((u & 1) * N_OBUF + (grp == 0 ? 0 : 4)) * OARR + j * OROW + ul;
            st2(ob, dr); st2(ob + OARR, dz); st2(ob + 2 * OARR, dn); st2(ob + 3 * OARR, dnr);
            st[1][0] += dr.x; st[1][1] += dr.y; st[1][2] += dz.x; st[1][3] += dz.y;
            st[2][0] += dn.x; st[2][1] += dn.y; st[2][2] += dnr.x; st[2][3] += dnr.y;
        }
        bar_lds();                                        // #1: gate-gradient planes and the write-out arrays are in LDS
        if (u == T) break;                                // layer 0's last step (t = 0) has no predecessor to publish for
        // ---- partial products on the matrix cores, published in fragment order
        const unsigned pbase = (unsigned)(u & 1) * pstride + tile_base;
        if (grp == 2 ? (u >= 1) : true) {
            const unsigned short* gsrc = (grp == 2 ? dg0 : dg1) + (lane & 15) * LDGB + (lane >> 4) * 8;
            const int k2 = grp == 1 ? 96 : 64;            // third k-step: dn for the input path, dn*r for the recurrent one
            f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int ko = ks == 2 ? k2 : ks * 32;
                const bf16x8 gh = *reinterpret_cast<const bf16x8*>(gsrc + ko);
                const bf16x8 gl = *reinterpret_cast<const bf16x8*>(gsrc + GPLANE + ko);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[i][ks][0]), wl = __builtin_bit_cast(bf16x8, wq[i][ks][1]);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gl, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, gh, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gh, acc[i], 0, 0, 0);
                }
            }
            auto publish = [&](const f32x4& a, unsigned fo) {
                u32x4 v;
                v.x = __float_as_uint(a[0]); v.y = __float_as_uint(a[1]); v.z = __float_as_uint(a[2]); v.w = __float_as_uint(a[3]);
                if (fast) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, fo * 4, 0, 0 /* plain: stays in this XCD's L2 */);
                else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, fo * 4, 0, 16 /* sc1: write-through */);
            };
            if (grp == 1) {
                // group 1 never stores to the payload itself: a publishing wave must wait for its stores' acknowledgement
                // (vmcnt, which counts its HBM streams too).  Its block goes through LDS to the waves of groups 0 / 2.
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(pblk + ((gw * 4 + i) * 64 + lane) * 4) = acc[i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) publish(acc[i], pbase + (unsigned)grp * B_BLOCK + (unsigned)((c * 16 + gw * 4 + i) * 64 + lane) * 4);
            }
        }
        bar_lds();                                        // #1b: group 1's block is in LDS
        if (grp != 1) {
            // group 0 forwards tiles 0, 1 of each of group 1's waves, group 2 tiles 2, 3
            const int i0 = grp == 0 ? 0 : 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 a = ld4(pblk + ((gw * 4 + i0 + i) * 64 + lane) * 4);
                u32x4 v;
                v.x = __float_as_uint(a[0]); v.y = __float_as_uint(a[1]); v.z = __float_as_uint(a[2]); v.w = __float_as_uint(a[3]);
                const unsigned fo = pbase + (unsigned)B_BLOCK + (unsigned)((c * 16 + gw * 4 + i0 + i) * 64 + lane) * 4;
                if (fast) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, fo * 4, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, fo * 4, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // payload stores acknowledged
        }
        bar_lds();                                        // #2
        if (tid == 0) { if (fast) st_local(myflag, (unsigned)u + 1u); else st_agent(myflag, (unsigned)u + 1u); }
        if (grp != 1) {
            // next step's dropout mask for layer 0, in the shadow of the wait
            if (DROP && grp == 2 && u + 1 <= T) { const float2 m = draw(tv, T - (u + 1)); st[3][2] = m.x; st[3][3] = m.y; }
            if (!wait_flags(tflags, BNC, (unsigned)u + 1u, p.status, 7)) return;
            // reduce-scatter: this thread's two columns of every member's partial, summed in member order
            const float* src = p.payload + pbase + ((unsigned)(2 * c + jl) * 64 + lp) * 4 + 2 * half;
            if (grp == 0) {                               // dh1_{t-1}
                if (u + 1 <= T - 1) {
                    float2 part[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) part[m] = ld2_agent(src + (size_t)m * (16 * 256));
                    float2 s = f2(0.f, 0.f);
#pragma unroll
                    for (int m = 0; m < 8; ++m) { s.x += part[m].x; s.y += part[m].y; }
                    st[0][0] = st[3][0] + s.x; st[0][1] = st[3][1] + s.y;
                }
            } else {                                      // layer 0 of the next step: dy0 (from group 1's block) and dh0
                // two rounds of eight loads (registers): first the recurrent partials -- the longer dependency chain --
                float2 part[8];
                if (u >= 1) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) part[m] = ld2_agent(src + 2 * B_BLOCK + (size_t)m * (16 * 256));
                    float2 q = f2(0.f, 0.f);
#pragma unroll
                    for (int m = 0; m < 8; ++m) { q.x += part[m].x; q.y += part[m].y; }
                    st[0][0] = st[0][2] + q.x; st[0][1] = st[0][3] + q.y; st[0][2] = 0.f; st[0][3] = 0.f;
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) part[m] = ld2_agent(src + B_BLOCK + (size_t)m * (16 * 256));
                float2 s = f2(0.f, 0.f);
#pragma unroll
                for (int m = 0; m < 8; ++m) { s.x += part[m].x; s.y += part[m].y; }
                st[3][0] = s.x * st[3][2]; st[3][1] = s.y * st[3][3];
            }
        }
        bar_lds();                                        // #3: next step's inputs are in LDS, this step's write-out left it
    }
    if (grp == 1) flush(tid, T);                          // layer 0's last step
    if (grp != 1) {
        // bias-gradient partials [batch tile][4][H]: sum over the 16 utterance rows = lanes that differ in bits 1..4
        float2 a[4] = {f2(st[1][0], st[1][1]), f2(st[1][2], st[1][3]), f2(st[2][0], st[2][1]), f2(st[2][2], st[2][3])};
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

size_t dep_fused2_bwd_xbuf_bytes(int B) {
    const int CH = dep_cluster_chunk(BNC, 1, 256);
    const int nbtp = (dep_cdiv(B < CH ? B : CH, BT) + 7) / 8 * 8;
    return PAYLOAD_OFF + (size_t)2 * nbtp * 3 * B_BLOCK * sizeof(float) + 4096;
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
    DEP_CHECK_ARG(a.dbpart_rows >= nbt && a.wh1 && a.wi1 && a.wh0 && a.dgi1 && a.dgi0 && a.dghn1 && a.dghn0);
    const size_t pay = (size_t)2 * nbtp_max * 3 * B_BLOCK * sizeof(float);
    DEP_CHECK_ARG(xbuf && PAYLOAD_OFF + pay <= xbuf_bytes && (size_t)nbtp_max * BNC <= 256 && pay < (1ull << 32));
    p.status = (unsigned*)xbuf; p.flags = (unsigned*)(hdr_base(xbuf, 0) + FLAG_OFF); p.hello = (unsigned*)(hdr_base(xbuf, 0) + HELLO_OFF);
    p.payload = (float*)((char*)xbuf + PAYLOAD_OFF); p.payload_bytes = (unsigned)pay; p.nofast = nofast_env();
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gru2_bwd_fused<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gru2_bwd_fused<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gru2_bwd_fused<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gru2_bwd_fused<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B_LDS_BYTES);
        attr = true;
    }
    DepProfScope prof(DEP_PROF_GRU_BWD, a.stream);
    for (int b0 = 0; b0 < a.B; b0 += CH) {
        const int cb = a.B - b0 < CH ? a.B - b0 : CH;
        p.b0 = b0; p.nbtp = (dep_cdiv(cb, BT) + 7) / 8 * 8;
        // flags / hello words only: the status word is sticky over every sweep of a step (cleared by dep_rnn_forward)
        { const int rc_h = hdr_prepare(xbuf, 0, false, a.stream); if (rc_h) return rc_h; }
        const dim3 grid(BNC * p.nbtp), blk(BTHREADS);
        if (drop) { if (a.dy) hipLaunchKernelGGL((gru2_bwd_fused<true, true>), grid, blk, B_LDS_BYTES, a.stream, p);
                    else hipLaunchKernelGGL((gru2_bwd_fused<true, false>), grid, blk, B_LDS_BYTES, a.stream, p); }
        else      { if (a.dy) hipLaunchKernelGGL((gru2_bwd_fused<false, true>), grid, blk, B_LDS_BYTES, a.stream, p);
                    else hipLaunchKernelGGL((gru2_bwd_fused<false, false>), grid, blk, B_LDS_BYTES, a.stream, p); }
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}
