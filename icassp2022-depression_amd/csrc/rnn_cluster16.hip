// Cluster-parallel GRU sweeps, 16 hidden units per member (NC = H/16 members per 16-utterance tile).
//
// With 32 units per member (rnn_cluster_bwd.hip) a B = 512 batch gives 256 workgroups, one per CU, and every CU
// idles through the exchange latency of its step (~55 % of the step).  Halving the member doubles the workgroup
// count to 512 = two per CU from DIFFERENT tiles: while one waits for its cluster's flags the other runs its MFMAs,
// so the hardware overlaps exchange latency and compute without any software pipelining.  Per member and step:
//   forward : 3 gate tiles x K=H on 4 waves (each a K quarter, W in 48 VGPRs at H = 256), 4-way reduction in LDS,
//             one (utterance, unit) element per thread, 1 KB published, 16xH block read back;
//   backward: K = 48 (the member's 3x16 gate rows) x all H output columns (4 tiles per wave), partial dh published
//             in fragment order (16 KB), each thread sums its column over the NC partials in member order.
// Exchange protocol, same-XCD fast path, parity double-buffering, bounded spins: identical to rnn_cluster_bwd.hip.
#include "rnn_cluster_common.h"

namespace {
using namespace depc;

// debug stamps go to LDS and are copied out after the sweep: a global store per stamp would sit in vmcnt and distort the
// very waits being measured
constexpr int RED_BLK = 4 * 80;      // floats per (wave, gate) partial-sum fragment in LDS, see gru_fwd_cluster16
#define DEP_STAMP(slot) do { if (tr && t >= 100 && t < 104) trl[(t - 100) * 8 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

struct F16 {
    int B, T, H, nbtp, b0;      // b0: first utterance of this launch's batch chunk (B is the whole batch)
    const f32x4* wp; const float* b_hh;
    const float* gi; int ldgi;
    float* y; int ldy;
    float* ydrop; float drop_p, drop_scale; uint64_t seed; uint32_t site;
    float* pooled; float pool_scale;
    float* h_n;
    float* sv0; float* sv1; float* sv2; float* sv3;
    unsigned* status; unsigned* flags; unsigned* hello; float* payload; unsigned payload_bytes;
    int nofast;
    long long* trace;
};

// =============================================================================== forward
// matvec roles : lane = (utterance j = lane&15, k-quad q = lane>>4), wave w = K quarter
// finalise roles: thread = (utterance fj = tid>>4, unit fu = tid&15)  -> 64-byte row segments in every global access
//
// SPLIT: the recurrent product h_{t-1} W_hh^T runs on the bf16 matrix cores with the 3-term split of gemm_bf16x3.hip
// (w_hi h_lo + w_lo h_hi + w_hi h_hi, fp32 accumulate): v_mfma_f32_16x16x32_bf16 covers 32 k per 16 cycles against 4 k
// per 32 cycles for v_mfma_f32_16x16x4_f32, so a step's 48 fp32 MFMAs (1536 cycles, ~a quarter of the step) become 18
// (288 cycles).  W_hh stays register-resident as (hi, lo) bf16 planes (same 48 VGPRs); h travels between the members
// already split -- one 32-bit word (hi << 16 | lo) per value, made once by the thread that owns the value -- and is kept
// in LDS as two bf16 planes.  Everything elementwise (gates, h, saved state, pooled sum) stays fp32.
//
// A fifth wave owns every HBM stream of the member.  One step ahead it fetches the member's slice of the input projection
// (3 gates x 16 utterances x 16 units per step) into a parity-double-buffered LDS block; one step behind it writes the
// step's results (h, dropped h, saved r / z / n / hn), which the compute waves only deposit in LDS, with 16-byte stores
// (and draws the dropout mask once per 4 values).  The four compute waves therefore never have an HBM access outstanding:
// the s_waitcnt vmcnt(0) in front of the flag publication waits for the payload store alone, and the flag polls / gather
// loads (VMEM returns in order) never queue behind an HBM load or a write acknowledgement.
template <int KCQ, bool SPLIT>      // k-chunks of 16 per wave = H/64
// launch bounds: two 5-wave workgroups must fit on a CU wherever their wave rotations start, i.e. 4 waves on one SIMD:
// <= 128 VGPRs.  (At 136 the second workgroup of a CU could not always be placed, part of the cluster members never became
// resident and every step ran into the spin limit: 395 ms instead of 2.)
__global__ __launch_bounds__(CT + 64, 4) void gru_fwd_cluster16(F16 p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, LDH = H + LPAD, KC = H / 16, NC = H / 16;
    const int LDHB = H + 8;                           // bf16 elements per row of a split plane (528-byte rows: conflict-free b128 reads)
    const int c = blockIdx.x / p.nbtp, bt = blockIdx.x % p.nbtp;
    if (p.b0 + bt * BT >= p.B) return;
    if (ld_agent(p.status) != 0) return;           // an earlier sweep of this step gave up: the status word is sticky until the next dep_rnn_forward
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int fj = tid >> 4, fu = tid & 15;
    const int col = c * 16 + fu;
    const int b = p.b0 + bt * BT + fj;
    const bool valid = b < p.B;
    float* hs = smem;                                 // [16][LDH] fp32, or (SPLIT) two bf16 planes [16][LDHB]
    const int hs_floats = SPLIT ? BT * LDHB : BT * LDH;      // 2 planes x 2 bytes == 4 bytes per element
    unsigned short* hs_hi = reinterpret_cast<unsigned short*>(smem);
    unsigned short* hs_lo = hs_hi + BT * LDHB;
    float* red = smem + hs_floats;                    // [4 waves][3 gates][64 lanes][4]
    float* gbuf = red + 4 * 3 * RED_BLK;              // [2 parities][3 gates][16 utterances][16 units]
    float* obuf = gbuf + 2 * 768;                     // [2 parities][h, r, z, n, hn][16 utterances][16 units]
    for (int i = tid; i < hs_floats; i += CT + 64) hs[i] = 0.f;

    constexpr int KS2 = KCQ / 2;                      // 32-wide k-steps per wave (SPLIT)
    f32x4 wr[SPLIT ? 1 : 3][SPLIT ? 1 : KCQ];
    u32x4 wq[SPLIT ? 3 : 1][SPLIT ? KS2 : 1][2];      // [gate][k-step][hi, lo]
    if constexpr (SPLIT) {
        const u32x4* wpq = reinterpret_cast<const u32x4*>(p.wp);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    wq[g][ks][pl] = wpq[(size_t)((((c * 3 + g) * 4 + (w & 3)) * KS2 + ks) * 2 + pl) * 64 + lane];
    } else {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k = 0; k < KCQ; ++k)
                wr[g][k] = p.wp[(size_t)((c * 3 + g) * KC + (w & 3) * KCQ + k) * 64 + lane];
    }
    float bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = p.b_hh[g * H + col];
    float hprev = 0.f, pool = 0.f;
    const size_t pstride = (size_t)p.nbtp * BT * H;
    const size_t tile_base = (size_t)bt * BT * H;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.payload, 0, p.payload_bytes, 0x00020000);
    unsigned* myflag = p.flags + bt * NC + c;
    unsigned* tflags = p.flags + bt * NC;
    const int hshift = __ffs(H) - 1;
    // where the finalising thread finds its element inside the fragment-ordered partial sums
    // (fragment lane (fu>>2)*16 + fj, register fu&3).  A fragment block is stored as 4 quads of 16 lanes x 4 floats with 16
    // floats of padding after each quad: unpadded, the 32 lanes of a read group hit 8 banks 4 ways (measured: 42 % of
    // the kernel's LDS cycles were bank-conflict cycles)
    const int rsrc_off = (fu >> 2) * 80 + fj * 4 + (fu & 3);
    const int sx = p.nofast ? 0 : cluster_same_xcd(p.hello + bt * NC, NC, c, p.status);
    if (sx < 0) return;
    const bool fast = sx == 1;
    if (tid >= CT) {                                  // ---- loader wave
        const int ll = tid - CT, u = ll >> 2, qd = ll & 3;
        const int bu = p.b0 + bt * BT + u;
        const bool uv = bu < p.B;
        const float* gsrc = p.gi + (size_t)bu * T * p.ldgi + c * 16 + qd * 4;
        f32x4 v[3];
        auto issue = [&](int t) {
#pragma unroll
            for (int g = 0; g < 3; ++g) v[g] = (uv && t < T) ? ld4(gsrc + (size_t)t * p.ldgi + g * H) : zero4();
        };
        auto flush = [&](int t) {                     // results of step t: LDS -> HBM
            if (!uv) return;
            const float* ob = obuf + (t & 1) * 1280 + u * 16 + qd * 4;
            const size_t row = (size_t)bu * T + t;
            const size_t o = row * p.ldy + c * 16 + qd * 4;
            const f32x4 h4 = ld4(ob);
            *reinterpret_cast<f32x4*>(p.y + o) = h4;
            if (p.ydrop) *reinterpret_cast<f32x4*>(p.ydrop + o) = h4 * dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
            if (p.sv0) {
                const size_t so = row * H + c * 16 + qd * 4;
                *reinterpret_cast<f32x4*>(p.sv0 + so) = ld4(ob + 256); *reinterpret_cast<f32x4*>(p.sv1 + so) = ld4(ob + 512);
                *reinterpret_cast<f32x4*>(p.sv2 + so) = ld4(ob + 768); *reinterpret_cast<f32x4*>(p.sv3 + so) = ld4(ob + 1024);
            }
        };
        issue(1);
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            float* gb = gbuf + ((t + 1) & 1) * 768 + u * 16 + qd * 4;
#pragma unroll
            for (int g = 0; g < 3; ++g) *reinterpret_cast<f32x4*>(gb + g * 256) = v[g];       // step t+1, landed
            if (t > 0) flush(t - 1);
            issue(t + 2);
            bar_lds();                                // the compute waves' partial-sum barrier
            if (t + 1 < T) { __builtin_amdgcn_s_barrier(); bar_lds(); }                       // publish / gather barriers
        }
        bar_lds();                                    // the last step's results are in LDS
        flush(T - 1);
        return;
    }
    float gin[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) gin[g] = valid ? p.gi[(size_t)b * T * p.ldgi + g * H + col] : 0.f;
    __syncthreads();

    long long* tr = (p.trace && blockIdx.x == 0 && tid == 0) ? p.trace : nullptr;
    long long* trl = reinterpret_cast<long long*>(obuf + 2 * 1280);
    for (int t = 0; t < T; ++t) {
        DEP_STAMP(0);
        f32x4 acc[3] = {zero4(), zero4(), zero4()};
        if constexpr (SPLIT) {
            // B fragment of k-step ks: lane (utterance j, k-group q) holds h[j][64 w + 32 ks + 8 q .. +7], hi and lo
            const int ho = j * LDHB + w * KCQ * 16 + q * 8;
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
            const float* hrow = hs + j * LDH + w * KCQ * 16 + q * 4;
            f32x4 hv[KCQ];
#pragma unroll
            for (int k = 0; k < KCQ; ++k) hv[k] = ld4(hrow + k * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < KCQ; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[g][k][e], hv[k][e], acc[g], 0, 0, 0);
        }
        DEP_STAMP(1);
#pragma unroll
        for (int g = 0; g < 3; ++g) *reinterpret_cast<f32x4*>(red + (w * 3 + g) * RED_BLK + (lane >> 4) * 80 + (lane & 15) * 4) = acc[g];
        bar_lds();
        DEP_STAMP(2);
        float tot[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s += red[(ww * 3 + g) * RED_BLK + rsrc_off];
            tot[g] = s;
        }
        const float r = fast_sigmoid(gin[0] + tot[0] + bh[0]);
        const float z = fast_sigmoid(gin[1] + tot[1] + bh[1]);
        const float hn = tot[2] + bh[2];
        const float n = fast_tanh(gin[2] + r * hn);
        const float h = (1.0f - z) * n + z * hprev;
        hprev = h; pool += h;
        const unsigned epoch = (unsigned)t + 1u;
        const size_t pbase = (size_t)(t & 1) * pstride + tile_base;
        const bool more = t + 1 < T;
        {                                             // results for the streaming wave (written out during step t+1)
            float* ob = obuf + (t & 1) * 1280 + tid;
            ob[0] = h; ob[256] = r; ob[512] = z; ob[768] = n; ob[1024] = hn;
        }
        DEP_STAMP(3);
        if (more) {
            gu32* dst = (gu32*)(p.payload + pbase + (size_t)fj * H + col);
            unsigned word = __float_as_uint(h);
            if constexpr (SPLIT) word = split_word(h);        // bf16 hi << 16 | bf16 lo
            if (fast) __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DEP_STAMP(4);
            __builtin_amdgcn_s_barrier();
            if (tid == 0) { if (fast) st_local(myflag, epoch); else st_agent(myflag, epoch); }
#pragma unroll
            for (int g = 0; g < 3; ++g) gin[g] = gbuf[((t + 1) & 1) * 768 + g * 256 + tid];   // written by the loader wave before this step's first barrier
        }
        if (more) {
            // every wave polls the (L2-resident) flags itself: saves the barrier that used to broadcast wave 0's verdict.
            // A wave that gives up leaves; the hardware barrier only counts live waves and the others give up as well
            // (the status word is raised).
            if (!wait_flags(tflags, NC, epoch, p.status, 4)) return;
            DEP_STAMP(5);
            DEP_STAMP(6);
            constexpr int PER = KCQ;                  // 16-byte pieces per thread = 16*H/4/256 = H/64
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i4 = (tid + CT * k) * 4;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)((pbase + i4) * 4), 0, 16 /* sc1 */);
                if constexpr (SPLIT) {                // four (hi << 16 | lo) words -> 8 bytes into each plane
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
    bar_lds();                                        // hands the last step's results to the streaming wave
    if (tr) for (int i = 0; i < 32; ++i) tr[i] = trl[i];
    if (valid) {
        if (p.pooled) p.pooled[(size_t)b * H + col] = pool * p.pool_scale;
        if (p.h_n) p.h_n[(size_t)b * H + col] = hprev;
    }
}

// split-precision forward image (see gru_fwd_cluster16<., true>): 16-byte piece
//   [((((c*3 + g)*4 + w)*KS2 + ks)*2 + plane)*64 + lane]  =  bf16 plane (0 hi, 1 lo) of
//   W[(g*H + 16c + (lane&15)) * H + 64w + 32ks + 8(lane>>4) + 0..7]          (KS2 = H/128 k-steps of 32 per wave)
__global__ void pack_cluster16_fwd_split_kernel(const float* __restrict__ W, u32x4* __restrict__ out, int H) {
    const int KS2 = H / 128;
    const long n = (long)(H / 16) * 3 * 4 * KS2 * 64;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int lane = idx & 63; long r = idx >> 6;
    const int ks = r % KS2; r /= KS2;
    const int w = r % 4; r /= 4;
    const int g = r % 3; const int c = r / 3;
    const float* src = W + (size_t)(g * H + 16 * c + (lane & 15)) * H + (H / 4) * w + 32 * ks + 8 * (lane >> 4);
    u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned a = split_word(src[2 * e]), b = split_word(src[2 * e + 1]);
        hi[e] = (a >> 16) | (b & 0xffff0000u);
        lo[e] = (a & 0xffffu) | (b << 16);
    }
    out[(idx - lane) * 2 + lane] = hi;
    out[(idx - lane) * 2 + 64 + lane] = lo;
}

}  // namespace

int dep_pack_cluster16_fwd_split(const float* w_hh, float* out, int H, hipStream_t s) {
    const long n = (long)(H / 16) * 3 * 4 * (H / 128) * 64;
    DEP_LAUNCH(pack_cluster16_fwd_split_kernel, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, w_hh, (u32x4*)out, H);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

// 16-unit members are used when they fit two per CU and the 32-unit clustering would leave CUs sharing nothing:
bool dep_cluster16_ok(int cell, int H, int B) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("DEP_CLUSTER16"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off || cell != DEP_CELL_GRU || H != 256) return false;
    (void)B;                                          // any batch: launches cover chunks of at most 512 utterances
    return true;
}

int dep_launch_cluster16_fwd(const dep_sweep_args& a, void* xbuf, size_t xbuf_bytes) {
    // co-residency bounds a launch to two workgroups per CU (512 = 32 tiles = 512 utterances on a full MI355X); larger
    // batches run chunk after chunk
    const int NC = a.H / 16, CH = dep_cluster_chunk(NC, 2, 512);
    const int nbtp_max = (dep_cdiv(a.B < CH ? a.B : CH, BT) + 7) / 8 * 8;
    F16 p{};
    p.B = a.B; p.T = a.T; p.H = a.H;
    p.wp = (const f32x4*)a.wp[0]; p.b_hh = a.b_hh[0];
    p.gi = a.gi; p.ldgi = 3 * a.H; p.y = a.y; p.ldy = a.ldy;
    p.ydrop = (a.drop_p > 0.f) ? a.ydrop : nullptr;
    p.drop_p = a.drop_p; p.drop_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f; p.seed = a.seed; p.site = a.site;
    p.pooled = a.pooled; p.pool_scale = a.pool_scale; p.h_n = a.h_n;
    p.sv0 = a.training ? a.sv0 : nullptr; p.sv1 = a.sv1; p.sv2 = a.sv2; p.sv3 = a.sv3;
    const size_t pay = (size_t)2 * nbtp_max * BT * a.H * sizeof(float);
    DEP_CHECK_ARG(xbuf && PAYLOAD_OFF + pay <= xbuf_bytes && (size_t)nbtp_max * NC <= 512);
    p.status = (unsigned*)xbuf; p.flags = (unsigned*)(hdr_base(xbuf, 0) + FLAG_OFF); p.hello = (unsigned*)(hdr_base(xbuf, 0) + HELLO_OFF);
    p.payload = (float*)((char*)xbuf + PAYLOAD_OFF); p.payload_bytes = (unsigned)pay; p.nofast = nofast_env();
    p.trace = trace_env() ? (long long*)(hdr_base(xbuf, 0) + TRACE_OFF) : nullptr;
    DepProfScope prof(DEP_PROF_GRU_FWD, a.stream);
    const size_t lds = (size_t)(BT * (a.H + 8) + 4 * 3 * RED_BLK + 2 * 768 + 2 * 1280 + 64) * sizeof(float);
    for (int b0 = 0; b0 < a.B; b0 += CH) {
        const int cb = a.B - b0 < CH ? a.B - b0 : CH;
        p.b0 = b0; p.nbtp = (dep_cdiv(cb, BT) + 7) / 8 * 8;
        // flags / hello words only: the status word is sticky over every sweep of a step (cleared by dep_rnn_forward)
        { const int rc_h = hdr_prepare(xbuf, 0, false, a.stream); if (rc_h) return rc_h; }
        if (a.split) DEP_LAUNCH((gru_fwd_cluster16<4, true>), dim3(NC * p.nbtp), dim3(CT + 64), lds, a.stream, p);
        else DEP_LAUNCH((gru_fwd_cluster16<4, false>), dim3(NC * p.nbtp), dim3(CT + 64), lds, a.stream, p);
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}
