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

#define DEP_STAMP(slot) do { if (tr && t >= 100 && t < 104) tr[(t - 100) * 8 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

struct F16 {
    int B, T, H, nbtp;
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

struct B16 {
    int B, T, H, nbtp;
    const f32x4* wp;
    const float* y; int ldy;
    const float* dy; int lddy;
    float drop_p, drop_scale; uint64_t seed; uint32_t site;
    const float* dpooled; float pool_scale;
    const float* dh_n;
    const float* sv0; const float* sv1; const float* sv2; const float* sv3;
    float* dgi; int lddg;
    float* dghn;
    float* dbpart;
    unsigned* status; unsigned* flags; unsigned* hello; float* payload; unsigned payload_bytes;
    int nofast;
};

// =============================================================================== forward
// matvec roles : lane = (utterance j = lane&15, k-quad q = lane>>4), wave w = K quarter
// finalise roles: thread = (utterance fj = tid>>4, unit fu = tid&15)  -> 64-byte row segments in every global access
template <int KCQ>      // k-chunks of 16 per wave = H/64
__global__ __launch_bounds__(CT) void gru_fwd_cluster16(F16 p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, LDH = H + LPAD, KC = H / 16, NC = H / 16;
    const int c = blockIdx.x / p.nbtp, bt = blockIdx.x % p.nbtp;
    if (bt * BT >= p.B) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int fj = tid >> 4, fu = tid & 15;
    const int col = c * 16 + fu;
    const int b = bt * BT + fj;
    const bool valid = b < p.B;
    float* hs = smem;                                 // [16][LDH]
    float* red = smem + BT * LDH;                     // [4 waves][3 gates][64 lanes][4]
    volatile int* deadflag = reinterpret_cast<volatile int*>(red + 4 * 3 * 64 * 4);
    for (int i = tid; i < BT * LDH; i += CT) hs[i] = 0.f;
    if (tid == 0) *deadflag = 0;

    f32x4 wr[3][KCQ];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < KCQ; ++k)
            wr[g][k] = p.wp[(size_t)((c * 3 + g) * KC + w * KCQ + k) * 64 + lane];
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
    const int rsrc_lane = (fu >> 2) * 16 + fj, rsrc_e = fu & 3;
    bool dead = false;
    const int sx = p.nofast ? 0 : cluster_same_xcd(p.hello + bt * NC, NC, c, p.status);
    if (sx < 0) return;
    const bool fast = sx == 1;
    float gin[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) gin[g] = valid ? p.gi[(size_t)b * T * p.ldgi + g * H + col] : 0.f;
    __syncthreads();

    long long* tr = (p.trace && blockIdx.x == 0 && tid == 0) ? p.trace : nullptr;
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)b * T + t;
        DEP_STAMP(0);
        // next step's input projection: issued here, under the MFMAs, and complete by the drain that precedes the
        // flag -- a load still in flight during the poll/gather would delay those (VMEM loads return in order)
        float gnx[3] = {0.f, 0.f, 0.f};
        if (valid && t + 1 < T) {
#pragma unroll
            for (int g = 0; g < 3; ++g) gnx[g] = p.gi[(row + 1) * p.ldgi + g * H + col];
        }
        f32x4 acc[3] = {zero4(), zero4(), zero4()};
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
        DEP_STAMP(1);
#pragma unroll
        for (int g = 0; g < 3; ++g) *reinterpret_cast<f32x4*>(red + ((w * 3 + g) * 64 + lane) * 4) = acc[g];
        bar_lds();
        DEP_STAMP(2);
        float tot[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s += red[((ww * 3 + g) * 64 + rsrc_lane) * 4 + rsrc_e];
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
        DEP_STAMP(3);
        if (more) {
            gu32* dst = (gu32*)(p.payload + pbase + (size_t)fj * H + col);
            if (fast) __hip_atomic_store(dst, __float_as_uint(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(dst, __float_as_uint(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DEP_STAMP(4);
            __builtin_amdgcn_s_barrier();
            if (tid == 0) { if (fast) st_local(myflag, epoch); else st_agent(myflag, epoch); }
#pragma unroll
            for (int g = 0; g < 3; ++g) gin[g] = gnx[g];      // landed: the drain above waited for it
        }
        if (valid) {
            const size_t o = row * p.ldy + col;
            p.y[o] = h;
            if (p.ydrop) p.ydrop[o] = h * dep_dropmask1(p.seed, p.site, o, p.drop_p, p.drop_scale);
            if (p.sv0) { const size_t so = row * H + col; p.sv0[so] = r; p.sv1[so] = z; p.sv2[so] = n; p.sv3[so] = hn; }
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
                f32x4 f;
                f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
                *reinterpret_cast<f32x4*>(hs + (i4 >> hshift) * LDH + (i4 & (H - 1))) = f;
            }
            bar_lds();
            DEP_STAMP(7);
        }
    }
    if (valid) {
        if (p.pooled) p.pooled[(size_t)b * H + col] = pool * p.pool_scale;
        if (p.h_n) p.h_n[(size_t)b * H + col] = hprev;
    }
}

// =============================================================================== backward
struct StepIn { float r, z, n, hn, hp, dy; };

template <int NTW>      // output tiles per wave = H/64
__global__ __launch_bounds__(CT) void gru_bwd_cluster16(B16 p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 48, KCB = KS / 16, LDG = KS + LPAD;
    const int H = p.H, T = p.T, NC = H / 16, NTT = H / 16;
    const int c = blockIdx.x / p.nbtp, bt = blockIdx.x % p.nbtp;
    if (bt * BT >= p.B) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fj = tid >> 4, fu = tid & 15;
    const int col = 16 * c + fu;
    const int b = bt * BT + fj;
    const bool valid = b < p.B;
    float* dgs = smem;                                // [16][LDG]

    f32x4 wr[NTW][KCB];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int k = 0; k < KCB; ++k)
            wr[i][k] = p.wp[(size_t)((c * NTT + w * NTW + i) * KCB + k) * 64 + lane];
    float dhrec = (p.dh_n && valid) ? p.dh_n[(size_t)b * H + col] : 0.f;
    const float dpl = (p.dpooled && valid) ? p.dpooled[(size_t)b * H + col] * p.pool_scale : 0.f;
    float dbr = 0.f, dbz = 0.f, dbn = 0.f, dbh = 0.f;
    const size_t pstride = (size_t)p.nbtp * NC * BT * H;
    const size_t tile_base = (size_t)bt * NC * BT * H;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.payload, 0, p.payload_bytes, 0x00020000);
    unsigned* myflag = p.flags + bt * NC + c;
    unsigned* tflags = p.flags + bt * NC;
    const int ml = lane & 15, mq = lane >> 4;
    const int g_lane = (fu >> 2) * 16 + fj, g_e = fu & 3;       // this thread's element inside a published fragment
    bool dead = false;
    const int sx = p.nofast ? 0 : cluster_same_xcd(p.hello + bt * NC, NC, c, p.status);
    if (sx < 0) return;
    const bool fast = sx == 1;

    auto load_step = [&](int t, StepIn& s) {
        s.r = s.z = s.n = s.hn = s.hp = s.dy = 0.f;
        if (valid && t >= 0) {
            const size_t row = (size_t)b * T + t;
            const size_t so = row * H + col;
            s.r = p.sv0[so]; s.z = p.sv1[so]; s.n = p.sv2[so]; s.hn = p.sv3[so];
            if (t > 0) s.hp = p.y[(row - 1) * p.ldy + col];
            if (p.dy) s.dy = p.dy[row * p.lddy + col];
        }
    };
    StepIn cur, nxt;
    load_step(T - 1, cur);

    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)b * T + t;
        float dyv = cur.dy;
        if (p.dy && p.drop_p > 0.f && valid) dyv *= dep_dropmask1(p.seed, p.site, row * p.lddy + col, p.drop_p, p.drop_scale);
        const float r = cur.r, z = cur.z, n = cur.n, hn = cur.hn, hp = cur.hp;
        const float d = dhrec + dpl + dyv;
        const float dn = d * (1.0f - z) * (1.0f - n * n);
        const float dz = d * (hp - n) * z * (1.0f - z);
        const float dr = dn * hn * r * (1.0f - r);
        const float dnr = dn * r;
        const float dzt = d * z;
        dgs[fj * LDG + fu] = dr; dgs[fj * LDG + 16 + fu] = dz; dgs[fj * LDG + 32 + fu] = dnr;
        if (valid) {
            float* g = p.dgi + row * p.lddg;
            g[col] = dr; g[H + col] = dz; g[2 * H + col] = dn;
            p.dghn[row * H + col] = dnr;
        }
        dbr += dr; dbz += dz; dbn += dn; dbh += dnr;
        __syncthreads();
        if (t == 0) break;
        load_step(t - 1, nxt);
        f32x4 acc[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) acc[i] = zero4();
        const float* drow = dgs + ml * LDG + mq * 4;
        f32x4 hv[KCB];
#pragma unroll
        for (int k = 0; k < KCB; ++k) hv[k] = ld4(drow + k * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KCB; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NTW; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[i][k][e], hv[k][e], acc[i], 0, 0, 0);
        const unsigned epoch = (unsigned)(T - t);
        const size_t pbase = (size_t)(t & 1) * pstride + tile_base;
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const size_t fo = pbase + ((size_t)(c * NTT + w * NTW + i) * 64 + lane) * 4;
            u32x4 v;
            v.x = __float_as_uint(acc[i][0]); v.y = __float_as_uint(acc[i][1]);
            v.z = __float_as_uint(acc[i][2]); v.w = __float_as_uint(acc[i][3]);
            if (fast) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (unsigned)(fo * 4), 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (unsigned)(fo * 4), 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) { if (fast) st_local(myflag, epoch); else st_agent(myflag, epoch); }
        if (w == 0 && !wait_flags(tflags, NC, epoch, p.status, 3)) dead = true;
        if (__syncthreads_or(dead)) return;
        // this thread's column of every member's partial, summed in member order
        const float* src = p.payload + pbase + ((size_t)c * 64 + g_lane) * 4 + g_e;
        float part[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) part[m] = (m < NC) ? ldf_agent(src + (size_t)m * NTT * 256) : 0.f;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) s += part[m];
        dhrec = dzt + s;
        cur = nxt;
    }
    // bias-gradient partials dbpart[bt][4][H]: sum over the 16 utterances = lanes differing in bits 4,5 and the 4 waves
    float a[4] = {dbr, dbz, dbn, dbh};
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] += __shfl_xor(a[k], 16, 64); a[k] += __shfl_xor(a[k], 32, 64); }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) smem[(w * 4 + k) * 16 + lane] = a[k];
    }
    __syncthreads();
    if (tid < 64) {
        const int k = tid >> 4, u = tid & 15;
        const float s = smem[(0 * 4 + k) * 16 + u] + smem[(1 * 4 + k) * 16 + u] + smem[(2 * 4 + k) * 16 + u] + smem[(3 * 4 + k) * 16 + u];
        p.dbpart[(size_t)bt * 4 * H + k * H + 16 * c + u] = s;
    }
}

// member image for the backward: out[((c*(H/16) + jt)*3 + kc)*256 + l*4 + e] = W[(g*H + 16c + u)*H + jt*16 + (l&15)],
// k = kc*16 + (l>>4)*4 + e, g = k/16, u = k%16
__global__ void pack_cluster16_bwd_kernel(const float* __restrict__ W, float* __restrict__ out, int H) {
    const long n = 3L * H * H;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int e = idx & 3, l = (idx >> 2) & 63;
    const long blk = idx >> 8;
    const int kc = blk % 3; const long r = blk / 3;
    const int jt = r % (H / 16); const int c = r / (H / 16);
    const int k = kc * 16 + (l >> 4) * 4 + e;
    const int g = k / 16, u = k % 16;
    out[idx] = W[(size_t)(g * H + 16 * c + u) * H + jt * 16 + (l & 15)];
}

}  // namespace

// 16-unit members are used when they fit two per CU and the 32-unit clustering would leave CUs sharing nothing:
bool dep_cluster16_ok(int cell, int H, int B) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("DEP_CLUSTER16"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off || cell != DEP_CELL_GRU || H != 256) return false;
    const int nbtp = (dep_cdiv(B, BT) + 7) / 8 * 8;
    return (H / 16) * nbtp <= 512;
}

int dep_pack_cluster16_bwd(const float* w_hh, float* out, int H, hipStream_t s) {
    const long n = 3L * H * H;
    hipLaunchKernelGGL(pack_cluster16_bwd_kernel, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, w_hh, out, H);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

int dep_launch_cluster16_fwd(const dep_sweep_args& a, void* xbuf, size_t xbuf_bytes) {
    const int NC = a.H / 16, nbt = dep_cdiv(a.B, BT), nbtp = (nbt + 7) / 8 * 8;
    F16 p{};
    p.B = a.B; p.T = a.T; p.H = a.H; p.nbtp = nbtp;
    p.wp = (const f32x4*)a.wp[0]; p.b_hh = a.b_hh[0];
    p.gi = a.gi; p.ldgi = 3 * a.H; p.y = a.y; p.ldy = a.ldy;
    p.ydrop = (a.drop_p > 0.f) ? a.ydrop : nullptr;
    p.drop_p = a.drop_p; p.drop_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f; p.seed = a.seed; p.site = a.site;
    p.pooled = a.pooled; p.pool_scale = a.pool_scale; p.h_n = a.h_n;
    p.sv0 = a.training ? a.sv0 : nullptr; p.sv1 = a.sv1; p.sv2 = a.sv2; p.sv3 = a.sv3;
    const size_t pay = (size_t)2 * nbtp * BT * a.H * sizeof(float);
    DEP_CHECK_ARG(xbuf && PAYLOAD_OFF + pay <= xbuf_bytes && (size_t)nbtp * NC <= 512);
    p.status = (unsigned*)xbuf; p.flags = (unsigned*)((char*)xbuf + FLAG_OFF); p.hello = (unsigned*)((char*)xbuf + HELLO_OFF);
    p.payload = (float*)((char*)xbuf + PAYLOAD_OFF); p.payload_bytes = (unsigned)pay; p.nofast = nofast_env();
    p.trace = trace_env() ? (long long*)((char*)xbuf + TRACE_OFF) : nullptr;
    if (hipMemsetAsync(xbuf, 0, PAYLOAD_OFF, a.stream) != hipSuccess) { dep_set_error("hipMemsetAsync failed"); return DEP_ERR_HIP; }
    DepProfScope prof(DEP_PROF_GRU_FWD, a.stream);
    const size_t lds = (size_t)(BT * (a.H + LPAD) + 4 * 3 * 64 * 4 + 16) * sizeof(float);
    hipLaunchKernelGGL(gru_fwd_cluster16<4>, dim3(NC * nbtp), dim3(CT), lds, a.stream, p);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

int dep_launch_cluster16_bwd(const dep_sweep_bwd_args& a, void* xbuf, size_t xbuf_bytes) {
    const int NC = a.H / 16, nbt = dep_cdiv(a.B, BT), nbtp = (nbt + 7) / 8 * 8;
    B16 p{};
    p.B = a.B; p.T = a.T; p.H = a.H; p.nbtp = nbtp;
    p.wp = (const f32x4*)a.wpT[0];
    p.y = a.y; p.ldy = a.ldy; p.dy = a.dy; p.lddy = a.lddy;
    p.drop_p = a.dy ? a.drop_p : 0.f; p.drop_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    p.seed = a.seed; p.site = a.site;
    p.dpooled = a.dpooled; p.pool_scale = a.pool_scale; p.dh_n = a.dh_n;
    p.sv0 = a.sv0; p.sv1 = a.sv1; p.sv2 = a.sv2; p.sv3 = a.sv3;
    p.dgi = a.dgi; p.lddg = 3 * a.H; p.dghn = a.dghn; p.dbpart = a.dbpart;
    DEP_CHECK_ARG(a.dbpart_rows >= nbt);
    const size_t pay = (size_t)2 * nbtp * NC * BT * a.H * sizeof(float);
    DEP_CHECK_ARG(xbuf && PAYLOAD_OFF + pay <= xbuf_bytes && (size_t)nbtp * NC <= 512);
    p.status = (unsigned*)xbuf; p.flags = (unsigned*)((char*)xbuf + FLAG_OFF); p.hello = (unsigned*)((char*)xbuf + HELLO_OFF);
    p.payload = (float*)((char*)xbuf + PAYLOAD_OFF); p.payload_bytes = (unsigned)pay; p.nofast = nofast_env();
    if (hipMemsetAsync(xbuf, 0, PAYLOAD_OFF, a.stream) != hipSuccess) { dep_set_error("hipMemsetAsync failed"); return DEP_ERR_HIP; }
    DepProfScope prof(DEP_PROF_GRU_BWD, a.stream);
    const size_t lds = (size_t)(BT * (48 + LPAD) + 64) * sizeof(float);
    hipLaunchKernelGGL(gru_bwd_cluster16<4>, dim3(NC * nbtp), dim3(CT), lds, a.stream, p);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
