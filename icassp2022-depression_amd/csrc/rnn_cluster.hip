// Cluster-parallel recurrent sweeps: the serial critical path of nn.GRU / nn.LSTM spread over the chip.
//
// Problem with one workgroup per 16-utterance tile (rnn_sweep.hip): B = 512 gives 32 workgroups, each
// streaming the whole W_hh from L2 every step and doing 6.3 MFLOP of fp32 MFMA on ONE CU -> ~20 us/step.
// Here a tile is owned by a CLUSTER of NC = H/32 workgroups (one per CU, all co-resident).  Member c keeps
// the W_hh rows of hidden units [32c, 32c+32) in VGPRs for the whole sweep (96 KB per CU at H = 256: no
// weight traffic at all), computes its slice of the gates with v_mfma_f32_16x16x4_f32 and exchanges
//   forward : its 16x32 slice of h_t            (all-gather)
//   backward: its 16xH partial of dh_{t-1}      (reduce-scatter, summed in fixed member order)
// with the other members through 8-byte {epoch tag, value} granules in global memory, written with
// agent-scope (sc1, write-through) stores and polled with agent-scope loads -- the data is its own flag,
// so no fences and no L1 staleness (cdna_hip_programming.md G16 R2).  Buffers are double-buffered by step
// parity (a member can run at most one step ahead of any consumer) and zeroed before every launch; every
// spin is bounded and raises a status word instead of hanging.
//
// Numerics: identical to rnn_sweep.hip up to fp32 summation order (K is split across two waves / the
// dh reduction across members); still exact-f32 MFMA products.
#include "dep_common.h"

namespace {

constexpr int BT = 16;
constexpr int LPAD = 4;
constexpr int CT = 256;                 // threads per workgroup (4 waves, one per SIMD)
constexpr unsigned SPIN_LIMIT = 1u << 19;

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

struct CDir { const f32x4* wp; const float* b_hh; };

struct CFwdP {
    int B, T, H, dirs, nbtp;
    CDir d[2];
    const float* gi; int ldgi;
    float* y; int ldy;
    float* ydrop; float drop_p, drop_scale; uint64_t seed; uint32_t site;
    float* pooled; float pool_scale;
    float* h_n;
    float* sv0; float* sv1; float* sv2; float* sv3;
    u64* xbuf; unsigned* status;
};

struct CBwdP {
    int B, T, H, dirs, nbtp;
    CDir d[2];
    const float* y; int ldy;
    const float* dy; int lddy;
    float drop_p, drop_scale; uint64_t seed; uint32_t site;
    const float* dpooled; float pool_scale;
    const float* dh_n;
    const float* sv0; const float* sv1; const float* sv2; const float* sv3;
    float* dgi; int lddg;
    float* dghn;
    float* dbpart; int nwg;
    u64* xbuf; unsigned* status;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

__device__ __forceinline__ void put_granule(u64* p, unsigned epoch, float v) {
    __hip_atomic_store((gu64*)p, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 get_granule(const u64* p) {
    return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned peek_status(unsigned* s) {
    return __hip_atomic_load((gu32*)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void raise_status(unsigned* s, unsigned code) {
    __hip_atomic_store((gu32*)s, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Re-read N granules (addresses base + off[k]) every pass until all carry `epoch`; wave-uniform exit.
template <int N>
__device__ __forceinline__ bool sweep(const u64* base, const int (&off)[N], unsigned epoch, float (&val)[N],
                                      unsigned* status) {
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const u64 x = get_granule(base + off[k]);
            val[k] = __uint_as_float((unsigned)x);
            ok = ok && ((unsigned)(x >> 32) == epoch);
        }
        if (__all(ok)) return true;
        if (spins > SPIN_LIMIT) { raise_status(status, 2); return false; }
        if ((spins & 63) == 63 && peek_status(status) != 0) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ float2 rowsum16_2(float2 v) {
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { v.x += __shfl_xor(v.x, m, 64); v.y += __shfl_xor(v.y, m, 64); }
    return v;
}

// =============================================================================== GRU forward
// grid.x = NC * nbtp (member-major so that all members of a tile share blockIdx % 8, i.e. the XCD).
// Wave w: hidden tile jl = w>>1 of the member's two, K half kh = w&1; the pair (w, w^1) sums its halves
// through LDS and each wave finalises two of the four elements a lane holds per gate.
template <int KCH>      // k-chunks of 16 per wave = H/32
__global__ __launch_bounds__(CT) void gru_fwd_cluster(CFwdP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, LDH = H + LPAD, KC = H / 16;
    const int c = blockIdx.x / p.nbtp, bt = blockIdx.x % p.nbtp;
    if (bt * BT >= p.B) return;                       // padding tile: the whole cluster leaves together
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4, jl = w >> 1, kh = w & 1;
    const int jt = c * 2 + jl;
    const int b = bt * BT + j;
    const bool valid = b < p.B;
    float* hs = smem;                                 // [16][LDH]  h_{t-1}, all H columns
    float* red = smem + BT * LDH;                     // [4][3][64][4] pair exchange of partial gate sums
    for (int i = tid; i < BT * LDH; i += CT) hs[i] = 0.f;

    f32x4 wr[3][KCH];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < KCH; ++k)
            wr[g][k] = p.d[0].wp[(size_t)((jt * 3 + g) * KC + kh * KCH + k) * 64 + lane];
    const int col = jt * 16 + q * 4 + 2 * kh;         // first of the two hidden units this lane finalises
    float2 bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = ld2(p.d[0].b_hh + g * H + col);
    float2 hprev = {0.f, 0.f}, pool = {0.f, 0.f};
    const size_t xstride = (size_t)p.nbtp * BT * H;   // one parity buffer
    u64* xb = p.xbuf + (size_t)bt * BT * H;
    const int hshift = __ffs(H) - 1;                  // H is a power of two here (dep_cluster_ok)
    __syncthreads();
    bool dead = false;
    float2 gin[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
        gin[g] = valid ? ld2(p.gi + (size_t)b * T * p.ldgi + g * H + col) : make_float2(0.f, 0.f);

    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)b * T + t;
        float2 gi[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) gi[g] = gin[g];
        f32x4 acc[3] = {zero4(), zero4(), zero4()};
        const float* hrow = hs + j * LDH + kh * KCH * 16 + q * 4;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const f32x4 hv = ld4(hrow + k * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[g][k][e], hv[e], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) st4(red + ((w * 3 + g) * 64 + lane) * 4, acc[g]);
        __syncthreads();
        float2 tot[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float2 pv = ld2(red + (((w ^ 1) * 3 + g) * 64 + lane) * 4 + 2 * kh);
            tot[g].x = (kh ? acc[g][2] : acc[g][0]) + pv.x;
            tot[g].y = (kh ? acc[g][3] : acc[g][1]) + pv.y;
        }
        float2 r, z, hn, n, h;
        r.x = dep_sigmoid(gi[0].x + tot[0].x + bh[0].x); r.y = dep_sigmoid(gi[0].y + tot[0].y + bh[0].y);
        z.x = dep_sigmoid(gi[1].x + tot[1].x + bh[1].x); z.y = dep_sigmoid(gi[1].y + tot[1].y + bh[1].y);
        hn.x = tot[2].x + bh[2].x; hn.y = tot[2].y + bh[2].y;
        n.x = tanhf(gi[2].x + r.x * hn.x); n.y = tanhf(gi[2].y + r.y * hn.y);
        h.x = (1.0f - z.x) * n.x + z.x * hprev.x; h.y = (1.0f - z.y) * n.y + z.y * hprev.y;
        hprev = h; pool.x += h.x; pool.y += h.y;
        // publish this lane's two values of h_t (every row, so that consumers never wait on padding rows)
        const unsigned epoch = (unsigned)t + 1u;
        u64* xp = xb + (size_t)(t & 1) * xstride;
        put_granule(xp + j * H + col, epoch, h.x);
        put_granule(xp + j * H + col + 1, epoch, h.y);
        if (valid) {
            const size_t o = row * p.ldy + col;
            st2(p.y + o, h);
            if (p.ydrop) {
                const f32x4 m = dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
                st2(p.ydrop + o, make_float2(h.x * (kh ? m[2] : m[0]), h.y * (kh ? m[3] : m[1])));
            }
            if (p.sv0) {
                const size_t so = row * H + col;
                st2(p.sv0 + so, r); st2(p.sv1 + so, z); st2(p.sv2 + so, n); st2(p.sv3 + so, hn);
            }
        }
        // gather all of h_t (own slice included) into LDS for the next step
        if (t + 1 < T) {
            // next step's input projection does not depend on the recurrence: fetch it under the exchange
            if (valid) {
#pragma unroll
                for (int g = 0; g < 3; ++g) gin[g] = ld2(p.gi + (row + 1) * p.ldgi + g * H + col);
            }
            constexpr int PER = 2 * KCH;              // granules per thread = 16*H/256, all in flight at once
            int off[PER]; float val[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) off[k] = tid + CT * k;
            if (!sweep<PER>(xp, off, epoch, val, p.status)) dead = true;
#pragma unroll
            for (int k = 0; k < PER; ++k) { const int idx = off[k]; hs[(idx >> hshift) * LDH + (idx & (H - 1))] = val[k]; }
            if (__syncthreads_or(dead)) return;
        }
    }
    if (valid) {
        if (p.pooled) st2(p.pooled + (size_t)b * H + col, make_float2(pool.x * p.pool_scale, pool.y * p.pool_scale));
        if (p.h_n) st2(p.h_n + (size_t)b * H + col, hprev);
    }
}

// =============================================================================== GRU backward
// Member c holds W_hh[rows of units 32c..32c+31 (all 3 gates), all H columns] and computes, from ITS 16x96
// slice of dgh_t, a partial dh_{t-1} for ALL H columns; members exchange partials (reduce-scatter).
// Thread -> element map: j = tid & 15 (utterance), u = 2*(tid >> 4) + {0,1} (unit inside the member's 32).
template <int NTW>      // output tiles (16 columns) per wave = H/64
__global__ __launch_bounds__(CT) void gru_bwd_cluster(CBwdP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 96, KCB = KS / 16, LDG = KS + LPAD;
    const int H = p.H, T = p.T, NC = H / 32;
    const int c = blockIdx.x / p.nbtp, bt = blockIdx.x % p.nbtp;
    if (bt * BT >= p.B) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = tid & 15, u2 = tid >> 4;
    const int col = 32 * c + 2 * u2;                  // global hidden unit of this thread's first element
    const int b = bt * BT + j;
    const bool valid = b < p.B;
    float* dgs = smem;                                // [16][LDG] this member's slice of dgh_t

    f32x4 wr[NTW][KCB];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int k = 0; k < KCB; ++k)
            wr[i][k] = p.d[0].wp[(size_t)((c * (H / 16) + w * NTW + i) * KCB + k) * 64 + lane];
    float2 dhrec = (p.dh_n && valid) ? ld2(p.dh_n + (size_t)b * H + col) : make_float2(0.f, 0.f);
    float2 dpl = make_float2(0.f, 0.f);
    if (p.dpooled && valid) { dpl = ld2(p.dpooled + (size_t)b * H + col); dpl.x *= p.pool_scale; dpl.y *= p.pool_scale; }
    float2 dbr = {0.f, 0.f}, dbz = {0.f, 0.f}, dbn = {0.f, 0.f}, dbh = {0.f, 0.f};
    const size_t xstride = (size_t)p.nbtp * NC * BT * H;
    u64* xb = p.xbuf + (size_t)bt * NC * BT * H;
    const int ml = lane & 15, mq = lane >> 4;         // MFMA lane roles (batch row / k-quad) -- differ from (j,u2)
    bool dead = false;

    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)b * T + t;
        float2 r = {0.f, 0.f}, z = r, n = r, hn = r, hp = r, d = make_float2(dhrec.x + dpl.x, dhrec.y + dpl.y);
        if (valid) {
            const size_t so = row * H + col;
            r = ld2(p.sv0 + so); z = ld2(p.sv1 + so); n = ld2(p.sv2 + so); hn = ld2(p.sv3 + so);
            if (t > 0) hp = ld2(p.y + (row - 1) * p.ldy + col);
            if (p.dy) {
                const size_t o = row * p.lddy + col;
                float2 dyv = ld2(p.dy + o);
                if (p.drop_p > 0.f) {
                    const f32x4 m = dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
                    const int e0 = (int)(o & 3);
                    dyv.x *= (e0 ? m[2] : m[0]); dyv.y *= (e0 ? m[3] : m[1]);
                }
                d.x += dyv.x; d.y += dyv.y;
            }
        }
        float2 dn, dz, dr, dnr, dzt;
        dn.x = d.x * (1.0f - z.x) * (1.0f - n.x * n.x); dn.y = d.y * (1.0f - z.y) * (1.0f - n.y * n.y);
        dz.x = d.x * (hp.x - n.x) * z.x * (1.0f - z.x); dz.y = d.y * (hp.y - n.y) * z.y * (1.0f - z.y);
        dr.x = dn.x * hn.x * r.x * (1.0f - r.x); dr.y = dn.y * hn.y * r.y * (1.0f - r.y);
        dnr.x = dn.x * r.x; dnr.y = dn.y * r.y;
        dzt.x = d.x * z.x; dzt.y = d.y * z.y;
        st2(dgs + j * LDG + 2 * u2, dr); st2(dgs + j * LDG + 32 + 2 * u2, dz); st2(dgs + j * LDG + 64 + 2 * u2, dnr);
        if (valid) {
            float* g = p.dgi + row * p.lddg;
            st2(g + col, dr); st2(g + H + col, dz); st2(g + 2 * H + col, dn);
            st2(p.dghn + row * H + col, dnr);
        }
        dbr.x += dr.x; dbr.y += dr.y; dbz.x += dz.x; dbz.y += dz.y; dbn.x += dn.x; dbn.y += dn.y; dbh.x += dnr.x; dbh.y += dnr.y;
        __syncthreads();
        if (t > 0) {
            f32x4 acc[NTW];
#pragma unroll
            for (int i = 0; i < NTW; ++i) acc[i] = zero4();
            const float* drow = dgs + ml * LDG + mq * 4;
#pragma unroll
            for (int k = 0; k < KCB; ++k) {
                const f32x4 hv = ld4(drow + k * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < NTW; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[i][k][e], hv[e], acc[i], 0, 0, 0);
            }
            // publish this member's partial dh for all H columns: xb[parity][src c][row][col]
            const unsigned epoch = (unsigned)(T - t);
            u64* xp = xb + (size_t)(t & 1) * xstride;
            u64* mine = xp + ((size_t)c * BT + ml) * H;
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int oc = (w * NTW + i) * 16 + mq * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) put_granule(mine + oc + e, epoch, acc[i][e]);
            }
            // gather the NC partials of this thread's two columns, sum in member order
            float2 s = {0.f, 0.f};
            for (int s0 = 0; s0 < NC && !dead; s0 += 4) {
                int off[8]; float val[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    off[2 * k] = (int)(((size_t)(s0 + k) * BT + j) * H + col);
                    off[2 * k + 1] = off[2 * k] + 1;
                }
                if (!sweep<8>(xp, off, epoch, val, p.status)) dead = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) { s.x += val[2 * k]; s.y += val[2 * k + 1]; }
            }
            dhrec.x = dzt.x + s.x; dhrec.y = dzt.y + s.y;
            if (__syncthreads_or(dead)) return;
        }
    }
    // bias-gradient partials: dbpart[bt][4][H], every member writes its own 32 columns
    const float2 s0 = rowsum16_2(dbr), s1 = rowsum16_2(dbz), s2 = rowsum16_2(dbn), s3 = rowsum16_2(dbh);
    if (j == 0) {
        float* o = p.dbpart + (size_t)bt * 4 * H;
        st2(o + col, s0); st2(o + H + col, s1); st2(o + 2 * H + col, s2); st2(o + 3 * H + col, s3);
    }
}

// cluster-backward weight image: wpc[((c*(H/16) + jt)*KCB + kc)*256 + l*4 + e] = W[(g*H + 32c + u)*H + jt*16 + (l&15)]
// with k = kc*16 + (l>>4)*4 + e, g = k/32, u = k%32   (G*32 = KS rows of the member, all H columns)
__global__ void pack_cluster_bwd_kernel(const float* __restrict__ W, float* __restrict__ out, int G, int H) {
    const long n = (long)G * H * H;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int e = idx & 3, l = (idx >> 2) & 63;
    const long blk = idx >> 8;
    const int KCB = G * 2;
    const int kc = blk % KCB; const long r = blk / KCB;
    const int jt = r % (H / 16); const int c = r / (H / 16);
    const int k = kc * 16 + (l >> 4) * 4 + e;
    const int g = k / 32, u = k % 32;
    out[idx] = W[(size_t)(g * H + 32 * c + u) * H + jt * 16 + (l & 15)];
}

}  // namespace

// ------------------------------------------------------------------------------- host side
bool dep_cluster_ok(int cell, int H, int B, int dirs) {
    if (cell != DEP_CELL_GRU) return false;
    if (H != 128 && H != 256) return false;           // KCH = H/32 in {4,8}; NTW = H/64 in {2,4}; gather = H/16 granules/thread
    const int NC = H / 32;
    const int nbt = dep_cdiv(B, BT);
    const int nbtp = (nbt + 7) / 8 * 8;
    (void)dirs;
    return NC * nbtp <= 256;                          // every workgroup must be resident: one per CU
}

size_t dep_cluster_xbuf_bytes(int cell, int H, int B, int dirs) {
    if (!dep_cluster_ok(cell, H, B, dirs)) return 0;
    const int NC = H / 32, nbtp = (dep_cdiv(B, BT) + 7) / 8 * 8;
    return 16384 + (size_t)2 * nbtp * NC * BT * H * sizeof(u64) * dirs;    // header + the largest (backward) exchange
}

int dep_pack_cluster_bwd(const float* w_hh, float* out, int G, int H, hipStream_t s) {
    const long n = (long)G * H * H;
    hipLaunchKernelGGL(pack_cluster_bwd_kernel, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, w_hh, out, G, H);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

// Tagged-granule variant of the forward exchange (DEP_CLUSTER_FWD=granule); default: rnn_cluster_bwd.hip.
int dep_launch_cluster_fwd_granule(const dep_sweep_args& a, void* xbuf, size_t xbuf_bytes) {
    DEP_CHECK_ARG(dep_cluster_ok(a.cell, a.H, a.B, a.dirs) && xbuf && xbuf_bytes >= dep_cluster_xbuf_bytes(a.cell, a.H, a.B, a.dirs));
    const int NC = a.H / 32, nbtp = (dep_cdiv(a.B, BT) + 7) / 8 * 8;
    CFwdP p{};
    p.B = a.B; p.T = a.T; p.H = a.H; p.dirs = a.dirs; p.nbtp = nbtp;
    p.d[0].wp = (const f32x4*)a.wp[0]; p.d[0].b_hh = a.b_hh[0];
    p.gi = a.gi; p.ldgi = 3 * a.H; p.y = a.y; p.ldy = a.ldy;
    p.ydrop = (a.drop_p > 0.f) ? a.ydrop : nullptr;
    p.drop_p = a.drop_p; p.drop_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f; p.seed = a.seed; p.site = a.site;
    p.pooled = a.pooled; p.pool_scale = a.pool_scale; p.h_n = a.h_n;
    p.sv0 = a.training ? a.sv0 : nullptr; p.sv1 = a.sv1; p.sv2 = a.sv2; p.sv3 = a.sv3;
    p.status = (unsigned*)xbuf; p.xbuf = (u64*)((char*)xbuf + 256);
    const size_t used = 256 + (size_t)2 * nbtp * BT * a.H * sizeof(u64);
    if (hipMemsetAsync(xbuf, 0, used, a.stream) != hipSuccess) { dep_set_error("hipMemsetAsync failed"); return DEP_ERR_HIP; }
    DepProfScope prof(DEP_PROF_GRU_FWD, a.stream);
    const size_t lds = (size_t)(BT * (a.H + LPAD) + 4 * 3 * 64 * 4) * sizeof(float);
    dim3 grid(NC * nbtp);
    switch (a.H / 32) {
        case 2: hipLaunchKernelGGL(gru_fwd_cluster<2>, grid, dim3(CT), lds, a.stream, p); break;
        case 4: hipLaunchKernelGGL(gru_fwd_cluster<4>, grid, dim3(CT), lds, a.stream, p); break;
        case 6: hipLaunchKernelGGL(gru_fwd_cluster<6>, grid, dim3(CT), lds, a.stream, p); break;
        default: hipLaunchKernelGGL(gru_fwd_cluster<8>, grid, dim3(CT), lds, a.stream, p); break;
    }
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

// Tagged-granule variant of the backward exchange, kept for A/B runs (DEP_CLUSTER_BWD=granule); the default
// is the flag-published variant in rnn_cluster_bwd.hip.
int dep_launch_cluster_bwd_granule(const dep_sweep_bwd_args& a, void* xbuf, size_t xbuf_bytes) {
    DEP_CHECK_ARG(dep_cluster_ok(a.cell, a.H, a.B, a.dirs) && xbuf && xbuf_bytes >= dep_cluster_xbuf_bytes(a.cell, a.H, a.B, a.dirs));
    const int NC = a.H / 32, nbtp = (dep_cdiv(a.B, BT) + 7) / 8 * 8;
    CBwdP p{};
    p.B = a.B; p.T = a.T; p.H = a.H; p.dirs = a.dirs; p.nbtp = nbtp;
    p.d[0].wp = (const f32x4*)a.wpT[0];
    p.y = a.y; p.ldy = a.ldy; p.dy = a.dy; p.lddy = a.lddy;
    p.drop_p = a.dy ? a.drop_p : 0.f; p.drop_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    p.seed = a.seed; p.site = a.site;
    p.dpooled = a.dpooled; p.pool_scale = a.pool_scale; p.dh_n = a.dh_n;
    p.sv0 = a.sv0; p.sv1 = a.sv1; p.sv2 = a.sv2; p.sv3 = a.sv3;
    p.dgi = a.dgi; p.lddg = 3 * a.H; p.dghn = a.dghn; p.dbpart = a.dbpart; p.nwg = dep_cdiv(a.B, BT);
    DEP_CHECK_ARG(a.dbpart_rows >= p.nwg);
    p.status = (unsigned*)xbuf; p.xbuf = (u64*)((char*)xbuf + 256);
    const size_t used = 256 + (size_t)2 * nbtp * NC * BT * a.H * sizeof(u64);
    if (hipMemsetAsync(xbuf, 0, used, a.stream) != hipSuccess) { dep_set_error("hipMemsetAsync failed"); return DEP_ERR_HIP; }
    DepProfScope prof(DEP_PROF_GRU_BWD, a.stream);
    const size_t lds = (size_t)(BT * (96 + LPAD)) * sizeof(float);
    dim3 grid(NC * nbtp);
    switch (a.H / 64) {
        case 1: hipLaunchKernelGGL(gru_bwd_cluster<1>, grid, dim3(CT), lds, a.stream, p); break;
        case 2: hipLaunchKernelGGL(gru_bwd_cluster<2>, grid, dim3(CT), lds, a.stream, p); break;
        case 3: hipLaunchKernelGGL(gru_bwd_cluster<3>, grid, dim3(CT), lds, a.stream, p); break;
        default: hipLaunchKernelGGL(gru_bwd_cluster<4>, grid, dim3(CT), lds, a.stream, p); break;
    }
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
