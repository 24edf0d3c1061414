// Persistent recurrent sweeps for nn.GRU / nn.LSTM (forward and BPTT) on gfx950.
//
// One launch walks all T time steps of one layer (both directions of a BiLSTM run side by side
// in blockIdx.y).  A workgroup owns a tile of 16 utterances (the MFMA N dimension) and the whole
// recurrent matrix: per step it computes   G^T (gates x 16) = W_hh (gates x H) * h_{t-1}^T (H x 16)
// with v_mfma_f32_16x16x4_f32 (exact f32, == fmaf chain), W_hh streamed from L2 in a pre-packed
// MFMA-fragment order (one coalesced 1 KiB load per wave per 16x16 k-chunk), h_{t-1} resident in
// LDS, and the gate nonlinearities / state update fused in the epilogue of the same step.
// The time-parallel parts (x_t W_ih^T, dW, dX) are hoisted out to the MFMA GEMM (gemm.hip).
//
// Reference semantics: torch.nn.GRU at Classification/audio_gru_whole.py:59-60,105 and
// torch.nn.LSTM at Classification/text_bilstm_whole.py:54-56,105 (gate order r,z,n / i,f,g,o,
// h0 = c0 = 0, inter-layer dropout on the layer output in training mode).
#include "dep_common.h"

namespace {

constexpr int BT = 16;       // utterances per workgroup = MFMA N
constexpr int LPAD = 4;      // LDS row padding (floats): keeps rows 16-B aligned, de-phases banks

struct DirP {                 // per-direction pointers
    const f32x4* wp;          // packed recurrent weights (fwd: W_hh ; bwd: W_hh^T)
    const float* w;           // plain (G*H, H) weights (generic kernels)
    const float* b_hh;        // GRU recurrent bias (3H) or nullptr
};

struct FwdP {
    int B, T, H, dirs;
    DirP d[2];
    const float* gi; int ldgi;          // (B,T,ldgi), direction d uses columns [d*G*H, (d+1)*G*H)
    float* y; int ldy;                   // (B,T,ldy),  direction d writes columns [d*H, (d+1)*H)
    float* ydrop; float drop_p, drop_scale; uint64_t seed; uint32_t site;
    float* pooled; float pool_scale;     // (B,H)
    float* h_n;                          // (dirs,B,H)
    float* sv0; float* sv1; float* sv2; float* sv3;   // GRU: r,z,n,hn (B,T,H) ; LSTM: gates (B,T,dirs*4H), c (B,T,dirs*H)
};

struct BwdP {
    int B, T, H, dirs;
    DirP d[2];
    const float* y; int ldy;
    const float* dy; int lddy;
    float drop_p, drop_scale; uint64_t seed; uint32_t site;
    const float* dpooled; float pool_scale;
    const float* dh_n;
    const float* sv0; const float* sv1; const float* sv2; const float* sv3;
    float* dgi; int lddg;                // (B,T,lddg) lddg = dirs*G*H
    float* dghn;                         // GRU (B,T,H)
    float* dbpart;                       // [dirs][nwg][G'][H]  G' = 4
    int nwg;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// acc[t] += Wtile[t] (16 x K) * v^T (K x 16).  wp: this wave's first packed tile + lane, tile t
// chunk kc at wp[(t*KC + kc)*64].  vs: LDS row (lane&15) + (lane>>4)*4 of the 16 x K operand.
template <int NTL>
__device__ __forceinline__ void matvec_tiles(f32x4 (&acc)[NTL], const f32x4* __restrict__ wp, int KC,
                                             const float* vs) {
    f32x4 wc[NTL], wn[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) wc[t] = wp[(size_t)(t * KC) * 64];
#pragma unroll 2
    for (int kc = 0; kc < KC; ++kc) {
        if (kc + 1 < KC) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) wn[t] = wp[(size_t)(t * KC + kc + 1) * 64];
        }
        const f32x4 hv = ld4(vs + kc * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NTL; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[t][e], hv[e], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTL; ++t) wc[t] = wn[t];
    }
}

__device__ __forceinline__ f32x4 sig4(f32x4 x) {
    f32x4 r; r[0] = dep_sigmoid(x[0]); r[1] = dep_sigmoid(x[1]); r[2] = dep_sigmoid(x[2]); r[3] = dep_sigmoid(x[3]); return r;
}
__device__ __forceinline__ f32x4 tanh4(f32x4 x) {
    f32x4 r; r[0] = tanhf(x[0]); r[1] = tanhf(x[1]); r[2] = tanhf(x[2]); r[3] = tanhf(x[3]); return r;
}

// sum over the 16 lanes that share (lane>>4): lanes differ in batch row j = lane&15
__device__ __forceinline__ f32x4 rowsum16(f32x4 v) {
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], m, 64);
    }
    return v;
}

// =============================================================================== GRU forward
template <int JPW>
__global__ __launch_bounds__(512) void gru_fwd_mfma(FwdP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, LDH = H + LPAD, KC = H / 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int b = blockIdx.x * BT + j;
    const bool valid = b < p.B;
    float* hs0 = smem; float* hs1 = smem + BT * LDH;
    for (int i = threadIdx.x; i < 2 * BT * LDH; i += blockDim.x) smem[i] = 0.f;

    const f32x4* wp = p.d[0].wp + (size_t)(w * JPW * 3) * KC * 64 + lane;
    f32x4 bh[JPW][3], hprev[JPW], pool[JPW];
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int col0 = (w * JPW + jj) * 16 + q * 4;
#pragma unroll
        for (int g = 0; g < 3; ++g) bh[jj][g] = ld4(p.d[0].b_hh + g * H + col0);
        hprev[jj] = zero4(); pool[jj] = zero4();
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const float* hcur = (t & 1) ? hs1 : hs0;
        float* hnext = (t & 1) ? hs0 : hs1;
        const size_t row = (size_t)b * T + t;
        f32x4 gi[JPW][3];
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int col0 = (w * JPW + jj) * 16 + q * 4;
#pragma unroll
            for (int g = 0; g < 3; ++g)
                gi[jj][g] = valid ? ld4(p.gi + row * p.ldgi + g * H + col0) : zero4();
        }
        f32x4 acc[JPW * 3];
#pragma unroll
        for (int i = 0; i < JPW * 3; ++i) acc[i] = zero4();
        matvec_tiles<JPW * 3>(acc, wp, KC, hcur + j * LDH + q * 4);
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int col0 = (w * JPW + jj) * 16 + q * 4;
            const f32x4 r = sig4(gi[jj][0] + acc[jj * 3 + 0] + bh[jj][0]);
            const f32x4 z = sig4(gi[jj][1] + acc[jj * 3 + 1] + bh[jj][1]);
            const f32x4 hn = acc[jj * 3 + 2] + bh[jj][2];
            const f32x4 n = tanh4(gi[jj][2] + r * hn);
            const f32x4 h = (1.0f - z) * n + z * hprev[jj];
            hprev[jj] = h;
            pool[jj] += h;
            st4(hnext + j * LDH + col0, h);
            if (valid) {
                const size_t o = row * p.ldy + col0;
                st4(p.y + o, h);
                if (p.ydrop) st4(p.ydrop + o, h * dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale));
                if (p.sv0) {
                    const size_t so = row * H + col0;
                    st4(p.sv0 + so, r); st4(p.sv1 + so, z); st4(p.sv2 + so, n); st4(p.sv3 + so, hn);
                }
            }
        }
        __syncthreads();
    }
    if (valid) {
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int col0 = (w * JPW + jj) * 16 + q * 4;
            if (p.pooled) st4(p.pooled + (size_t)b * H + col0, pool[jj] * p.pool_scale);
            if (p.h_n) st4(p.h_n + (size_t)b * H + col0, hprev[jj]);
        }
    }
}

// =============================================================================== GRU backward
template <int JPW>
__global__ __launch_bounds__(512) void gru_bwd_mfma(BwdP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, G3 = 3 * H, LDG = G3 + LPAD, KC = G3 / 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int b = blockIdx.x * BT + j;
    const bool valid = b < p.B;
    float* ds0 = smem; float* ds1 = smem + BT * LDG;

    const f32x4* wp = p.d[0].wp + (size_t)(w * JPW) * KC * 64 + lane;
    f32x4 dhrec[JPW], dbr[JPW], dbz[JPW], dbn[JPW], dbh[JPW], dpl[JPW];
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int col0 = (w * JPW + jj) * 16 + q * 4;
        dhrec[jj] = (p.dh_n && valid) ? ld4(p.dh_n + (size_t)b * H + col0) : zero4();
        dpl[jj] = (p.dpooled && valid) ? ld4(p.dpooled + (size_t)b * H + col0) * p.pool_scale : zero4();
        dbr[jj] = zero4(); dbz[jj] = zero4(); dbn[jj] = zero4(); dbh[jj] = zero4();
    }

    for (int t = T - 1; t >= 0; --t) {
        float* dcur = (t & 1) ? ds1 : ds0;
        const size_t row = (size_t)b * T + t;
        f32x4 dzt[JPW];
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int col0 = (w * JPW + jj) * 16 + q * 4;
            f32x4 r = zero4(), z = zero4(), n = zero4(), hn = zero4(), hp = zero4(), d = dhrec[jj] + dpl[jj];
            if (valid) {
                const size_t so = row * H + col0;
                r = ld4(p.sv0 + so); z = ld4(p.sv1 + so); n = ld4(p.sv2 + so); hn = ld4(p.sv3 + so);
                if (t > 0) hp = ld4(p.y + (row - 1) * p.ldy + col0);
                if (p.dy) {
                    const size_t o = row * p.lddy + col0;
                    f32x4 dyv = ld4(p.dy + o);
                    if (p.drop_p > 0.f) dyv *= dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
                    d += dyv;
                }
            }
            const f32x4 dn = d * (1.0f - z) * (1.0f - n * n);
            const f32x4 dz = d * (hp - n) * z * (1.0f - z);
            const f32x4 dr = dn * hn * r * (1.0f - r);
            const f32x4 dnr = dn * r;
            dzt[jj] = d * z;
            st4(dcur + j * LDG + col0, dr);
            st4(dcur + j * LDG + H + col0, dz);
            st4(dcur + j * LDG + 2 * H + col0, dnr);
            if (valid) {
                float* g = p.dgi + row * p.lddg;
                st4(g + col0, dr); st4(g + H + col0, dz); st4(g + 2 * H + col0, dn);
                st4(p.dghn + row * H + col0, dnr);
            }
            dbr[jj] += dr; dbz[jj] += dz; dbn[jj] += dn; dbh[jj] += dnr;
        }
        __syncthreads();
        f32x4 acc[JPW];
#pragma unroll
        for (int i = 0; i < JPW; ++i) acc[i] = zero4();
        matvec_tiles<JPW>(acc, wp, KC, dcur + j * LDG + q * 4);
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) dhrec[jj] = dzt[jj] + acc[jj];
    }
    // bias-gradient partials of this workgroup: dbpart[wg][4][H]  (r, z, n_input, n_hidden)
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int col0 = (w * JPW + jj) * 16 + q * 4;
        const f32x4 s0 = rowsum16(dbr[jj]), s1 = rowsum16(dbz[jj]), s2 = rowsum16(dbn[jj]), s3 = rowsum16(dbh[jj]);
        if (j == 0) {
            float* o = p.dbpart + (size_t)blockIdx.x * 4 * H;
            st4(o + col0, s0); st4(o + H + col0, s1); st4(o + 2 * H + col0, s2); st4(o + 3 * H + col0, s3);
        }
    }
}

// =============================================================================== LSTM forward
template <int JPW>
__global__ __launch_bounds__(512) void lstm_fwd_mfma(FwdP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, LDH = H + LPAD, KC = H / 16;
    const int dir = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int b = blockIdx.x * BT + j;
    const bool valid = b < p.B;
    float* hs0 = smem; float* hs1 = smem + BT * LDH;
    for (int i = threadIdx.x; i < 2 * BT * LDH; i += blockDim.x) smem[i] = 0.f;
    const f32x4* wp = p.d[dir].wp + (size_t)(w * JPW * 4) * KC * 64 + lane;
    const int ldsg = p.dirs * 4 * H, ldsc = p.dirs * H;
    f32x4 c[JPW], hlast[JPW];
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) { c[jj] = zero4(); hlast[jj] = zero4(); }
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir ? (T - 1 - s) : s;
        const float* hcur = (s & 1) ? hs1 : hs0;
        float* hnext = (s & 1) ? hs0 : hs1;
        const size_t row = (size_t)b * T + t;
        f32x4 gi[JPW][4];
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int col0 = (w * JPW + jj) * 16 + q * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                gi[jj][g] = valid ? ld4(p.gi + row * p.ldgi + dir * 4 * H + g * H + col0) : zero4();
        }
        f32x4 acc[JPW * 4];
#pragma unroll
        for (int i = 0; i < JPW * 4; ++i) acc[i] = zero4();
        matvec_tiles<JPW * 4>(acc, wp, KC, hcur + j * LDH + q * 4);
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int col0 = (w * JPW + jj) * 16 + q * 4;
            const f32x4 ig = sig4(gi[jj][0] + acc[jj * 4 + 0]);
            const f32x4 fg = sig4(gi[jj][1] + acc[jj * 4 + 1]);
            const f32x4 gg = tanh4(gi[jj][2] + acc[jj * 4 + 2]);
            const f32x4 og = sig4(gi[jj][3] + acc[jj * 4 + 3]);
            c[jj] = fg * c[jj] + ig * gg;
            const f32x4 h = og * tanh4(c[jj]);
            hlast[jj] = h;
            st4(hnext + j * LDH + col0, h);
            if (valid) {
                const size_t o = row * p.ldy + dir * H + col0;
                st4(p.y + o, h);
                if (p.ydrop) st4(p.ydrop + o, h * dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale));
                if (p.sv0) {
                    float* gs = p.sv0 + row * ldsg + dir * 4 * H + col0;
                    st4(gs, ig); st4(gs + H, fg); st4(gs + 2 * H, gg); st4(gs + 3 * H, og);
                    st4(p.sv1 + row * ldsc + dir * H + col0, c[jj]);
                }
            }
        }
        __syncthreads();
    }
    if (valid && p.h_n) {
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int col0 = (w * JPW + jj) * 16 + q * 4;
            st4(p.h_n + ((size_t)dir * p.B + b) * H + col0, hlast[jj]);
        }
    }
}

// =============================================================================== LSTM backward
template <int JPW>
__global__ __launch_bounds__(512) void lstm_bwd_mfma(BwdP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, G4 = 4 * H, LDG = G4 + LPAD, KC = G4 / 16;
    const int dir = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int b = blockIdx.x * BT + j;
    const bool valid = b < p.B;
    float* ds0 = smem; float* ds1 = smem + BT * LDG;
    const f32x4* wp = p.d[dir].wp + (size_t)(w * JPW) * KC * 64 + lane;
    const int ldsg = p.dirs * 4 * H, ldsc = p.dirs * H;
    f32x4 dhrec[JPW], dcrec[JPW], db[JPW][4];
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int col0 = (w * JPW + jj) * 16 + q * 4;
        dhrec[jj] = (p.dh_n && valid) ? ld4(p.dh_n + ((size_t)dir * p.B + b) * H + col0) : zero4();
        dcrec[jj] = zero4();
#pragma unroll
        for (int g = 0; g < 4; ++g) db[jj][g] = zero4();
    }
    for (int s = T - 1; s >= 0; --s) {
        const int t = dir ? (T - 1 - s) : s;
        float* dcur = (s & 1) ? ds1 : ds0;
        const size_t row = (size_t)b * T + t;
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int col0 = (w * JPW + jj) * 16 + q * 4;
            f32x4 ig = zero4(), fg = zero4(), gg = zero4(), og = zero4(), ct = zero4(), cp = zero4(), d = dhrec[jj];
            if (valid) {
                const float* gs = p.sv0 + row * ldsg + dir * 4 * H + col0;
                ig = ld4(gs); fg = ld4(gs + H); gg = ld4(gs + 2 * H); og = ld4(gs + 3 * H);
                ct = ld4(p.sv1 + row * ldsc + dir * H + col0);
                if (s > 0) {
                    const size_t rowp = dir ? row + 1 : row - 1;
                    cp = ld4(p.sv1 + rowp * ldsc + dir * H + col0);
                }
                if (p.dy) {
                    const size_t o = row * p.lddy + dir * H + col0;
                    f32x4 dyv = ld4(p.dy + o);
                    if (p.drop_p > 0.f) dyv *= dep_dropmask4(p.seed, p.site, o >> 2, p.drop_p, p.drop_scale);
                    d += dyv;
                }
            }
            const f32x4 tc = tanh4(ct);
            const f32x4 dog = d * tc * og * (1.0f - og);
            const f32x4 dct = d * og * (1.0f - tc * tc) + dcrec[jj];
            const f32x4 dig = dct * gg * ig * (1.0f - ig);
            const f32x4 dfg = dct * cp * fg * (1.0f - fg);
            const f32x4 dgg = dct * ig * (1.0f - gg * gg);
            dcrec[jj] = dct * fg;
            float* dl = dcur + j * LDG + col0;
            st4(dl, dig); st4(dl + H, dfg); st4(dl + 2 * H, dgg); st4(dl + 3 * H, dog);
            if (valid) {
                float* g = p.dgi + row * p.lddg + dir * 4 * H + col0;
                st4(g, dig); st4(g + H, dfg); st4(g + 2 * H, dgg); st4(g + 3 * H, dog);
            }
            db[jj][0] += dig; db[jj][1] += dfg; db[jj][2] += dgg; db[jj][3] += dog;
        }
        __syncthreads();
        f32x4 acc[JPW];
#pragma unroll
        for (int i = 0; i < JPW; ++i) acc[i] = zero4();
        matvec_tiles<JPW>(acc, wp, KC, dcur + j * LDG + q * 4);
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) dhrec[jj] = acc[jj];
    }
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int col0 = (w * JPW + jj) * 16 + q * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 s = rowsum16(db[jj][g]);
            if (j == 0) st4(p.dbpart + ((size_t)dir * p.nwg + blockIdx.x) * 4 * H + g * H + col0, s);
        }
    }
}

// =============================================================================== generic kernels
// Any H (no alignment requirement): one workgroup per utterance, threads over hidden units,
// plain fmaf dot products against the row-major weights.  Used for shapes the MFMA tiling does
// not cover (H % 16 != 0) and as an in-library cross-check (desc.impl = 1).
constexpr int GEN_T = 128;

__global__ __launch_bounds__(GEN_T) void gru_fwd_generic(FwdP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, b = blockIdx.x;
    float* hs = smem;            // [H] h_{t-1}
    float* hnew = smem + H;      // [H]
    const float* W = p.d[0].w; const float* bh = p.d[0].b_hh;
    for (int i = threadIdx.x; i < H; i += GEN_T) hs[i] = 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)b * T + t;
        for (int i = threadIdx.x; i < H; i += GEN_T) {
            float ar = 0.f, az = 0.f, an = 0.f;
            for (int k = 0; k < H; ++k) {
                const float hv = hs[k];
                ar = fmaf(W[(size_t)i * H + k], hv, ar);
                az = fmaf(W[(size_t)(H + i) * H + k], hv, az);
                an = fmaf(W[(size_t)(2 * H + i) * H + k], hv, an);
            }
            const float* g = p.gi + row * p.ldgi;
            const float r = dep_sigmoid(g[i] + ar + bh[i]);
            const float z = dep_sigmoid(g[H + i] + az + bh[H + i]);
            const float hn = an + bh[2 * H + i];
            const float n = tanhf(g[2 * H + i] + r * hn);
            const float h = (1.0f - z) * n + z * hs[i];
            hnew[i] = h;
            const size_t o = row * p.ldy + i;
            p.y[o] = h;
            if (p.ydrop) p.ydrop[o] = h * dep_dropmask1(p.seed, p.site, o, p.drop_p, p.drop_scale);
            if (p.sv0) { const size_t so = row * H + i; p.sv0[so] = r; p.sv1[so] = z; p.sv2[so] = n; p.sv3[so] = hn; }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < H; i += GEN_T) hs[i] = hnew[i];
        __syncthreads();
    }
    if (p.h_n) for (int i = threadIdx.x; i < H; i += GEN_T) p.h_n[(size_t)b * H + i] = hs[i];
    if (p.pooled) {
        for (int i = threadIdx.x; i < H; i += GEN_T) {
            float s = 0.f;
            for (int t = 0; t < T; ++t) s += p.y[((size_t)b * T + t) * p.ldy + i];
            p.pooled[(size_t)b * H + i] = s * p.pool_scale;
        }
    }
}

__global__ __launch_bounds__(GEN_T) void gru_bwd_generic(BwdP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, b = blockIdx.x, G3 = 3 * H;
    float* dg = smem;            // [3H] dgh of this step
    float* dh = smem + G3;       // [H] recurrent dh
    float* dbp = p.dbpart + (size_t)b * 4 * H;
    const float* W = p.d[0].w;
    for (int i = threadIdx.x; i < H; i += GEN_T) dh[i] = p.dh_n ? p.dh_n[(size_t)b * H + i] : 0.f;
    for (int i = threadIdx.x; i < 4 * H; i += GEN_T) dbp[i] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)b * T + t;
        for (int i = threadIdx.x; i < H; i += GEN_T) {
            const size_t so = row * H + i;
            const float r = p.sv0[so], z = p.sv1[so], n = p.sv2[so], hn = p.sv3[so];
            const float hp = t > 0 ? p.y[(row - 1) * p.ldy + i] : 0.f;
            float d = dh[i];
            if (p.dpooled) d += p.dpooled[(size_t)b * H + i] * p.pool_scale;
            if (p.dy) {
                const size_t o = row * p.lddy + i;
                float dyv = p.dy[o];
                if (p.drop_p > 0.f) dyv *= dep_dropmask1(p.seed, p.site, o, p.drop_p, p.drop_scale);
                d += dyv;
            }
            const float dn = d * (1.0f - z) * (1.0f - n * n);
            const float dz = d * (hp - n) * z * (1.0f - z);
            const float dr = dn * hn * r * (1.0f - r);
            const float dnr = dn * r;
            dg[i] = dr; dg[H + i] = dz; dg[2 * H + i] = dnr;
            float* g = p.dgi + row * p.lddg;
            g[i] = dr; g[H + i] = dz; g[2 * H + i] = dn;
            p.dghn[row * H + i] = dnr;
            dbp[i] += dr; dbp[H + i] += dz; dbp[2 * H + i] += dn; dbp[3 * H + i] += dnr;
            dh[i] = d * z;           // own element only; the matvec term is added after the barrier
        }
        __syncthreads();
        for (int i = threadIdx.x; i < H; i += GEN_T) {
            float s = 0.f;
            for (int k = 0; k < G3; ++k) s = fmaf(dg[k], W[(size_t)k * H + i], s);
            dh[i] += s;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(GEN_T) void lstm_fwd_generic(FwdP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, b = blockIdx.x, dir = blockIdx.y;
    float* hs = smem; float* hnew = smem + H; float* cs = smem + 2 * H;
    const float* W = p.d[dir].w;
    const int ldsg = p.dirs * 4 * H, ldsc = p.dirs * H;
    for (int i = threadIdx.x; i < H; i += GEN_T) { hs[i] = 0.f; cs[i] = 0.f; }
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = dir ? (T - 1 - s) : s;
        const size_t row = (size_t)b * T + t;
        for (int i = threadIdx.x; i < H; i += GEN_T) {
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < H; ++k) {
                const float hv = hs[k];
#pragma unroll
                for (int g = 0; g < 4; ++g) a[g] = fmaf(W[(size_t)(g * H + i) * H + k], hv, a[g]);
            }
            const float* gi = p.gi + row * p.ldgi + dir * 4 * H;
            const float ig = dep_sigmoid(gi[i] + a[0]), fg = dep_sigmoid(gi[H + i] + a[1]);
            const float gg = tanhf(gi[2 * H + i] + a[2]), og = dep_sigmoid(gi[3 * H + i] + a[3]);
            const float c = fg * cs[i] + ig * gg;
            const float h = og * tanhf(c);
            cs[i] = c; hnew[i] = h;
            const size_t o = row * p.ldy + dir * H + i;
            p.y[o] = h;
            if (p.ydrop) p.ydrop[o] = h * dep_dropmask1(p.seed, p.site, o, p.drop_p, p.drop_scale);
            if (p.sv0) {
                float* gs = p.sv0 + row * ldsg + dir * 4 * H;
                gs[i] = ig; gs[H + i] = fg; gs[2 * H + i] = gg; gs[3 * H + i] = og;
                p.sv1[row * ldsc + dir * H + i] = c;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < H; i += GEN_T) hs[i] = hnew[i];
        __syncthreads();
    }
    if (p.h_n) for (int i = threadIdx.x; i < H; i += GEN_T) p.h_n[((size_t)dir * p.B + b) * H + i] = hs[i];
}

__global__ __launch_bounds__(GEN_T) void lstm_bwd_generic(BwdP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, T = p.T, b = blockIdx.x, dir = blockIdx.y, G4 = 4 * H;
    float* dg = smem; float* dh = smem + G4; float* dc = smem + G4 + H;
    const float* W = p.d[dir].w;
    const int ldsg = p.dirs * 4 * H, ldsc = p.dirs * H;
    float* dbp = p.dbpart + ((size_t)dir * p.nwg + b) * 4 * H;
    for (int i = threadIdx.x; i < H; i += GEN_T) {
        dh[i] = p.dh_n ? p.dh_n[((size_t)dir * p.B + b) * H + i] : 0.f;
        dc[i] = 0.f;
    }
    for (int i = threadIdx.x; i < 4 * H; i += GEN_T) dbp[i] = 0.f;
    __syncthreads();
    for (int s = T - 1; s >= 0; --s) {
        const int t = dir ? (T - 1 - s) : s;
        const size_t row = (size_t)b * T + t;
        for (int i = threadIdx.x; i < H; i += GEN_T) {
            const float* gs = p.sv0 + row * ldsg + dir * 4 * H;
            const float ig = gs[i], fg = gs[H + i], gg = gs[2 * H + i], og = gs[3 * H + i];
            const float ct = p.sv1[row * ldsc + dir * H + i];
            float cp = 0.f;
            if (s > 0) { const size_t rowp = dir ? row + 1 : row - 1; cp = p.sv1[rowp * ldsc + dir * H + i]; }
            float d = dh[i];
            if (p.dy) {
                const size_t o = row * p.lddy + dir * H + i;
                float dyv = p.dy[o];
                if (p.drop_p > 0.f) dyv *= dep_dropmask1(p.seed, p.site, o, p.drop_p, p.drop_scale);
                d += dyv;
            }
            const float tc = tanhf(ct);
            const float dog = d * tc * og * (1.0f - og);
            const float dct = d * og * (1.0f - tc * tc) + dc[i];
            const float dig = dct * gg * ig * (1.0f - ig);
            const float dfg = dct * cp * fg * (1.0f - fg);
            const float dgg = dct * ig * (1.0f - gg * gg);
            dc[i] = dct * fg;
            dg[i] = dig; dg[H + i] = dfg; dg[2 * H + i] = dgg; dg[3 * H + i] = dog;
            float* g = p.dgi + row * p.lddg + dir * 4 * H;
            g[i] = dig; g[H + i] = dfg; g[2 * H + i] = dgg; g[3 * H + i] = dog;
            dbp[i] += dig; dbp[H + i] += dfg; dbp[2 * H + i] += dgg; dbp[3 * H + i] += dog;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < H; i += GEN_T) {
            float sum = 0.f;
            for (int k = 0; k < G4; ++k) sum = fmaf(dg[k], W[(size_t)k * H + i], sum);
            dh[i] = sum;
        }
        __syncthreads();
    }
}

// =============================================================================== weight packing
// fwd: wp[((jt*G + g)*KC + kc)*256 + l*4 + e]  = W[(g*H + jt*16 + (l&15))*H + kc*16 + (l>>4)*4 + e],  KC = H/16
// bwd: wpT[(jt*KC2 + kc)*256 + l*4 + e]        = W[(kc*16 + (l>>4)*4 + e)*H + jt*16 + (l&15)],        KC2 = G*H/16
__global__ void pack_whh_kernel(const float* __restrict__ W, float* __restrict__ wp, float* __restrict__ wpT, int G, int H) {
    const long n = (long)G * H * H;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int e = idx & 3, l = (idx >> 2) & 63;
    const long blk = idx >> 8;
    {
        const int KC = H / 16;
        const int kc = blk % KC; const long tg = blk / KC;
        const int g = tg % G; const int jt = tg / G;
        wp[idx] = W[(size_t)(g * H + jt * 16 + (l & 15)) * H + kc * 16 + (l >> 4) * 4 + e];
    }
    {
        const int KC2 = G * H / 16;
        const int kc = blk % KC2; const int jt = blk / KC2;
        wpT[idx] = W[(size_t)(kc * 16 + (l >> 4) * 4 + e) * H + jt * 16 + (l & 15)];
    }
}

__global__ void finish_db_kernel(const float* __restrict__ part, int nrows, int H, int G, int cell,
                                 float* db_ih, float* db_hh) {
    // part: [nrows][4][H].  GRU (G=3): db_ih = [r,z,n_in], db_hh = [r,z,n_hid].  LSTM (G=4): both = gates.
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * H) return;
    float s = 0.f;
    int r = 0;
    for (; r + 8 <= nrows; r += 8) {                          // eight loads in flight, added in row order (same sum as a plain loop)
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(r + k) * 4 * H + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; r < nrows; ++r) s += part[(size_t)r * 4 * H + i];
    const int g = i / H, c = i % H;
    if (cell == DEP_CELL_GRU) {
        if (g < 2) { db_ih[g * H + c] = s; db_hh[g * H + c] = s; }
        else if (g == 2) db_ih[2 * H + c] = s;
        else db_hh[2 * H + c] = s;
    } else {
        db_ih[i] = s; db_hh[i] = s;
    }
}

int pick_jpw(int H, int* nw) {
    if (H % 16) return 0;
    const int tiles = H / 16;
    for (int w = 8; w >= 1; w >>= 1) {
        if (tiles % w == 0 && tiles / w <= 4) { *nw = w; return tiles / w; }
    }
    return 0;
}

}  // namespace

bool dep_sweep_use_mfma(int H, int impl) {
    if (impl == 1) return false;
    int nw; const int jpw = pick_jpw(H, &nw);
    if (!jpw) return false;
    // backward LDS: 2 * 16 * (4H + 4) floats must fit the 160 KiB LDS
    return (size_t)2 * BT * (4 * H + LPAD) * sizeof(float) <= 160 * 1024;
}

int dep_sweep_num_wg(int B, int H, int impl) { return dep_sweep_use_mfma(H, impl) ? dep_cdiv(B, BT) : B; }

size_t dep_pack_floats(int G, int H) { return (size_t)G * H * H; }

int dep_pack_whh(const float* w_hh, float* wp, float* wpT, int G, int H, hipStream_t s) {
    const long n = (long)G * H * H;
    DEP_LAUNCH(pack_whh_kernel, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, w_hh, wp, wpT, G, H);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

#define LAUNCH_JPW(kern, grid, nthr, lds, s, P)                                                    \
    do {                                                                                           \
        switch (jpw) {                                                                             \
            case 1: (void)hipFuncSetAttribute((const void*)kern<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds)); \
                    DEP_LAUNCH(kern<1>, grid, dim3(nthr), lds, s, P); break;               \
            case 2: (void)hipFuncSetAttribute((const void*)kern<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds)); \
                    DEP_LAUNCH(kern<2>, grid, dim3(nthr), lds, s, P); break;               \
            case 3: (void)hipFuncSetAttribute((const void*)kern<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds)); \
                    DEP_LAUNCH(kern<3>, grid, dim3(nthr), lds, s, P); break;               \
            default: (void)hipFuncSetAttribute((const void*)kern<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds)); \
                    DEP_LAUNCH(kern<4>, grid, dim3(nthr), lds, s, P); break;               \
        }                                                                                          \
    } while (0)

int dep_launch_sweep_fwd(const dep_sweep_args& a) {
    const int G = a.cell == DEP_CELL_GRU ? 3 : 4;
    FwdP p{};
    p.B = a.B; p.T = a.T; p.H = a.H; p.dirs = a.dirs;
    for (int d = 0; d < a.dirs; ++d) { p.d[d].wp = (const f32x4*)a.wp[d]; p.d[d].w = a.w_hh[d]; p.d[d].b_hh = a.b_hh[d]; }
    p.gi = a.gi; p.ldgi = a.dirs * G * a.H; p.y = a.y; p.ldy = a.ldy;
    p.ydrop = (a.drop_p > 0.f) ? a.ydrop : nullptr;
    p.drop_p = a.drop_p; p.drop_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f; p.seed = a.seed; p.site = a.site;
    p.pooled = a.pooled; p.pool_scale = a.pool_scale; p.h_n = a.h_n;
    p.sv0 = a.training ? a.sv0 : nullptr; p.sv1 = a.sv1; p.sv2 = a.sv2; p.sv3 = a.sv3;
    DEP_CHECK_ARG(a.y && a.gi);
    DepProfScope prof(a.cell == DEP_CELL_GRU ? DEP_PROF_GRU_FWD : DEP_PROF_LSTM_FWD, a.stream);
    if (dep_sweep_use_mfma(a.H, a.impl)) {
        int nw = 0; const int jpw = pick_jpw(a.H, &nw);
        const size_t lds = (size_t)2 * BT * (a.H + LPAD) * sizeof(float);
        dim3 grid(dep_cdiv(a.B, BT), a.dirs);
        if (a.cell == DEP_CELL_GRU) LAUNCH_JPW(gru_fwd_mfma, grid, nw * 64, lds, a.stream, p);
        else LAUNCH_JPW(lstm_fwd_mfma, grid, nw * 64, lds, a.stream, p);
    } else {
        dim3 grid(a.B, a.dirs);
        const size_t lds = (size_t)3 * a.H * sizeof(float);
        if (a.cell == DEP_CELL_GRU) DEP_LAUNCH(gru_fwd_generic, grid, dim3(GEN_T), lds, a.stream, p);
        else DEP_LAUNCH(lstm_fwd_generic, grid, dim3(GEN_T), lds, a.stream, p);
    }
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

int dep_launch_sweep_bwd(const dep_sweep_bwd_args& a) {
    const int G = a.cell == DEP_CELL_GRU ? 3 : 4;
    BwdP p{};
    p.B = a.B; p.T = a.T; p.H = a.H; p.dirs = a.dirs;
    for (int d = 0; d < a.dirs; ++d) { p.d[d].wp = (const f32x4*)a.wpT[d]; p.d[d].w = a.w_hh[d]; p.d[d].b_hh = nullptr; }
    p.y = a.y; p.ldy = a.ldy; p.dy = a.dy; p.lddy = a.lddy;
    p.drop_p = a.dy ? a.drop_p : 0.f; p.drop_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    p.seed = a.seed; p.site = a.site;
    p.dpooled = a.dpooled; p.pool_scale = a.pool_scale; p.dh_n = a.dh_n;
    p.sv0 = a.sv0; p.sv1 = a.sv1; p.sv2 = a.sv2; p.sv3 = a.sv3;
    p.dgi = a.dgi; p.lddg = a.dirs * G * a.H; p.dghn = a.dghn; p.dbpart = a.dbpart;
    p.nwg = dep_sweep_num_wg(a.B, a.H, a.impl);
    DEP_CHECK_ARG(a.dbpart_rows >= p.nwg * a.dirs);
    DEP_CHECK_ARG(a.sv0 && a.dgi && a.dbpart);
    DepProfScope prof(a.cell == DEP_CELL_GRU ? DEP_PROF_GRU_BWD : DEP_PROF_LSTM_BWD, a.stream);
    if (dep_sweep_use_mfma(a.H, a.impl)) {
        int nw = 0; const int jpw = pick_jpw(a.H, &nw);
        const size_t lds = (size_t)2 * BT * (G * a.H + LPAD) * sizeof(float);
        dim3 grid(dep_cdiv(a.B, BT), a.dirs);
        if (a.cell == DEP_CELL_GRU) LAUNCH_JPW(gru_bwd_mfma, grid, nw * 64, lds, a.stream, p);
        else LAUNCH_JPW(lstm_bwd_mfma, grid, nw * 64, lds, a.stream, p);
    } else {
        dim3 grid(a.B, a.dirs);
        const size_t lds = (size_t)(G * a.H + 2 * a.H) * sizeof(float);
        if (a.cell == DEP_CELL_GRU) DEP_LAUNCH(gru_bwd_generic, grid, dim3(GEN_T), lds, a.stream, p);
        else DEP_LAUNCH(lstm_bwd_generic, grid, dim3(GEN_T), lds, a.stream, p);
    }
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}

int dep_finish_db(const dep_sweep_bwd_args& a, float* const* db_ih, float* const* db_hh) {
    const int G = a.cell == DEP_CELL_GRU ? 3 : 4;
    DEP_CHECK_ARG(a.dirs >= 1 && a.dbpart_rows >= a.dirs && a.dbpart_rows % a.dirs == 0);
    const int nwg = a.dbpart_rows / a.dirs;           // rows per direction the sweep that ran has written
    for (int d = 0; d < a.dirs; ++d) {
        DEP_LAUNCH(finish_db_kernel, dim3(dep_cdiv(4 * a.H, 128)), dim3(128), 0, a.stream,
                           a.dbpart + (size_t)d * nwg * 4 * a.H, nwg, a.H, G, a.cell, db_ih[d], db_hh[d]);
        DEP_CHECK_LAUNCH();
    }
    return DEP_OK;
}
