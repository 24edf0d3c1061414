// C-ABI orchestration of the stacked GRU / BiLSTM operator (dep_rnn_forward / dep_rnn_backward):
// per layer  [pack W_hh] -> input-projection GEMM -> persistent sweep ; backward mirrors it.
// See include/dep_rnn.h for the contract and the reference call sites each entry replaces.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>
#include "dep_common.h"

static thread_local char g_err[512] = "";

void dep_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dep_last_error(void) { return g_err; }
extern "C" int dep_version(void) { return 100; }
extern "C" const char* dep_arch(void) { return "gfx950"; }

// ---- event-based kernel timing -------------------------------------------------------
// Process-wide recorder, safe to use from several host threads / streams at once (SURVEY 8b "reentrant per stream"): the open
// begin/end pair is per thread, the event pool and the record list are guarded by one mutex (taken only while profiling is on).
namespace {
struct ProfRec { hipEvent_t a, b; int cat; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfRec> g_recs;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
thread_local ProfRec g_cur;
}  // namespace
bool dep_prof_on() { return g_prof_on.load(std::memory_order_relaxed); }
void dep_prof_begin(int cat, hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_pool.empty()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            g_pool.push_back({a, b});
        }
        g_cur.a = g_pool.back().first; g_cur.b = g_pool.back().second; g_cur.cat = cat;
        g_pool.pop_back();
    }
    (void)hipEventRecord(g_cur.a, s);
}
void dep_prof_end(hipStream_t s) {
    (void)hipEventRecord(g_cur.b, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_recs.push_back(g_cur);
}
extern "C" int dep_profile_enable(int on) { g_prof_on.store(on != 0); return DEP_OK; }
// Sums the recorded launch durations per category (ms) and resets; blocks until the events completed.
extern "C" int dep_profile_read(double* total_ms, int* counts, int ncat) {
    for (int i = 0; i < ncat; ++i) { total_ms[i] = 0.0; counts[i] = 0; }
    std::vector<ProfRec> recs;
    { std::lock_guard<std::mutex> lk(g_prof_mu); recs.swap(g_recs); }
    for (auto& r : recs) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        if (r.cat < ncat) { total_ms[r.cat] += ms; counts[r.cat] += 1; }
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : recs) g_pool.push_back({r.a, r.b});
    return DEP_OK;
}

// ---- launch-instance log (DEP_LAUNCH, dep_common.h) ------------------------------------------------
namespace {
std::atomic<bool> g_ilog_on{false};
std::mutex g_ilog_mu;
std::set<std::string> g_ilog;
}  // namespace
std::atomic<int> g_ilog_any{0};          // bit 0: instance log, bit 1: enqueue-order log (one relaxed load per launch when both are off)
bool dep_ilog_on() { return g_ilog_any.load(std::memory_order_relaxed) != 0; }
namespace { std::vector<std::string> g_olog; }
void dep_olog_add(char kind, const char* text, long n) {
    if (!(g_ilog_any.load(std::memory_order_relaxed) & 2)) return;
    std::string e(1, kind); e += ' '; e += text;
    if (n >= 0) { e += " n="; e += std::to_string(n); }
    std::lock_guard<std::mutex> lk(g_ilog_mu);
    g_olog.push_back(std::move(e));
}
extern "C" int dep_order_log_enable(int on) {
    std::lock_guard<std::mutex> lk(g_ilog_mu);
    if (on) { g_olog.clear(); g_ilog_any.fetch_or(2); } else g_ilog_any.fetch_and(~2);
    return DEP_OK;
}
extern "C" int dep_order_log_note(const char* text) { if (!text) return DEP_ERR_ARG; dep_olog_add('N', text, -1); return DEP_OK; }
extern "C" long dep_order_log_read(char* buf, long cap, int reset) {
    std::lock_guard<std::mutex> lk(g_ilog_mu);
    std::string all;
    for (const auto& s : g_olog) { all += s; all += '\n'; }
    if (buf && cap > 0) {
        const long n = (long)all.size() < cap - 1 ? (long)all.size() : cap - 1;
        memcpy(buf, all.data(), (size_t)n); buf[n] = 0;
    }
    if (reset) g_olog.clear();
    return (long)all.size() + 1;
}
void dep_ilog_note(const char* kern, const char* where) {
    dep_olog_add('K', kern, -1);
    if (!g_ilog_on.load(std::memory_order_relaxed)) return;
    std::string k(kern);
    // the kernel as written is enough when it names every template argument; launchers that are templates themselves
    // (kernel<TA, TB, ..>) are told apart by their own signature
    bool symbolic = false;
    for (size_t i = 0; i + 1 < k.size(); ++i)
        if ((k[i] == '<' || k[i] == ' ' ) && k[i + 1] >= 'A' && k[i + 1] <= 'Z') symbolic = true;
    if (symbolic) { k += " @ "; k += where; }
    std::lock_guard<std::mutex> lk(g_ilog_mu);
    g_ilog.insert(std::move(k));
}
extern "C" int dep_instance_log_enable(int on) {
    std::lock_guard<std::mutex> lk(g_ilog_mu);
    if (on) { g_ilog.clear(); g_ilog_any.fetch_or(1); } else g_ilog_any.fetch_and(~1);
    g_ilog_on.store(on != 0);
    return DEP_OK;
}
// Newline-separated distinct launch instances recorded so far -> buf (NUL-terminated, truncated to cap); returns the bytes the
// full list needs (incl. the NUL).  reset != 0 empties the log afterwards.
extern "C" long dep_instance_log_read(char* buf, long cap, int reset) {
    std::lock_guard<std::mutex> lk(g_ilog_mu);
    std::string all;
    for (const auto& s : g_ilog) { all += s; all += '\n'; }
    if (buf && cap > 0) {
        const long n = (long)all.size() < cap - 1 ? (long)all.size() : cap - 1;
        memcpy(buf, all.data(), (size_t)n); buf[n] = 0;
    }
    if (reset) g_ilog.clear();
    return (long)all.size() + 1;
}

namespace {

constexpr int MAXL = 8;

struct Layout {
    int G, D, L;
    size_t BT;                       // B*T rows
    // reserve (float offsets)
    size_t y[MAXL], ydrop[MAXL], sv[MAXL][4], wp[MAXL][2], wpT[MAXL][2];
    size_t reserve_floats;
    // workspace (float offsets)
    size_t gi, dghn, dx[2], dbpart, biastmp, gemm, xbuf;
    size_t wstack[MAXL], bstack[MAXL], dwstack;   // bidirectional: both directions' W_ih / (b_ih + b_hh) stacked (2 G H x in), one dW_ih scratch
    size_t gi2, dghn2, dbpart2;      // second set of gate-gradient buffers: the fused backward keeps both layers' dgi / dghn
    size_t gemm_bytes, xbuf_bytes, ws_floats;
    int nwg;
    bool drop;
    bool cluster;                    // cluster-parallel sweeps (rnn_cluster*.hip)
    bool cluster16;                  // forward with 16-unit members, two workgroups per CU (rnn_cluster16.hip)
    bool dg4;                        // GRU cluster backward: gate gradients as ONE (B*T, 4H) array [dr | dz | dn | dn*r] (dW_hh is then one contraction)
    bool bf16st;                     // dep_set_gemm_mode(3) on a stack whose kernels have the bf16-storage variants (2-layer GRU, H = 256, fused forward): y, hn as bf16, gate gradients as PKH
    bool sv16;                       // GRU cluster sweeps: saved gates r, z, n as 16-bit fixed point (split-precision mode only; decided per call)
    bool fused2;                     // 2-layer GRU, H = 256: both layers in one launch (rnn_fused2.hip), split-precision mode only
    size_t wih_img;                  // workspace: packed W_ih of layer 1 for the fused forward
};

bool make_layout(const dep_rnn_desc* d, Layout& lo) {
    if (!d || d->B <= 0 || d->T <= 0 || d->F <= 0 || d->H <= 0 || d->L < 1 || d->L > MAXL) return false;
    if (d->cell == DEP_CELL_GRU) { if (d->dirs != 1) return false; }
    else if (d->cell == DEP_CELL_LSTM) { if (d->dirs != 1 && d->dirs != 2) return false; }
    else return false;
    if (d->dropout_p < 0.f || d->dropout_p >= 1.f) return false;
    lo.G = d->cell == DEP_CELL_GRU ? 3 : 4; lo.D = d->dirs; lo.L = d->L;
    lo.BT = (size_t)d->B * d->T;
    lo.drop = d->training && d->dropout_p > 0.f;
    const size_t H = d->H, D = d->dirs, G = lo.G;
    auto al = [](size_t f) { return (f + 63) / 64 * 64; };
    size_t off = 0;
    for (int l = 0; l < d->L; ++l) {
        lo.y[l] = off; off += al(lo.BT * D * H);
        lo.ydrop[l] = off; if (lo.drop && l < d->L - 1) off += al(lo.BT * D * H);
        if (d->training) {
            if (d->cell == DEP_CELL_GRU) { for (int k = 0; k < 4; ++k) { lo.sv[l][k] = off; off += al(lo.BT * H); } }
            else { lo.sv[l][0] = off; off += al(lo.BT * D * 4 * H); lo.sv[l][1] = off; off += al(lo.BT * D * H); lo.sv[l][2] = lo.sv[l][3] = 0; }
        }
        for (size_t dd = 0; dd < D; ++dd) {
            lo.wp[l][dd] = off; off += al(G * H * H);
            lo.wpT[l][dd] = off; off += al(G * H * H);
        }
    }
    // bidirectional stacks: the two directions share their input, so their input projections, dX and dW_ih are ONE contraction
    // each over stacked weights [W_ih(fwd); W_ih(bwd)] (2 G H x in) -- the input (629 MB at cfg3's layer 0) is read once, and
    // dX needs no read-modify-write.  The stacked copies live in the reserve (the backward's dX reads them again).
    for (int l = 0; l < d->L; ++l) {
        lo.wstack[l] = lo.bstack[l] = 0;
        if (d->dirs == 2) {
            const size_t in = l == 0 ? (size_t)d->F : D * H;
            lo.wstack[l] = off; off += al(D * G * H * in);
            lo.bstack[l] = off; off += al(D * G * H);
        }
    }
    lo.reserve_floats = off;
    // workspace
    // rows of bias-gradient partials: one per 16-utterance tile for the tile-MFMA and the cluster sweeps, one per utterance for
    // the generic sweep.  (Sized for the larger; dep_finish_db sums the rows the sweep that ran has written: dbpart_rows.)
    lo.nwg = dep_sweep_num_wg(d->B, d->H, d->impl);
    size_t w = 0;
    lo.gi = w; w += al(lo.BT * D * (G + (d->cell == DEP_CELL_GRU && d->training ? 1 : 0)) * H);     // GI (fwd) / dGI (bwd; GRU: room for the 4H-wide [dr|dz|dn|dn*r] rows)
    lo.dghn = w; w += al(lo.BT * H);
    const size_t maxin = D * H > (size_t)d->F ? D * H : (size_t)d->F;
    lo.dx[0] = w; w += al(lo.BT * D * H);
    lo.dx[1] = w; w += al(lo.BT * D * H);
    (void)maxin;
    lo.dbpart = w; w += al((size_t)D * lo.nwg * 4 * H);
    lo.biastmp = w; w += al(G * H);
    lo.dwstack = w; if (d->dirs == 2 && d->training) { const size_t mx = D * H > (size_t)d->F ? D * H : (size_t)d->F; w += al(D * G * H * mx); }
    // split-K scratch: the largest weight-gradient contraction
    size_t gb = 0;
    {   // every (rows, cols) block dep_rnn_backward contracts over B*T: dW_ih (G H x F | D H), dW_hh whole or as the GRU's
        // (2H x H) + (H x H) pair -- the split count depends on the block shape, so take the maximum over all of them
        const int Ms[4] = {(int)(D * G * H), (int)(G * H), (int)(2 * H), (int)H};      // (D G H: the direction-stacked dW_ih of a bidirectional stack)
        const int Ns[3] = {d->F, (int)(D * H), (int)H};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 3; ++j) {
                const size_t b1 = dep_gemm_workspace_bytes(1, 0, Ms[i], Ns[j], (int)lo.BT);
                if (b1 > gb) gb = b1;
            }
    }
    if (d->training && ((d->cell == DEP_CELL_GRU && d->dirs == 1) || (d->cell == DEP_CELL_LSTM && d->dirs == 2))) {
        // dW_ih + dW_hh of a GRU layer / dW_hh of both directions of a BiLSTM layer as one launch (dep_gemm_tn_pair): two sets of partials
        const size_t b2 = 2 * dep_gemm_workspace_bytes(1, 0, (int)(G * H), (int)H, (int)lo.BT);
        if (b2 > gb) gb = b2;
    }
    {   // ... and the forward's use of the same scratch: the stage image of a layer's (direction-stacked) W_ih (gemm_bf16x3_nt_dma)
        const size_t mx = D * H > (size_t)d->F ? D * H : (size_t)d->F;
        const size_t b3 = (size_t)D * G * H * mx * sizeof(float);
        if (b3 > gb) gb = b3;
    }
    lo.gemm = w; lo.gemm_bytes = gb; w += al(gb / sizeof(float) + 64);
    // impl: 0 auto (cluster > tile-MFMA > generic), 1 generic, 2 tile-MFMA, 3 cluster (must be supported)
    const bool cok = d->cell == DEP_CELL_GRU ? dep_cluster_ok(d->cell, d->H, d->B, d->dirs) : dep_cluster_lstm_ok(d->H, d->B, d->dirs);
    if (d->impl == 3 && !cok) return false;
    lo.cluster = cok && (d->impl == 0 || d->impl == 3);
    lo.cluster16 = lo.cluster && dep_cluster16_ok(d->cell, d->H, d->B);
    lo.dg4 = lo.cluster && d->cell == DEP_CELL_GRU && d->dirs == 1;
    lo.xbuf = w; lo.xbuf_bytes = !lo.cluster ? 0 : (d->cell == DEP_CELL_GRU ? dep_cluster_xbuf_bytes(d->cell, d->H, d->B, d->dirs)
                                                                : dep_cluster_lstm_xbuf_bytes(d->H, d->B, d->dirs));
    if (lo.cluster) lo.nwg = dep_cdiv(d->B, 16);      // the cluster sweeps write one row per tile whatever the tile-MFMA sweep could do at this H
    lo.fused2 = lo.cluster && dep_fused2_ok(d->cell, d->H, d->L, d->dirs);
    if (lo.fused2) {
        size_t fb = dep_fused2_xbuf_bytes(d->B); if (fb > lo.xbuf_bytes) lo.xbuf_bytes = fb;
        if (d->training) { fb = dep_fused2_bwd_xbuf_bytes(d->B); if (fb > lo.xbuf_bytes) lo.xbuf_bytes = fb; }
    }
    w += al(lo.xbuf_bytes / sizeof(float) + 64);
    lo.wih_img = w; if (lo.fused2) w += al(G * H * H);
    lo.gi2 = lo.dghn2 = lo.dbpart2 = 0;
    if (lo.fused2 && d->training) {
        lo.gi2 = w; w += al(lo.BT * (G + 1) * H);      // (room for the 4H-wide [dr | dz | dn | dn*r] rows, like lo.gi)
        lo.dghn2 = w; w += al(lo.BT * H);
        lo.dbpart2 = w; w += al((size_t)lo.nwg * 4 * H);
    }
    lo.ws_floats = w;
    // 16-bit saved gates: the kernels that implement them are the fused forward, the 32-unit-member forward and the 32-unit-member
    // backward (burst or not); the 16-unit-member kernels and the opt-in fused backward read / write fp32 gates
    {
        static int sv_env = -1;
        if (sv_env < 0) { const char* e = getenv("DEP_SV16"); sv_env = (e && e[0] == '0') ? 0 : 1; }
        lo.sv16 = sv_env && d->training && lo.cluster &&
                  (d->cell == DEP_CELL_GRU ? (lo.fused2 || !lo.cluster16) : dep_cluster_lstm_sv16_ok());
        // bf16-STORAGE mode (dep_set_gemm_mode(3); a labelled throughput mode, never the parity path): only where every kernel of the
        // stack has the variant -- the fused 2-layer GRU forward and the burst backward with the 4H-wide gate-gradient rows.  Other
        // stacks run mode 3 exactly like mode 2 (single bf16 products, fp32 storage).
        lo.bf16st = dep_get_gemm_mode() == 3 && lo.sv16 && lo.fused2 && lo.dg4 && d->cell == DEP_CELL_GRU && d->T % 2 == 0 &&
                    dep_cluster_bwd_pk_ok(d->H, d->T);
    }
    return true;
}

}  // namespace

extern "C" size_t dep_rnn_reserve_bytes(const dep_rnn_desc* d) {
    Layout lo;
    if (!make_layout(d, lo)) return 0;
    return lo.reserve_floats * sizeof(float);
}
extern "C" size_t dep_rnn_workspace_bytes(const dep_rnn_desc* d) {
    Layout lo;
    if (!make_layout(d, lo)) return 0;
    return lo.ws_floats * sizeof(float);
}
// byte offset of layer l's output sequence (B,T,H*dirs) inside the reserve (zero-copy access for the caller)
extern "C" size_t dep_rnn_reserve_y_offset(const dep_rnn_desc* d, int layer) {
    Layout lo;
    if (!make_layout(d, lo) || layer < 0 || layer >= d->L) return (size_t)-1;
    return lo.y[layer] * sizeof(float);
}
extern "C" size_t dep_rnn_reserve_ydrop_offset(const dep_rnn_desc* d, int layer) {
    Layout lo;
    if (!make_layout(d, lo) || layer < 0 || layer >= d->L - 1 || !lo.drop) return (size_t)-1;
    return lo.ydrop[layer] * sizeof(float);
}

// Byte offset of the cluster exchange buffer inside the workspace (debug tooling: DEP_TRACE stamps live at +4096).
extern "C" size_t dep_rnn_workspace_xbuf_offset(const dep_rnn_desc* d) {
    Layout lo;
    if (!make_layout(d, lo) || !lo.cluster) return (size_t)-1;
    return lo.xbuf * sizeof(float);
}

// Status of the cluster sweeps that ran on (workspace): 0 ok, 2 = a bounded spin gave up (a cluster member was
// not resident or died).  Synchronises the stream.
extern "C" int dep_rnn_status(const dep_rnn_desc* d, void* workspace, void* stream) {
    Layout lo;
    DEP_CHECK_ARG(make_layout(d, lo) && workspace);
    if (!lo.cluster) return DEP_OK;
    unsigned st = 0;
    if (hipMemcpyAsync(&st, (float*)workspace + lo.xbuf, sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        dep_set_error("dep_rnn_status: HIP copy failed"); return DEP_ERR_HIP;
    }
    if (st != 0) { dep_set_error("cluster sweep gave up waiting for a member (status %u)", st); return DEP_ERR_HIP; }
    return DEP_OK;
}

// Kernels that need every CU to themselves (the fused two-layer GRU forward, the 16-unit-member forward): allowed unless the
// process said otherwise (DEP_EXCLUSIVE=0 / dep_rnn_set_exclusive(0): the GPU is shared with other streams or processes).
static std::atomic<int> g_exclusive{-1};
bool dep_exclusive_on() {
    int v = g_exclusive.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = getenv("DEP_EXCLUSIVE"); v = (e && e[0] == '0') ? 0 : 1; g_exclusive.store(v); }
    return v == 1;
}
extern "C" int dep_rnn_set_exclusive(int on) { g_exclusive.store(on ? 1 : 0); return DEP_OK; }
extern "C" int dep_rnn_get_exclusive(void) { return dep_exclusive_on() ? 1 : 0; }

// Precision of the recurrent products inside the cluster sweeps: follows the GEMM mode (include/dep_rnn.h,
// dep_set_gemm_mode): 1 = 3-term bf16 split on the bf16 matrix cores, 0 = exact fp32 MFMA.
static bool sweep_split_mode() {
    return dep_get_gemm_mode() >= 1;                // mode 2 (single bf16 products in the GEMMs) keeps the split sweeps
}

// Precision mode of the packed recurrent-weight images a reserve holds (host-side record, no device traffic): the
// backward must run the kernels of the SAME mode, so dep_rnn_backward refuses a reserve whose forward ran in the other
// mode (a caller flipping dep_set_gemm_mode in between would otherwise get silently wrong gradients).  Small ring: the
// newest record of a pointer wins; a reserve with no record (evicted after 256 other forwards) is trusted.
namespace {
struct ModeRec { const void* p; int mode; };
ModeRec g_modes[256];
int g_mode_next = 0;
std::mutex g_mode_mu;
void record_reserve_mode(const void* reserve, int mode) {
    std::lock_guard<std::mutex> lk(g_mode_mu);
    for (auto& r : g_modes) if (r.p == reserve) { r.mode = mode; return; }
    g_modes[g_mode_next] = {reserve, mode};
    g_mode_next = (g_mode_next + 1) % 256;
}
int lookup_reserve_mode(const void* reserve) {
    std::lock_guard<std::mutex> lk(g_mode_mu);
    for (auto& r : g_modes) if (r.p == reserve) return r.mode;
    return -1;
}
}  // namespace

extern "C" int dep_rnn_forward(const dep_rnn_desc* d, const float* x, const float* const* weights, float* y,
                               float* pooled, float* h_n, void* reserve, size_t reserve_bytes, void* workspace,
                               size_t workspace_bytes, void* stream) {
    Layout lo;
    DEP_CHECK_ARG(make_layout(d, lo));
    DEP_CHECK_ARG(x && weights && reserve && workspace);
    DEP_CHECK_ARG(!(pooled && (d->cell != DEP_CELL_GRU || d->pool == DEP_POOL_NONE)));
    if (reserve_bytes < lo.reserve_floats * sizeof(float) || workspace_bytes < lo.ws_floats * sizeof(float)) {
        dep_set_error("dep_rnn_forward: reserve/workspace too small (%zu/%zu given, %zu/%zu needed)", reserve_bytes,
                      workspace_bytes, lo.reserve_floats * sizeof(float), lo.ws_floats * sizeof(float));
        return DEP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* R = (float*)reserve; float* W = (float*)workspace;
    const int B = d->B, T = d->T, H = d->H, D = d->dirs, G = lo.G, L = d->L;
    const int BTr = (int)lo.BT;
    // the unsplit projections' scratch (the weight's stage image of gemm_bf16x3_nt_dma): the split-K region, idle during the forward
    struct ScratchGuard { ScratchGuard(void* q, size_t n) { dep_gemm_set_scratch(q, n); } ~ScratchGuard() { dep_gemm_set_scratch(nullptr, 0); } } scratch_guard(W + lo.gemm, lo.gemm_bytes);
    const bool mfma = lo.cluster || dep_sweep_use_mfma(H, d->impl);
    const bool excl = dep_exclusive_on();
    const bool use16 = lo.cluster16 && excl;                        // the 16-unit-member forward fills the CUs: not on a shared GPU
    const bool split_fwd = use16 && sweep_split_mode();
    const bool split_fwd32 = lo.cluster && !use16 && d->cell == DEP_CELL_GRU && sweep_split_mode();     // 32-unit members
    const bool split_lstm = lo.cluster && d->cell == DEP_CELL_LSTM && sweep_split_mode();
    int rc;
    if (lo.cluster) { rc = dep_cluster_reset_status(W + lo.xbuf, s); if (rc) return rc; }
    // the recurrent-weight images packed below are precision-mode specific: remember which mode this reserve holds
    // bit 0: precision mode; bit 1: the backward image is the 16-unit-member one (a caller flipping DEP_CLUSTER16_BWD is refused too)
    const bool sv16 = lo.sv16 && sweep_split_mode();      // (exact-fp32 mode keeps fp32 gates: its 16-unit-member forward has no 16-bit path)
    if (lo.bf16st && (!excl || y)) {
        dep_set_error("dep_rnn_forward: bf16-storage mode (dep_set_gemm_mode(3)) runs the exclusive fused forward only (dep_rnn_set_exclusive(1)) "
                      "and has no fp32 copy of the output sequence (y must be NULL)");
        return DEP_ERR_ARG;
    }
    record_reserve_mode(reserve, (sweep_split_mode() ? 1 : 0) | (sv16 ? 4 : 0) | (lo.bf16st ? 8 : 0));
    if (lo.fused2 && sweep_split_mode() && excl) {
        // both layers in one launch: layer 1 runs one step behind layer 0 and takes its input straight from the exchanged
        // h0_t (no layer-1 input-projection GEMM, no GI round trip through HBM for it)
        const float* const* w0 = weights; const float* const* w1 = weights + 4;
        for (int k = 0; k < 4; ++k) DEP_CHECK_ARG(w0[k] && w1[k]);
        {   // every weight image of the step in one launch: W_hh(l0), W_hh(l1), W_ih(l1) forward images; the backward's two
            const float* srcs[5] = {w0[1], w1[1], w1[0], w0[1], w1[1]};
            float* dsts[5] = {R + lo.wp[0][0], R + lo.wp[1][0], W + lo.wih_img, R + lo.wpT[0][0], R + lo.wpT[1][0]};
            const int kinds[5] = {0, 0, 0, 1, 1};
            const bool bwd_multi = d->training != 0;
            rc = dep_pack_cluster_split_multi(bwd_multi ? 5 : 3, srcs, dsts, kinds, H, s); if (rc) return rc;
        }
        float* gi = W + lo.gi;
        rc = dep_gemm_internal(0, 1, BTr, G * H, d->F, x, d->F, w0[0], d->F, gi, G * H, w0[2], 0.f, 0, 0, nullptr, 0, s);
        if (rc) return rc;
        dep_fused2_args f{};
        f.B = B; f.T = T; f.training = d->training;
        f.wp0 = R + lo.wp[0][0]; f.wp1 = R + lo.wp[1][0]; f.wpi = W + lo.wih_img;
        f.b_hh0 = w0[3]; f.b_ih1 = w1[2]; f.b_hh1 = w1[3];
        f.ostride = (lo.BT * H + 63) / 64 * 64;
        f.gi = gi; f.y0 = R + lo.y[0]; f.y0d = lo.drop ? R + lo.ydrop[0] : nullptr; f.y1 = R + lo.y[1];
        f.drop_p = lo.drop ? d->dropout_p : 0.f; f.seed = d->seed; f.site = DEP_SITE_RNN0;
        f.pooled = pooled; f.pool_scale = d->pool == DEP_POOL_MEAN ? 1.0f / (float)T : 1.0f;
        f.hn0 = h_n; f.hn1 = h_n ? h_n + (size_t)B * H : nullptr;
        for (int l = 0; l < 2; ++l) for (int k = 0; k < 4; ++k) f.sv[l][k] = d->training ? R + lo.sv[l][k] : nullptr;
        f.stream = s;
        f.soft_fallback = lo.bf16st ? 0 : 1;           // (the tolerant per-layer kernels have no bf16-storage variant: a failed hello raises the status)
        f.sv16 = sv16 ? 1 : 0; f.bf16st = lo.bf16st ? 1 : 0;
        f.hdr_clean = 1;                                   // dep_cluster_reset_status above zeroed every header slot
        rc = dep_launch_fused2_fwd(f, W + lo.xbuf, lo.xbuf_bytes); if (rc) return rc;
        // Fallback, decided ON THE DEVICE (no host synchronisation, identical on every data-parallel rank): the launch above
        // needs every CU to itself; if a foreign workgroup kept its clusters from assembling it has set the workspace's soft
        // flag and left.  The per-layer kernels that tolerate co-scheduled work are enqueued behind it in any case and return
        // at entry unless the flag is set (3 near-empty launches per forward, ~10 us); they write the same reserve layout, so
        // the backward does not care which of the two produced it.
        const unsigned* soft = reinterpret_cast<const unsigned*>(W + lo.xbuf) + 1;
        for (int l = 0; l < 2 && !lo.bf16st; ++l) {
            const float* const* wl = l == 0 ? w0 : w1;
            if (l == 1) {
                const float* in = lo.drop ? R + lo.ydrop[0] : R + lo.y[0];
                dep_gemm_set_predicate(soft);
                rc = dep_gemm_internal(0, 1, BTr, G * H, H, in, H, wl[0], H, gi, G * H, wl[2], 0.f, 0, 0, nullptr, 0, s);
                dep_gemm_set_predicate(nullptr);
                if (rc) return rc;
            }
            dep_sweep_args a{};
            a.B = B; a.T = T; a.H = H; a.cell = d->cell; a.dirs = 1; a.training = d->training; a.impl = d->impl; a.split = 1;
            a.w_hh[0] = wl[1]; a.b_hh[0] = wl[3]; a.wp[0] = R + lo.wp[l][0];
            a.gi = gi; a.y = R + lo.y[l]; a.ldy = H;
            const bool dropl = lo.drop && l == 0;
            a.ydrop = dropl ? R + lo.ydrop[0] : nullptr;
            a.drop_p = dropl ? d->dropout_p : 0.f; a.seed = d->seed; a.site = DEP_SITE_RNN0 + l;
            a.pooled = (l == 1 && pooled) ? pooled : nullptr; a.pool_scale = f.pool_scale;
            a.h_n = h_n ? h_n + (size_t)l * B * H : nullptr;
            if (d->training) { a.sv0 = R + lo.sv[l][0]; a.sv1 = R + lo.sv[l][1]; a.sv2 = R + lo.sv[l][2]; a.sv3 = R + lo.sv[l][3]; }
            a.only_if = soft; a.stream = s; a.sv16 = sv16 ? 1 : 0;
            a.hdr_slot = 1 + l; a.hdr_clean = 1;             // own header slots: still zero from the call's one memset
            rc = dep_launch_cluster_fwd(a, W + lo.xbuf, lo.xbuf_bytes); if (rc) return rc;
        }
        if (y) { rc = dep_axpby(R + lo.y[1], y, (long)lo.BT * H, 1.f, 0.f, s); if (rc) return rc; }
        return DEP_OK;
    }
    if (D == 2) {
        // gather every layer's direction-stacked W_ih and folded bias (b_ih [+ b_hh]) in one launch (16 jobs at a time)
        const float* src[16]; const float* add[16]; float* dst[16]; long cnt[16]; int nj = 0;
        for (int l = 0; l < L; ++l) {
            const int Kl = l == 0 ? d->F : D * H;
            for (int dd = 0; dd < D; ++dd) {
                const float* const* wl = weights + (size_t)(l * D + dd) * 4;
                DEP_CHECK_ARG(wl[0] && wl[1] && wl[2] && wl[3]);
                src[nj] = wl[0]; add[nj] = nullptr; dst[nj] = R + lo.wstack[l] + (size_t)dd * G * H * Kl; cnt[nj++] = (long)G * H * Kl;
                src[nj] = wl[2]; add[nj] = d->cell == DEP_CELL_LSTM ? wl[3] : nullptr; dst[nj] = R + lo.bstack[l] + (size_t)dd * G * H; cnt[nj++] = (long)G * H;
                if (nj == 16) { rc = dep_multi_copy(nj, src, add, dst, cnt, s); if (rc) return rc; nj = 0; }
            }
        }
        if (nj) { rc = dep_multi_copy(nj, src, add, dst, cnt, s); if (rc) return rc; }
    }
    for (int l = 0; l < L; ++l) {
        const float* in = l == 0 ? x : (lo.drop ? R + lo.ydrop[l - 1] : R + lo.y[l - 1]);
        const int Kl = l == 0 ? d->F : D * H;
        float* gi = W + lo.gi;
        const bool stacked = D == 2;                   // both directions' projections as one GEMM over stacked weights
        for (int dd = 0; dd < D; ++dd) {
            const float* const* wl = weights + (size_t)(l * D + dd) * 4;
            DEP_CHECK_ARG(wl[0] && wl[1] && wl[2] && wl[3]);
            // recurrent weight images in MFMA fragment order (precision / clustering decide the format)
            if (split_lstm) {
                rc = dep_pack_cluster_lstm_split(wl[1], R + lo.wp[l][dd], d->training ? R + lo.wpT[l][dd] : nullptr, H, s);
                if (rc) return rc;
            } else {
                const bool split_bwd = lo.cluster && d->cell == DEP_CELL_GRU && sweep_split_mode();
                // the fp32 fragment images are only needed by kernels that are not running on split-precision images
                const bool need_f32 = mfma && !((split_fwd || split_fwd32) && (split_bwd || !d->training));
                if (need_f32) { rc = dep_pack_whh(wl[1], R + lo.wp[l][dd], R + lo.wpT[l][dd], G, H, s); if (rc) return rc; }
                if (split_fwd) { rc = dep_pack_cluster16_fwd_split(wl[1], R + lo.wp[l][dd], H, s); if (rc) return rc; }
                if (split_fwd32) { rc = dep_pack_cluster_fwd_split(wl[1], R + lo.wp[l][dd], H, s); if (rc) return rc; }
                if (lo.cluster && d->training) {     // the cluster backward wants its own member-sliced image
                    rc = split_bwd ? dep_pack_cluster_bwd_split(wl[1], R + lo.wpT[l][dd], H, s)
                                   : dep_pack_cluster_bwd(wl[1], R + lo.wpT[l][dd], G, H, s);
                    if (rc) return rc;
                }
            }
            const float* bias = wl[2];
            if (stacked) continue;                     // weights and biases were stacked above; the GEMM follows the loop
            if (d->cell == DEP_CELL_LSTM) {          // both biases fold into the projection
                float* tb = W + lo.biastmp;
                rc = dep_axpby(wl[2], tb, (long)G * H, 1.f, 0.f, s); if (rc) return rc;
                rc = dep_axpby(wl[3], tb, (long)G * H, 1.f, 1.f, s); if (rc) return rc;
                bias = tb;
            }
            rc = dep_gemm_internal(0, 1, BTr, G * H, Kl, in, Kl, wl[0], Kl, gi + (size_t)dd * G * H, D * G * H, bias,
                                   0.f, 0, 0, nullptr, 0, s);
            if (rc) return rc;
            // the bias scratch is reused by the next direction: stream order keeps this safe
        }
        if (stacked) {
            rc = dep_gemm_internal(0, 1, BTr, D * G * H, Kl, in, Kl, R + lo.wstack[l], Kl, gi, D * G * H, R + lo.bstack[l],
                                   0.f, 0, 0, nullptr, 0, s);
            if (rc) return rc;
        }
        dep_sweep_args a{};
        a.B = B; a.T = T; a.H = H; a.cell = d->cell; a.dirs = D; a.training = d->training; a.impl = d->impl;
        a.split = (split_fwd || split_fwd32 || split_lstm) ? 1 : 0;
        for (int dd = 0; dd < D; ++dd) {
            const float* const* wl = weights + (size_t)(l * D + dd) * 4;
            a.w_hh[dd] = wl[1]; a.b_hh[dd] = wl[3]; a.wp[dd] = R + lo.wp[l][dd];
        }
        a.gi = gi; a.y = R + lo.y[l]; a.ldy = D * H;
        const bool dropl = lo.drop && l < L - 1;
        a.ydrop = dropl ? R + lo.ydrop[l] : nullptr;
        a.drop_p = dropl ? d->dropout_p : 0.f; a.seed = d->seed; a.site = DEP_SITE_RNN0 + l;
        const bool top = l == L - 1;
        a.pooled = (top && pooled) ? pooled : nullptr;
        a.pool_scale = d->pool == DEP_POOL_MEAN ? 1.0f / (float)T : 1.0f;
        a.h_n = h_n ? h_n + (size_t)l * D * B * H : nullptr;
        if (d->training) { a.sv0 = R + lo.sv[l][0]; a.sv1 = R + lo.sv[l][1]; a.sv2 = R + lo.sv[l][2]; a.sv3 = R + lo.sv[l][3]; }
        a.stream = s;
        a.sv16 = (sv16 && lo.cluster && (d->cell == DEP_CELL_LSTM || !use16)) ? 1 : 0;
        a.hdr_slot = l < DEP_HDR_SLOTS ? l : 0; a.hdr_clean = l < DEP_HDR_SLOTS;      // one header slot per layer, zeroed once per call
        rc = use16 ? dep_launch_cluster16_fwd(a, W + lo.xbuf, lo.xbuf_bytes)
           : (lo.cluster && d->cell == DEP_CELL_LSTM) ? dep_launch_cluster_lstm_fwd(a, W + lo.xbuf, lo.xbuf_bytes)
           : lo.cluster ? dep_launch_cluster_fwd(a, W + lo.xbuf, lo.xbuf_bytes) : dep_launch_sweep_fwd(a);
        if (rc) return rc;
    }
    if (y) {
        rc = dep_axpby(R + lo.y[L - 1], y, (long)lo.BT * D * H, 1.f, 0.f, s);
        if (rc) return rc;
    }
    return DEP_OK;
}

// DEP_COMM_OVERLAP=sweep: enqueue a layer's gradient all-reduce as soon as its GEMMs are enqueued (it may then run beside the
// next layer's backward sweep); default: behind that sweep, beside its GEMMs.
static bool comm_beside_sweeps() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DEP_COMM_OVERLAP"); v = (e && e[0] == 's') ? 1 : 0; }
    return v != 0;
}

static int rnn_backward_impl(const dep_rnn_desc* d, const float* x, const float* const* weights, const float* dy,
                             const float* dpooled, const float* dh_n, float* const* dweights, float* dx,
                             void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                             void* stream, const dep_grad_sync* gs) {
    Layout lo;
    DEP_CHECK_ARG(make_layout(d, lo));
    DEP_CHECK_ARG(d->training);
    DEP_CHECK_ARG(x && weights && dweights && reserve && workspace);
    DEP_CHECK_ARG(dy || dpooled || dh_n);
    DEP_CHECK_ARG(!(dpooled && (d->cell != DEP_CELL_GRU || d->pool == DEP_POOL_NONE)));
    if (reserve_bytes < lo.reserve_floats * sizeof(float) || workspace_bytes < lo.ws_floats * sizeof(float)) {
        dep_set_error("dep_rnn_backward: reserve/workspace too small");
        return DEP_ERR_WORKSPACE;
    }
    if (lo.cluster) {
        const int fm = lookup_reserve_mode(reserve);
        if (fm >= 0 && (fm & 1) != (sweep_split_mode() ? 1 : 0)) {
            dep_set_error("dep_rnn_backward: the reserve was produced by a forward in %s mode, the current mode is %s "
                          "(dep_set_gemm_mode must not change between a forward and its backward)",
                          (fm & 1) ? "bf16x3" : "f32", (fm & 1) ? "f32" : "bf16x3");
            return DEP_ERR_ARG;
        }
        if (fm >= 0 && ((fm >> 3) & 1) != (lo.bf16st ? 1 : 0)) {
            dep_set_error("dep_rnn_backward: the reserve was %swritten in bf16-storage mode (dep_set_gemm_mode(3)), this call runs in the other", (fm & 8) ? "" : "not ");
            return DEP_ERR_ARG;
        }
        if (fm >= 0 && ((fm >> 2) & 1) != ((lo.sv16 && sweep_split_mode()) ? 1 : 0)) {
            dep_set_error("dep_rnn_backward: the reserve holds %s saved gates, this call expects the other format (DEP_SV16 / DEP_EXCLUSIVE changed?)",
                          (fm & 4) ? "16-bit" : "fp32");
            return DEP_ERR_ARG;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    float* R = (float*)reserve; float* W = (float*)workspace;
    const int B = d->B, T = d->T, H = d->H, D = d->dirs, G = lo.G, L = d->L;
    const int BTr = (int)lo.BT;
    void* gws = W + lo.gemm; const size_t gwsb = lo.gemm_bytes;
    // (dX's weight image shares the region with the split-K partials of the contractions enqueued behind it: stream order keeps them apart)
    struct ScratchGuard { ScratchGuard(void* q, size_t n) { dep_gemm_set_scratch(q, n); } ~ScratchGuard() { dep_gemm_set_scratch(nullptr, 0); } } scratch_guard(gws, gwsb);
    int rc;
    // The fused two-layer backward (rnn_fused2_bwd.hip; round 5: all-gather form): both layers' BPTT in ONE launch, layer 1's dX -- the gradient
    // entering layer 0 -- formed in-kernel.  It writes the same gate-gradient arrays as the per-layer sweeps (4H-wide rows, PK image when the
    // contractions take it), so everything behind the sweeps -- bias finish, dW GEMMs (paired), layer 0's dX, the gradient ranges -- is the
    // per-layer loop below with the sweep launches and layer 1's dX GEMM left out.  DEP_FUSED2_BWD=1 / 0.
    static int fused_bwd_on = -1;
    if (fused_bwd_on < 0) { const char* e = getenv("DEP_FUSED2_BWD"); fused_bwd_on = e ? ((e[0] == '1') ? 1 : 0) : DEP_FUSED2_BWD_DEFAULT; }
    const bool fused = lo.fused2 && fused_bwd_on && sweep_split_mode() && d->cell == DEP_CELL_GRU && L == 2 && D == 1 && dep_fused2_bwd_fits(B, T);
    static int pk_env = -1;
    if (pk_env < 0) { const char* e = getenv("DEP_DGI_PK"); pk_env = (e && e[0] == '0') ? 0 : 1; }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    // may layer l's gate gradients be the PK image?  Only when all contractions that read them really run the three-term kernel on its vector path.
    auto pk_gru_ok = [&](int l, bool has_dxl, const float* dxl_probe) {
        const float* in = l == 0 ? x : (lo.drop ? R + lo.ydrop[l - 1] : R + lo.y[l - 1]);
        const int Kl = l == 0 ? d->F : D * H;
        return pk_env && lo.dg4 && lo.cluster && d->cell == DEP_CELL_GRU && sweep_split_mode() && dep_get_gemm_mode() >= 1 &&
               (fused || dep_cluster_bwd_pk_ok(H, T)) && (T % 2 == 0) && (BTr % 2 == 0) && (Kl % 4 == 0) && al16(in) && al16(weights[(size_t)l * 4]) &&
               al16(dweights[(size_t)l * 4]) && al16(dweights[(size_t)l * 4 + 1]) &&
               dep_gemm_uses_bf16x3(G * H, Kl, BTr, 0) && dep_gemm_uses_bf16x3(3 * H, H, BTr, T) &&
               (!has_dxl || (dep_gemm_uses_bf16x3(BTr, Kl, G * H, 0) && al16(dxl_probe)));
    };
    bool fused_pk = false;
    if (fused) {
        const float* const* w0 = weights; const float* const* w1 = weights + 4;
        float* const* g0 = dweights; float* const* g1 = dweights + 4;
        for (int k = 0; k < 4; ++k) DEP_CHECK_ARG(w0[k] && w1[k] && g0[k] && g1[k]);
        const bool sv16 = lo.sv16 && sweep_split_mode();
        fused_pk = sv16 && pk_gru_ok(1, false, nullptr) && pk_gru_ok(0, dx != nullptr, dx);      // one kernel writes both layers: both or neither
        rc = dep_pack_cluster_bwd_split(w1[0], W + lo.wih_img, H, s); if (rc) return rc;
        dep_fused2_bwd_args f{};
        f.B = B; f.T = T;
        f.wh1 = R + lo.wpT[1][0]; f.wi1 = W + lo.wih_img; f.wh0 = R + lo.wpT[0][0];
        f.y1 = R + lo.y[1]; f.y0 = R + lo.y[0]; f.sv1 = R + lo.sv[1][0]; f.sv0 = R + lo.sv[0][0];
        f.svstride = (lo.BT * H + 63) / 64 * 64;
        f.dy = dy; f.dpooled = dpooled; f.pool_scale = d->pool == DEP_POOL_MEAN ? 1.0f / (float)T : 1.0f;
        f.dhn1 = dh_n ? dh_n + (size_t)B * H : nullptr; f.dhn0 = dh_n;
        f.drop_p = lo.drop ? d->dropout_p : 0.f; f.seed = d->seed; f.site = DEP_SITE_RNN0;
        f.dgi1 = W + lo.gi; f.dgi0 = W + lo.gi2;
        f.dghn1 = lo.dg4 ? f.dgi1 + 3 * H : W + lo.dghn; f.dghn0 = lo.dg4 ? f.dgi0 + 3 * H : W + lo.dghn2;
        f.lddg = lo.dg4 ? 4 * H : 3 * H; f.lddghn = lo.dg4 ? 4 * H : H;
        f.dbpart1 = W + lo.dbpart; f.dbpart0 = W + lo.dbpart2; f.dbpart_rows = lo.nwg; f.stream = s;
        if (lo.bf16st && !fused_pk) {
            dep_set_error("dep_rnn_backward: bf16-storage mode needs the pre-split gate-gradient path (aligned operands, DEP_DGI_PK not 0, contractions above the split threshold)");
            return DEP_ERR_ARG;
        }
        f.sv16 = sv16 ? 1 : 0; f.dg_pk = fused_pk ? 1 : 0; f.bf16st = lo.bf16st ? 1 : 0;
        rc = dep_launch_fused2_bwd(f, W + lo.xbuf, lo.xbuf_bytes); if (rc) return rc;
    }
    float* pending_ptr = nullptr; long pending_n = 0;      // data parallel: a finished layer's gradient range waiting for the next sweep to be enqueued
    if (lo.cluster && !fused) { rc = dep_cluster_reset_flags(W + lo.xbuf, s); if (rc) return rc; }      // every layer's header slot in one memset (the status words stay)
    for (int l = L - 1; l >= 0; --l) {
        const bool top = l == L - 1;
        const float* in = l == 0 ? x : (lo.drop ? R + lo.ydrop[l - 1] : R + lo.y[l - 1]);
        const int Kl = l == 0 ? d->F : D * H;
        float* dgi = (fused && l == 0) ? W + lo.gi2 : W + lo.gi;       // (the fused launch left both layers' gate gradients behind)
        dep_sweep_bwd_args a{};
        a.B = B; a.T = T; a.H = H; a.cell = d->cell; a.dirs = D; a.impl = d->impl;
        // must match the image dep_rnn_forward packed: the precision mode may not change between a forward and its backward
        a.split = (lo.cluster && sweep_split_mode()) ? 1 : 0;
        for (int dd = 0; dd < D; ++dd) {
            const float* const* wl = weights + (size_t)(l * D + dd) * 4;
            a.w_hh[dd] = wl[1]; a.wpT[dd] = R + lo.wpT[l][dd];
        }
        a.y = R + lo.y[l]; a.ldy = D * H;
        if (top) { a.dy = dy; a.drop_p = 0.f; }
        else { a.dy = W + lo.dx[(l + 1) & 1]; a.drop_p = lo.drop ? d->dropout_p : 0.f; }
        a.lddy = D * H; a.seed = d->seed; a.site = DEP_SITE_RNN0 + l;
        a.dpooled = top ? dpooled : nullptr;
        a.pool_scale = d->pool == DEP_POOL_MEAN ? 1.0f / (float)T : 1.0f;
        a.dh_n = dh_n ? dh_n + (size_t)l * D * B * H : nullptr;
        a.sv0 = R + lo.sv[l][0]; a.sv1 = R + lo.sv[l][1]; a.sv2 = R + lo.sv[l][2]; a.sv3 = R + lo.sv[l][3];
        const int ldg = lo.dg4 ? 4 * H : D * G * H;                  // row stride of the gate-gradient array
        float* dghn = lo.dg4 ? dgi + 3 * H : ((fused && l == 0) ? W + lo.dghn2 : W + lo.dghn);
        a.dgi = dgi; a.dghn = dghn; a.lddg = lo.dg4 ? ldg : 0; a.lddghn = lo.dg4 ? ldg : 0;
        a.dbpart = (fused && l == 0) ? W + lo.dbpart2 : W + lo.dbpart; a.dbpart_rows = D * lo.nwg; a.stream = s;
        a.hdr_slot = l < DEP_HDR_SLOTS ? l : 0; a.hdr_clean = l < DEP_HDR_SLOTS;
        // Round 4: the sweep writes the gate gradients as the PK image (rows (t even, t + 1) = (hi, lo) bf16 pairs of both steps,
        // gemm_bf16x3.hip) -- the three contractions that read them (dX, dW_ih, dW_hh) then stage them without converting; same
        // bytes, same bits.  Only when all three really run the three-term kernel on its vector path.
        a.split = (lo.cluster && sweep_split_mode()) ? 1 : 0;
        float* dxl_probe = l == 0 ? dx : (fused ? nullptr : W + lo.dx[l & 1]);
        const bool pk_gru = fused ? fused_pk : (a.split && pk_gru_ok(l, dxl_probe != nullptr, dxl_probe));
        // the BiLSTM cluster sweep (both directions in one launch, direction-stacked contractions): same image, same conditions
        bool pk_lstm = pk_env && lo.cluster && d->cell == DEP_CELL_LSTM && D == 2 && lo.wstack[l] != 0 && a.split && dep_get_gemm_mode() >= 1 &&
                       dep_cluster_lstm_bwd_pk_ok(T) && (BTr % 2 == 0) && (Kl % 4 == 0) && al16(in) && al16(W + lo.dwstack) &&
                       dep_gemm_uses_bf16x3(D * G * H, Kl, BTr, 0) && dep_gemm_uses_bf16x3(4 * H, H, BTr, T) &&
                       (!dxl_probe || (dep_gemm_uses_bf16x3(BTr, Kl, D * G * H, 0) && al16(dxl_probe)));
        for (int dd = 0; dd < D && pk_lstm; ++dd) pk_lstm = al16(dweights[(size_t)(l * D + dd) * 4 + 1]);
        const bool pk = pk_gru || pk_lstm;
        a.dg_pk = pk ? 1 : 0;
        if (lo.bf16st && !pk) {
            dep_set_error("dep_rnn_backward: bf16-storage mode needs the pre-split gate-gradient path (aligned operands, DEP_DGI_PK not 0, contractions above the split threshold)");
            return DEP_ERR_ARG;
        }
        a.bf16st = lo.bf16st ? 1 : 0;
        // FMT_PKH / FMT_PK (gemm_bf16x3.hip).  Round 6: the single-product modes (dep_set_gemm_mode(2 / 3)) read the sweep's PK image through its hi rows
        // on every stack (the same bf16 values their on-the-fly conversion formed: bit-identical, without the fp32 staging path that made
        // cfg3's weight gradients slower in that mode than with three products)
        const int fmt_a = (lo.bf16st || dep_get_gemm_mode() >= 2) ? 2 : 1;
        a.sv16 = (lo.sv16 && sweep_split_mode() && lo.cluster) ? 1 : 0;
        struct FmtGuard { bool on; ~FmtGuard() { if (on) dep_gemm_set_operand_formats(0, 0); } } fmt_guard{pk};
        if (!fused) {
            rc = (lo.cluster && d->cell == DEP_CELL_LSTM) ? dep_launch_cluster_lstm_bwd(a, W + lo.xbuf, lo.xbuf_bytes)
               : lo.cluster ? dep_launch_cluster_bwd(a, W + lo.xbuf, lo.xbuf_bytes) : dep_launch_sweep_bwd(a);
            if (rc) return rc;
        }
        if (pending_ptr) {
            // the layer above's gradient range: the event recorded here completes with this sweep, the all-reduce then runs
            // beside this layer's GEMMs.  A cluster sweep needs every one of its workgroups resident (one per CU, most of a CU's
            // registers and LDS): a collective kernel that holds CUs when the sweep is dispatched delays the members that
            // cannot be placed -- and with them the whole launch -- until the collective's peers on the other GPUs let it finish.
            rc = dep_comm_enqueue_after((dep_comm*)gs->comm, pending_ptr, pending_n, s, (hipStream_t)gs->comm_stream);
            if (rc) return rc;
            pending_ptr = nullptr;
        }
        float* dbi[2]; float* dbh[2];
        for (int dd = 0; dd < D; ++dd) {
            float* const* gl = dweights + (size_t)(l * D + dd) * 4;
            DEP_CHECK_ARG(gl[0] && gl[1] && gl[2] && gl[3]);
            dbi[dd] = gl[2]; dbh[dd] = gl[3];
        }
        rc = dep_finish_db(a, dbi, dbh);
        if (rc) return rc;
        if (pk) dep_gemm_set_operand_formats(fmt_a, 0);           // A = the PK gate gradients in dX, dW_ih and dW_hh below (reset by fmt_guard)
        float* dxl = l == 0 ? dx : (fused ? nullptr : W + lo.dx[l & 1]);      // (fused: the gradient entering layer 0 never left the chip)
        // dX (B*T, Kl) (+)= dG * W_ih first: it is the only product the next layer's sweep waits for
        const bool stacked = D == 2 && lo.wstack[l] != 0;
        if (dxl && stacked) {
            // dX = [dG_fwd | dG_bwd] [W_ih(fwd); W_ih(bwd)]: one contraction over K = 2 G H (the forward left the stacked copy in
            // the reserve) instead of two with a read-modify-write of dX in between
            rc = dep_gemm_internal(0, 0, BTr, Kl, D * G * H, dgi, ldg, R + lo.wstack[l], Kl, dxl, Kl, nullptr, 0.f, 0, 0, nullptr, 0, s);
            if (rc) return rc;
        } else if (dxl) {
            for (int dd = 0; dd < D; ++dd) {
                const float* const* wl = weights + (size_t)(l * D + dd) * 4;
                rc = dep_gemm_internal(0, 0, BTr, Kl, G * H, dgi + (size_t)dd * G * H, ldg, wl[0], Kl, dxl, Kl, nullptr,
                                       dd == 0 ? 0.f : 1.f, 0, 0, nullptr, 0, s);
                if (rc) return rc;
            }
        }
        if (stacked) {
            // dW_ih of both directions: (2 G H x Kl) = dG^T in, the input read once; rows [0, G H) / [G H, 2 G H) are the two tensors
            float* dws = W + lo.dwstack;
            rc = dep_gemm_internal(1, 0, D * G * H, Kl, BTr, dgi, ldg, in, Kl, dws, Kl, nullptr, 0.f, 0, 0, gws, gwsb, s);
            if (rc) return rc;
            const float* src[2]; float* dst[2]; long cnt[2];
            for (int dd = 0; dd < D; ++dd) {
                src[dd] = dws + (size_t)dd * G * H * Kl; dst[dd] = (dweights + (size_t)(l * D + dd) * 4)[0]; cnt[dd] = (long)G * H * Kl;
            }
            rc = dep_multi_copy(D, src, nullptr, dst, cnt, s); if (rc) return rc;
        }
        bool paired = false;
        // (the split-K target follows the layer's SHAPE, not whether the pair really runs: the fp32-row path (DEP_DGI_PK=0) and the unpaired path
        // (DEP_DW_PAIR=0) must keep summing in the same order as the pair -- tests/test_presplit_gpu.py holds them bit-identical)
        const bool pair_layer = d->cell == DEP_CELL_GRU && lo.dg4 && !stacked && D == 1 && Kl == H && dep_get_gemm_mode() == 1 && sweep_split_mode();      // (the other precision modes never pair: they keep the single launches' target)
        const bool pair_shape = pair_layer && pk_gru && !lo.bf16st;
        struct SplitGuard { bool on; SplitGuard(bool o) : on(o) { if (on) dep_gemm_set_split_target(512); } ~SplitGuard() { if (on) dep_gemm_set_split_target(0); } } split_guard(pair_layer);
        if (pair_shape) {
            // Round 5: dW_ih and dW_hh of this layer in ONE launch -- both read the PK gate gradients, [dr | dz] are the same bytes
            // (gemm_bf16x3_tn_pair; bit-identical to the two calls below, which remain the path for every other configuration)
            float* const* gl = dweights + (size_t)l * 4;
            const int pr = dep_gemm_tn_pair(G * H, H, BTr, dgi, dgi, ldg, 2 * H, H, in, Kl, 0, 0, R + lo.y[l], H, T, -1, gl[0], Kl, gl[1], H, gws, gwsb, s);
            if (pr < 0) return pr;
            paired = pr == 1;
        }
        bool paired_hh = false;
        if (pk_lstm && stacked && D == 2) {
            // Round 5: dW_hh of the two directions of a BiLSTM layer (4H x H each: 254 tile-jobs, half of the persistent grid) as ONE paired launch;
            // tiles, K chunks and split-K order per direction are those of the single launches (bit-identical)
            float* const* gf = dweights + (size_t)(l * D) * 4; float* const* gb = dweights + (size_t)(l * D + 1) * 4;
            const int pr = dep_gemm_tn_pair(G * H, H, BTr, dgi, dgi + (size_t)G * H, ldg, 0, 0, R + lo.y[l], D * H, T, -1, R + lo.y[l] + H, D * H, T, 1,
                                            gf[1], H, gb[1], H, gws, gwsb, s);
            if (pr < 0) return pr;
            paired_hh = pr == 1;
        }
        for (int dd = 0; dd < D && !paired && !paired_hh; ++dd) {
            float* const* gl = dweights + (size_t)(l * D + dd) * 4;
            const float* dg = dgi + (size_t)dd * G * H;
            // dW_ih (G*H, Kl) = dG^T * in
            if (!stacked) {
                if (lo.bf16st && l > 0) dep_gemm_set_operand_formats(fmt_a, 3);          // the layer below's (dropped) output is a bf16 array
                rc = dep_gemm_internal(1, 0, G * H, Kl, BTr, dg, ldg, in, Kl, gl[0], Kl, nullptr, 0.f, 0, 0, gws, gwsb, s);
                if (rc) return rc;
            }
            if (lo.bf16st) dep_gemm_set_operand_formats(fmt_a, 3);                       // dW_hh: B = this layer's bf16 output, shifted one step
            // dW_hh (G*H, H) = dGH^T * h_prev   (h_prev = layer output shifted by one step along the sweep)
            const float* yl = R + lo.y[l] + (size_t)dd * H;
            const int shift = dd == 0 ? -1 : 1;
            if (d->cell == DEP_CELL_GRU && lo.dg4 && dep_gemm_uses_bf16x3(3 * H, H, BTr, T)) {
                // ONE contraction over the 4H-wide rows: op(A) columns [dr | dz] and [dn*r] (the dn block in between is skipped by
                // the loader).  The separate (H x H) call for the n rows cost 100 us for a third of the (2H x H) call's 165 us work.
                dep_gemm_set_a_colskip(2 * H, H);
                rc = dep_gemm_internal(1, 0, 3 * H, H, BTr, dg, ldg, yl, D * H, gl[1], H, nullptr, 0.f, T, shift, gws, gwsb, s);
                dep_gemm_set_a_colskip(0, 0);
                if (rc) return rc;
            } else if (d->cell == DEP_CELL_GRU) {
                rc = dep_gemm_internal(1, 0, 2 * H, H, BTr, dg, ldg, yl, D * H, gl[1], H, nullptr, 0.f, T, shift, gws, gwsb, s);
                if (rc) return rc;
                rc = dep_gemm_internal(1, 0, H, H, BTr, dghn, lo.dg4 ? ldg : H, yl, D * H, gl[1] + (size_t)2 * H * H, H, nullptr,
                                       0.f, T, shift, gws, gwsb, s);
                if (rc) return rc;
            } else {
                rc = dep_gemm_internal(1, 0, 4 * H, H, BTr, dg, ldg, yl, D * H, gl[1], H, nullptr, 0.f, T, shift, gws, gwsb, s);
                if (rc) return rc;
            }
        }
        // data parallel: layer l's gradients are complete -- their range of the caller's flat gradient buffer goes to RCCL on the
        // communication stream.  Not right away: the collective is enqueued behind the NEXT layer's sweep (see `pending' above),
        // so that it travels over xGMI beside that layer's weight-gradient GEMMs and never beside a sweep.  The bottom layer's
        // range has nothing left to hide behind and goes out at once.
        if (gs && gs->comm && gs->range_ptr[l] && gs->range_count[l] > 0) {
            if (l > 0 && !comm_beside_sweeps() && !fused) { pending_ptr = gs->range_ptr[l]; pending_n = gs->range_count[l]; }
            else {
                rc = dep_comm_enqueue_after((dep_comm*)gs->comm, gs->range_ptr[l], gs->range_count[l], s, (hipStream_t)gs->comm_stream);
                if (rc) return rc;
            }
        }
    }
    return DEP_OK;
}

extern "C" int dep_rnn_backward(const dep_rnn_desc* d, const float* x, const float* const* weights, const float* dy,
                                const float* dpooled, const float* dh_n, float* const* dweights, float* dx,
                                void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                                void* stream) {
    return rnn_backward_impl(d, x, weights, dy, dpooled, dh_n, dweights, dx, reserve, reserve_bytes, workspace, workspace_bytes,
                             stream, nullptr);
}

extern "C" int dep_rnn_backward_overlapped(const dep_rnn_desc* d, const float* x, const float* const* weights, const float* dy,
                                           const float* dpooled, const float* dh_n, float* const* dweights, float* dx,
                                           void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                                           void* stream, const dep_grad_sync* gs) {
    DEP_CHECK_ARG(gs && gs->comm && gs->comm_stream && gs->comm_stream != stream);
    return rnn_backward_impl(d, x, weights, dy, dpooled, dh_n, dweights, dx, reserve, reserve_bytes, workspace, workspace_bytes,
                             stream, gs);
}
