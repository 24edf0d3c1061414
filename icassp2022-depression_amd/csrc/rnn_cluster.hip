// Cluster-parallel recurrent sweeps: the serial critical path of nn.GRU / nn.LSTM spread over the chip.
//
// Problem with one workgroup per 16-utterance tile (rnn_sweep.hip): B = 512 gives 32 workgroups, each
// streaming the whole W_hh from L2 every step and doing 6.3 MFLOP of fp32 MFMA on ONE CU -> ~20 us/step.
// In the cluster kernels a tile is owned by a CLUSTER of workgroups (one or two per CU, all co-resident).  Member c keeps
// its rows of W_hh in VGPRs for the whole sweep (no weight traffic at all), computes its slice of the gates on the matrix
// cores and exchanges
//   forward : its 16 x units slice of h_t        (all-gather)
//   backward: its 16 x H partial of dh_{t-1}     (reduce-scatter, summed in fixed member order)
// with the other members through global memory once per step (protocol: rnn_cluster_common.h).  The kernels live in
// rnn_cluster16.hip (GRU forward, 16-unit members), rnn_cluster_bwd.hip (GRU, 32-unit members) and rnn_cluster_lstm.hip;
// this file holds what they share on the host side.  A launch must keep every member of every cluster resident, which
// bounds it to 256 (one per CU) or 512 (two per CU) workgroups; larger batches run as consecutive chunks of the batch.
// (The first version of the exchange, 8-byte {epoch, value} granules polled by the consumers, was 2-3x slower than the
// flag-published payload for the 4096-value reduce-scatter and is no longer kept.)
#include "rnn_cluster_common.h"

using depc::BT; using depc::FLAG_OFF; using depc::PAYLOAD_OFF;

namespace {

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
// Every member of every cluster of a launch must be resident at the same time (they wait for each other inside the launch),
// so the number of tiles per launch follows the number of CUs actually present (a partitioned or harvested device has fewer
// than 256); larger batches run as consecutive chunks.
static int num_cus() {
    static int n = -1;
    if (n < 0) {
        const char* e = getenv("DEP_NUM_CUS");
        int dev = 0, v = 0;
        if (e && atoi(e) > 0) n = atoi(e);
        else if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

int dep_cluster_chunk(int members, int per_cu, int max_wgs) {
    long wgs = (long)num_cus() * per_cu;
    if (wgs > max_wgs) wgs = max_wgs;
    long tiles = wgs / members;
    if (tiles >= 8) tiles = tiles / 8 * 8;             // whole groups of 8 tiles: block id -> XCD stays member-invariant
    if (tiles < 1) tiles = 1;
    return (int)tiles * BT;
}

bool dep_cluster_ok(int cell, int H, int B, int dirs) {
    (void)B; (void)dirs;                              // any batch: the launchers chunk it
    return cell == DEP_CELL_GRU && (H == 64 || H == 128 || H == 256 || H == 512);     // KCH = H/32 in {2,4,8,16}; NTW = H/64 in {1,2,4,8}
}

// header + the largest (backward) exchange of one launch chunk
size_t dep_cluster_xbuf_bytes(int cell, int H, int B, int dirs) {
    if (!dep_cluster_ok(cell, H, B, dirs)) return 0;
    const int NC = H / 32, CH = dep_cluster_chunk(NC, 1, 256);
    const int nbtp = (dep_cdiv(B < CH ? B : CH, BT) + 7) / 8 * 8;
    return PAYLOAD_OFF + 8192 + (size_t)2 * nbtp * NC * BT * H * sizeof(float) * 2;
}

// The status word (first 256 bytes of the exchange buffer header, see rnn_cluster_common.h) is raised by any sweep whose
// bounded spin gave up and is STICKY: later sweeps of the same step see it at kernel entry and leave at once, so a failure
// in the layer-0 forward is still there when the host reads dep_rnn_status after the whole step.
int dep_cluster_reset_status(void* xbuf, hipStream_t s) {
    if (hipMemsetAsync(xbuf, 0, PAYLOAD_OFF, s) != hipSuccess) { dep_set_error("hipMemsetAsync failed"); return DEP_ERR_HIP; }
    return DEP_OK;
}
int dep_cluster_reset_flags(void* xbuf, hipStream_t s) {
    if (hipMemsetAsync((char*)xbuf + FLAG_OFF, 0, PAYLOAD_OFF - FLAG_OFF, s) != hipSuccess) { dep_set_error("hipMemsetAsync failed"); return DEP_ERR_HIP; }
    return DEP_OK;
}

int dep_pack_cluster_bwd(const float* w_hh, float* out, int G, int H, hipStream_t s) {
    const long n = (long)G * H * H;
    DEP_LAUNCH(pack_cluster_bwd_kernel, dim3(dep_cdiv(n, 256)), dim3(256), 0, s, w_hh, out, G, H);
    DEP_CHECK_LAUNCH();
    return DEP_OK;
}
