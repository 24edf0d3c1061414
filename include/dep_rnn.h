/*
 * dep_rnn.h -- C-ABI of libdep_rnn.so: the MI355X (gfx950) operator library that replaces the
 * PyTorch operator layer (torch.nn.GRU / LSTM / LayerNorm / Linear / Softmax / losses / Adam[W] +
 * autograd) under the reference's five training scripts.
 *
 * The reference (speechandlanguageprocessing/ICASSP2022-Depression) has no FFI layer of its own:
 * its boundary is the torch.nn operator contract used at the call sites cited on every entry
 * point below (paths relative to DepressionCollected/).  A maintainer binds these entry points
 * with ctypes; nothing torch-specific crosses the boundary.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (or int32 where stated), row-major, caller-owned;
 *   - sequences are batch-first: row (b,t) of a (B,T,X) tensor sits at ((b*T + t) * ld);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - return 0 on success, negative dep_status on error; dep_last_error() returns a message;
 *   - no allocation happens inside the library: workspace / reserve sizes come from the *_bytes
 *     queries and the caller provides the buffers (16-byte aligned);
 *   - gate order is PyTorch's: GRU r,z,n ; LSTM i,f,g,o ; h0 = c0 = 0 always.
 *   - threads: entry points may be called concurrently from several host threads as long as each call has its own stream,
 *     workspace and reserve (dep_last_error is per thread; the launch-time recorder behind dep_profile_* and the
 *     reserve-mode record are mutex-guarded).  PROCESS-GLOBAL, not per stream: dep_set_gemm_mode (precision mode of every
 *     later call on any thread), dep_profile_enable, and the DEP_* environment switches, which are read once per process.
 */
#ifndef DEP_RNN_H
#define DEP_RNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DEP_OK = 0,
    DEP_ERR_ARG = -1,      /* bad argument (null pointer, size <= 0, unsupported combination) */
    DEP_ERR_WORKSPACE = -2, /* workspace / reserve too small */
    DEP_ERR_HIP = -3       /* a HIP call failed; see dep_last_error() */
} dep_status;

const char* dep_last_error(void);
/* library/ABI version and compiled arch string ("gfx950") */
int dep_version(void);
const char* dep_arch(void);

/* ------------------------------------------------------------------ descriptors ---- */
enum { DEP_POOL_NONE = 0, DEP_POOL_MEAN = 1, DEP_POOL_SUM = 2 };
enum { DEP_CELL_GRU = 0, DEP_CELL_LSTM = 1 };

/* One stacked recurrent network: torch.nn.GRU(F,H,num_layers=L,dropout=p,batch_first=True)
 * (Classification/audio_gru_whole.py:59-60) or torch.nn.LSTM(F,H,num_layers=L,dropout=p,
 * bidirectional=True) (Classification/text_bilstm_whole.py:54-56). */
typedef struct {
    int32_t cell;        /* DEP_CELL_GRU | DEP_CELL_LSTM */
    int32_t B, T, F, H;  /* batch, steps, input features, hidden units */
    int32_t L;           /* stacked layers (>=1) */
    int32_t dirs;        /* 1 (GRU, unidirectional) or 2 (bidirectional LSTM) */
    int32_t training;    /* 1: keep the reserve for backward and apply inter-layer dropout */
    float   dropout_p;   /* inter-layer dropout probability (applied to layers 0..L-2 outputs) */
    uint64_t seed;       /* Philox key for this call's dropout masks */
    int32_t pool;        /* GRU only: DEP_POOL_* over T of the top layer (fused in the sweep) */
    int32_t impl;        /* 0 auto (cluster > tile-MFMA > generic), 1 generic kernels, 2 one-workgroup-per-tile
                            MFMA kernels, 3 cluster-parallel MFMA kernels (GRU with H in {128,256}, BiLSTM with
                            H = 128; any B -- batches beyond one co-resident launch run as consecutive chunks) */
} dep_rnn_desc;

size_t dep_rnn_reserve_bytes(const dep_rnn_desc* d);     /* activations kept fwd -> bwd */
size_t dep_rnn_workspace_bytes(const dep_rnn_desc* d);   /* scratch, either direction */
/* Byte offset, inside the reserve, of layer `layer`'s output sequence (B,T,H*dirs) -- zero-copy
 * access to `output` for the caller (attention reads it in place); (size_t)-1 on bad arguments. */
size_t dep_rnn_reserve_y_offset(const dep_rnn_desc* d, int layer);
/* Same for the dropped-out copy that feeds layer+1 (training && dropout_p > 0 only). */
size_t dep_rnn_reserve_ydrop_offset(const dep_rnn_desc* d, int layer);

/* Health of the cluster-parallel sweeps that ran on `workspace` since the last dep_rnn_forward (desc.impl 0/3 with a
 * supported H): they exchange data between workgroups inside one launch with bounded spins; if a spin ever gives up
 * the kernels exit early and this returns DEP_ERR_HIP.  The status is STICKY over a step: dep_rnn_forward clears it once,
 * every later sweep (the other layers, the whole backward) leaves it alone and exits at entry when it is raised, so one
 * query after forward + backward sees a failure of any of the step's sweeps.  Synchronises `stream`.  Always DEP_OK for
 * the single-workgroup kernels. */
int dep_rnn_status(const dep_rnn_desc* d, void* workspace, void* stream);
/* Debug tooling: byte offset of the cluster exchange buffer inside the workspace ((size_t)-1 if unused).
 * With DEP_TRACE=1 workgroup 0 of the GRU sweeps leaves shader-clock stamps of its phases at +6400
 * (tools/trace_fwd.py, tools/trace_bwd.py). */
size_t dep_rnn_workspace_xbuf_offset(const dep_rnn_desc* d);
/* Kernels that need every CU to themselves.  The default GRU forward at H = 256 (both layers fused into one launch of twelve
 * 168-register waves per CU) cannot share the GPU: a foreign workgroup in the dispatcher keeps its clusters from assembling.
 * It then gives up within a few ms WITHOUT an error -- it sets the "soft" word (the uint32 after the status word, i.e. at
 * dep_rnn_workspace_xbuf_offset() + 4) and the per-layer kernels enqueued behind it, which tolerate co-scheduled work and are
 * no-ops otherwise, redo the forward: same reserve layout, results within the same 1e-4 of the reference, no host
 * synchronisation, every data-parallel rank decides for itself on the device.  A host that finds the soft word set at a
 * synchronisation point it has anyway should call dep_rnn_set_exclusive(0): later forwards then skip the attempt (and its
 * time-out) for the rest of the process.  PROCESS-GLOBAL; default 1, or 0 with DEP_EXCLUSIVE=0 in the environment. */
int dep_rnn_set_exclusive(int on);
int dep_rnn_get_exclusive(void);

/* ------------------------------------------------------------------ RNN stacks ----- */
/* weights: array of 4*L*dirs device pointers ordered, for layer l and direction d (index
 * (l*dirs+d)*4 + {0,1,2,3}): weight_ih (G*H, in), weight_hh (G*H, H), bias_ih (G*H), bias_hh (G*H),
 * in = F for l = 0 else H*dirs; G = 3 (GRU) or 4 (LSTM)  -- the tensors of
 * state_dict()['lstm_net_audio.weight_ih_l0'] ... / ['lstm_net.weight_ih_l0_reverse'] ...
 *
 * Precision: the recurrent products of the cluster sweeps follow the GEMM mode (dep_set_gemm_mode below): 3-term bf16
 * split by default, exact fp32 MFMA in mode 0.  The packed recurrent weights in the reserve are mode-specific, so the
 * mode must not change between a dep_rnn_forward and the dep_rnn_backward that consumes its reserve: the library records
 * the mode each reserve was produced in and dep_rnn_backward returns DEP_ERR_ARG on a mismatch.
 *
 * dep_rnn_forward replaces `x, _ = self.lstm_net_audio(x)` (audio_gru_whole.py:105) and
 * `output, (h_n, _) = self.lstm_net(x)` (text_bilstm_whole.py:105).
 *   x      (B,T,F)
 *   y      (B,T,H*dirs) top-layer output, may be NULL when only `pooled` is wanted
 *   pooled (B,H) mean/sum over T of the top layer (GRU, desc.pool != NONE), else NULL
 *   h_n    (L*dirs, B, H) final hidden states ordered [l0_fwd, l0_bwd, l1_fwd, ...], may be NULL
 */
int dep_rnn_forward(const dep_rnn_desc* d, const float* x, const float* const* weights,
                    float* y, float* pooled, float* h_n,
                    void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Backward of dep_rnn_forward (what loss.backward() does through nn.GRU/nn.LSTM,
 * audio_gru_whole.py:190).  Gradients are WRITTEN (not accumulated) to dweights (same order and
 * shapes as weights).
 *   dy      (B,T,H*dirs) grad of y, or NULL
 *   dpooled (B,H) grad of pooled, or NULL  (the 1/T of a mean pool is applied inside)
 *   dh_n    (L*dirs,B,H) grad of h_n, or NULL
 *   dx      (B,T,F) grad of x, or NULL to skip it (the reference computes it but never uses it)
 */
int dep_rnn_backward(const dep_rnn_desc* d, const float* x, const float* const* weights,
                     const float* dy, const float* dpooled, const float* dh_n,
                     float* const* dweights, float* dx,
                     void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ------------------------------------------------------------------ data parallel -- */
/* RCCL over xGMI, one process per GPU (north_star; SURVEY 8b `dep_comm_{init,allreduce,destroy}`, 8e).  The reference has no
 * distributed code: these entry points are what a data-parallel `train()` binds -- the utterances of every global
 * mini-batch are split over the ranks, every rank normalises its loss gradient by the GLOBAL batch size, and the flat
 * fp32 gradient buffer is SUM all-reduced (the reference's `loss.backward(); optimizer.step()`,
 * Classification/audio_gru_whole.py:190-191, then sees the batch-mean gradient on every rank).
 * librccl is loaded at the first call (dlopen): single-GPU processes never touch it.
 *   dep_comm_available : 1 if librccl resolves in this process, else 0 (dep_last_error says why); not a collective -- the
 *                        host agrees on it across ranks before anyone enters dep_comm_init;
 *   dep_comm_unique_id : rank 0 obtains the 128-byte rendezvous id (ncclGetUniqueId) and hands it to the other ranks
 *                        through whatever side channel the host has (the Python layer broadcasts it with torch.distributed);
 *   dep_comm_init      : collective over all ranks (ncclCommInitRank), binds the communicator to HIP device `device`;
 *   dep_comm_allreduce : in-place SUM of n fp32 values, enqueued on `stream` (nothing synchronises);
 *   dep_comm_allreduce_ranges : several ranges as one grouped RCCL operation (one launch);
 *   dep_comm_destroy   : releases the communicator. */
typedef struct dep_comm dep_comm;
int dep_comm_available(void);
int dep_comm_unique_id(void* id_out, size_t bytes);                    /* bytes >= 128 */
int dep_comm_init(dep_comm** comm, int world, int rank, const void* unique_id, size_t id_bytes, int device);
int dep_comm_world(const dep_comm* comm);
int dep_comm_rank(const dep_comm* comm);
int dep_comm_allreduce(dep_comm* comm, float* buf, long n, void* stream);
int dep_comm_allreduce_ranges(dep_comm* comm, float* const* bufs, const long* counts, int nranges, void* stream);
int dep_comm_destroy(dep_comm* comm);

/* Gradient exchange overlapped with the backward pass ("overlapped with layer-0 backward").  range_ptr[l] /
 * range_count[l] name the contiguous span of the caller's flat gradient buffer that is FINAL once layer l's weight
 * gradients are written (the layer's own four tensors per direction plus whatever neighbours of the bucket were already
 * final, e.g. the head's gradients next to the top layer); count 0 skips a layer.  dep_rnn_backward_overlapped is
 * dep_rnn_backward plus: once layer l's dW are enqueued on `stream` AND the sweep of the layer below has been enqueued behind
 * them, `comm_stream` is made to wait for that point (event) and the SUM all-reduce of range l is enqueued there -- it runs
 * beside the weight-gradient GEMMs of the layer below (the bottom layer's range goes out at once).  The caller makes its
 * compute stream wait for `comm_stream` before the optimizer reads the gradients.
 * Co-scheduling: a cluster sweep needs all of its workgroups resident (one per CU, most of the CU's registers and LDS); a
 * collective kernel that holds CUs when a sweep is dispatched would stall it until the collective's peers let it finish.  So
 * collectives are ordered behind the sweeps and only ever run beside GEMMs (DEP_COMM_OVERLAP=sweep restores the earlier
 * enqueue point, beside the next backward sweep); the forward sweeps never have a collective beside them because the
 * optimizer step that precedes the next forward waits for the communication stream. */
typedef struct {
    dep_comm* comm;
    void* comm_stream;           /* hipStream_t, different from the compute stream */
    float* range_ptr[8];         /* indexed by layer */
    long range_count[8];
} dep_grad_sync;
int dep_rnn_backward_overlapped(const dep_rnn_desc* d, const float* x, const float* const* weights,
                                const float* dy, const float* dpooled, const float* dh_n,
                                float* const* dweights, float* dx,
                                void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                                void* stream, const dep_grad_sync* gs);

/* ------------------------------------------------------------------ dense --------- */
/* C[M,N] = opA(A)[M,K] * opB(B)[K,N] + bias[N] + beta*C      (fp32 MFMA, exact f32 products)
 *   transA = 0: A is (M,K) row-major ; 1: A is stored (K,M)
 *   transB = 0: B is stored (K,N)    ; 1: B is stored (N,K)   [nn.Linear weight layout]
 * Replaces nn.Linear forward/backward (audio_gru_whole.py:67,70) and the time-parallel
 * input-projection / weight-gradient contractions inside nn.GRU / nn.LSTM.
 * seq_T/shiftB: when seq_T > 0 and transB == 0, row r of B is read from row r+shiftB and is taken
 * as zero when (r % seq_T)+shiftB falls outside [0,seq_T)  (h_{t-1} operand of dW_hh).
 * workspace is needed for split-K (dep_gemm_workspace_bytes), may be NULL otherwise. */
size_t dep_gemm_workspace_bytes(int transA, int transB, int M, int N, int K);
int dep_gemm_f32(int transA, int transB, int M, int N, int K,
                 const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                 const float* bias, float beta, int seq_T, int shiftB,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Same contract with split-precision products: every fp32 operand element is split on the fly into
 * bf16 hi + bf16 lo and a*b is formed as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 matrix cores with
 * fp32 accumulation (relative error per product ~1e-5, inside the path's 1e-4 parity budget; 5x the MFMA
 * rate of the exact kernel).  dep_rnn_forward/backward use it for their time-parallel contractions of at
 * least `min_macs` multiply-adds when the mode is 1 (default; DEP_GEMM_MODE=f32 or dep_set_gemm_mode(0,-1)
 * selects the exact fp32 kernel everywhere).  dep_gemm_f32 itself is always exact. */
int dep_gemm_bf16x3(int transA, int transB, int M, int N, int K,
                    const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    const float* bias, float beta, int seq_T, int shiftB,
                    void* workspace, size_t workspace_bytes, void* stream);
/* Same contract, precision chosen exactly as dep_rnn_forward / dep_rnn_backward choose it for a contraction of this size under
 * the current dep_set_gemm_mode (0: exact, 1: three-term split above min_macs, 2: single bf16 products).  dep_gemm_bf16x3 above
 * is ALWAYS the three-term split, whatever the mode. */
int dep_gemm(int transA, int transB, int M, int N, int K,
             const float* A, int lda, const float* B, int ldb, float* C, int ldc,
             const float* bias, float beta, int seq_T, int shiftB,
             void* workspace, size_t workspace_bytes, void* stream);
/* Experiment hook (tools/exp_overlap2.py): confine the working workgroups of the calling thread's next
 * split-precision GEMMs to XCDs [lo, lo + n) -- block index % 8 is the XCD.  (0, 8) = whole chip, the default. */
int dep_gemm_set_xcds(int lo, int n);
/* mode 2 (DEP_GEMM_MODE=bf16) is the THROUGHPUT mode BASELINE configs[1] calls "bf16": the large contractions form a_hi * b_hi only
 * (plain bf16 products, fp32 accumulation; a third of the MFMAs).  Relative error per product ~4e-3: it cannot meet the path's
 * 1e-4 parity bar, is never the default and is benchmarked on its own labelled line (bench.py extra.bf16_products).
 * mode 3 (DEP_GEMM_MODE=bf16s) adds bf16 STORAGE to mode 2 for the stack whose kernels have the variants (2-layer GRU, H = 256, T even,
 * the exclusive fused forward): the hidden sequences y / dropout(y), hn and the gate gradients live in the reserve / workspace as bf16
 * (2-byte elements at the positions of the fp32 arrays; the gate gradients as the hi rows of the PK image), the saved gates as 16-bit
 * fixed point; state, accumulation and the recurrence stay fp32.  dep_rnn_forward then takes y == NULL only and
 * dep_rnn_reserve_y_offset points at bf16 data.  Other stacks run mode 3 exactly like mode 2.  Gradients within ~1e-2 of their scale
 * (tests/test_presplit_gpu.py); bench.py extra.bf16_storage. */
int dep_set_gemm_mode(int mode, long min_macs);   /* PROCESS-GLOBAL. mode 0 exact f32 | 1 bf16x3 split | 2 bf16 products | 3 bf16 products + storage ; min_macs < 0 keeps it */
int dep_get_gemm_mode(void);

/* nn.LayerNorm(F) over the last axis (audio_gru_whole.py:62,104). rows = B*T.
 * mean_rstd: (rows,2) saved statistics (may be NULL in inference). */
int dep_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                      float* mean_rstd, int rows, int F, float eps, void* stream);
/* gamma == beta == NULL: plain x-hat (no affine), used with the fold below.
 *
 * LayerNorm feeding a linear map (ln -> gru.weight_ih_l0, audio_gru_whole.py:104-105) with the affine folded into the map:
 *     (xhat*gamma + beta) W^T + b  ==  xhat (W*gamma)^T + (b + W beta)
 * dep_ln_fold_fwd builds Wf (J,F) = W*gamma and bf (J) = b + W beta; the recurrent stack then runs on x-hat with
 * (Wf, bf) in place of (W, b) and its backward returns dWf, dbf; dep_ln_fold_bwd turns those into
 *     dW = dWf*gamma + dbf beta^T, db = dbf, dgamma[f] = sum_j dWf[j,f] W[j,f], dbeta[f] = sum_j W[j,f] dbf[j]
 * -- the gradients nn.LayerNorm + nn.GRU would produce, without ever forming dL/d(xn) (B*T x F) or walking it. */
int dep_ln_fold_fwd(const float* W, const float* b, const float* gamma, const float* beta,
                    float* Wf, float* bf, int J, int F, void* stream);
int dep_ln_fold_bwd(const float* W, const float* dWf, const float* dbf, const float* gamma, const float* beta,
                    float* dW, float* db, float* dgamma, float* dbeta, int J, int F, void* stream);
/* dgamma/dbeta are written; dx may be NULL (the reference never consumes it).
 * workspace: dep_layernorm_bwd_workspace_bytes. */
size_t dep_layernorm_bwd_workspace_bytes(int rows, int F);
int dep_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean_rstd,
                      float* dx, float* dgamma, float* dbeta, int rows, int F,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ attention ------ */
/* attention_net_with_w (text_bilstm_whole.py:74-99).
 *   out (B,T,2H), h_n (K,B,H), Wa (H,H), ba (H)  ->  ctx (B,H)
 *   saved: alpha (B,T), pre (B,H) = Wa*sum_k(h_n)+ba, hsum (B,H)     [needed by the backward] */
int dep_attn_fwd(const float* out, const float* h_n, int K, const float* Wa, const float* ba,
                 float* ctx, float* alpha, float* pre, float* hsum, int B, int T, int H,
                 void* stream);
/*   dctx (B,H) -> dout (B,T,2H) written, dh_n (K,B,H) written, dWa (H,H), dba (H) written.
 *   workspace: dep_attn_bwd_workspace_bytes. */
size_t dep_attn_bwd_workspace_bytes(int B, int T, int H);
int dep_attn_bwd(const float* dctx, const float* out, const float* Wa, const float* alpha,
                 const float* pre, const float* hsum, int K,
                 float* dout, float* dh_n, float* dWa, float* dba, int B, int T, int H,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ heads / losses - */
/* Dropout as nn.Dropout(p) in training mode: y = x * m / (1-p), m ~ Bernoulli(1-p) from
 * Philox4x32-10(seed, site, element index).  The same call on a gradient applies the same mask.
 * p == 0 copies.  x may alias y.  */
int dep_dropout(const float* x, float* y, long n, float p, uint64_t seed, uint32_t site, void* stream);
/* The pre-scaled mask itself (0 or 1/(1-p)), for tests that feed the oracle the same masks. */
int dep_dropout_mask(float* mask, long n, float p, uint64_t seed, uint32_t site, void* stream);
/* a = relu(z) then dropout (nn.ReLU + nn.Dropout, audio_gru_whole.py:68-69); z and a may alias. */
int dep_relu_dropout_fwd(const float* z, float* a, long n, float p, uint64_t seed, uint32_t site, void* stream);
/* dz = da * mask * (z > 0) */
int dep_relu_dropout_bwd(const float* da, const float* z, float* dz, long n, float p, uint64_t seed,
                         uint32_t site, void* stream);
/* column sums of a (M,N) matrix (bias gradients): out[n] = sum_m x[m*ld+n] */
int dep_colsum(const float* x, int M, int N, int ld, float* out, void* stream);

enum { DEP_LOSS_CE_ON_SOFTMAX = 0, DEP_LOSS_L1_RELU = 1, DEP_LOSS_SMOOTHL1_RELU = 2, DEP_LOSS_CE_LOGITS = 3,
       DEP_LOSS_SMOOTHL1 = 4,
       DEP_LOSS_LABELS_I64 = 0x100 /* OR into a CE kind: target holds int64 labels (torch.long, as the reference's loops make
                                      them: audio_gru_whole.py:178) instead of int32 -- no cast kernel per mini-batch */ };
/* Output nonlinearity + loss + its gradient w.r.t. the pre-activation z (B,C), one kernel:
 *   CE_ON_SOFTMAX : out = softmax(z) ; loss = CrossEntropyLoss(out, y)  (double softmax,
 *                   audio_gru_whole.py:72,188,308)              target = int32 labels (int64 with DEP_LOSS_LABELS_I64)
 *   L1_RELU       : out = relu(z) ; L1Loss(out, y)   (audio_bilstm_perm.py:91,251)  target = float
 *   SMOOTHL1_RELU : out = relu(z) ; SmoothL1Loss(out, y) (text_bilstm_perm.py:247)   target = float
 *   CE_LOGITS / SMOOTHL1 : plain losses on z (MyLoss halves, fuse_net_whole.py:384-395)
 * out (B,C) written; loss_rows (B) per-row loss; dz (B,C) = dLoss/dz with Loss = sum_rows / norm
 * (norm = global batch size so that data-parallel shards sum to the reference's batch mean);
 * dz may be NULL (evaluate); target may be NULL when dz and loss_rows are NULL (pure forward). */
int dep_head_loss(int kind, const float* z, const void* target, float* out, float* loss_rows,
                  float* dz, int B, int C, float norm, void* stream);
/* loss = sum(loss_rows[0..B)) / norm, deterministic single-block tree; result on device. */
int dep_reduce_loss(const float* loss_rows, int B, float norm, float* loss_out, int accumulate, void* stream);

/* The models' MLP head as three launches (one forward, two backward) instead of fifteen:
 *     [Dropout(p)] -> Linear(Hin,H1) -> ReLU -> Dropout(p) -> [Linear(H1,C)]
 * (fc_audio: Classification/audio_gru_whole.py:66-73, Regression/audio_bilstm_perm.py:60-67 -- first_dropout = 1;
 *  fc_out: Classification/text_bilstm_whole.py:60-66 -- first_dropout = 0).  Exact fp32, fixed summation order; the masks are the
 * draws dep_dropout / dep_relu_dropout_* make at (seed, site0) / (seed, site1).  W1 (H1,Hin), W2 (C,H1) row-major as
 * torch.nn.Linear holds them; W1 16-byte aligned.  dep_head_mlp_supported says which widths are covered (Hin and H1 powers of
 * two in [8, 256], C <= 16); other shapes are composed from dep_gemm_f32 / dep_relu_dropout_* / dep_colsum.
 *   fwd : a0 (B,Hin) = dropout(x) (written only if first_dropout && p > 0; otherwise x itself is the saved input),
 *         z1, a1 (B,H1) saved for the backward, z2 (B,C) the pre-activation dep_head_loss takes (C = 0: stop at a1).
 *   bwd : dz2 (B,C) from dep_head_loss; a0 = what the forward saved (x when no first dropout ran); dW1, db1, dW2, db2
 *         OVERWRITTEN; dx (B,Hin) or NULL; dz1 (B,H1) scratch. */
int dep_head_mlp_supported(int Hin, int H1, int C);
int dep_head_mlp_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, float* a0,
                     float* z1, float* a1, float* z2, int B, int Hin, int H1, int C, float p, uint64_t seed,
                     uint32_t site0, uint32_t site1, int first_dropout, void* stream);
int dep_head_mlp_bwd(const float* dz2, const float* a0, const float* z1, const float* a1, const float* W1,
                     const float* W2, float* dW1, float* db1, float* dW2, float* db2, float* dx, float* dz1, int B,
                     int Hin, int H1, int C, float p, uint64_t seed, uint32_t site0, uint32_t site1, int first_dropout,
                     void* stream);

/* ------------------------------------------------------------------ optimizer ------ */
/* torch.optim.Adam / AdamW update on a contiguous parameter range (audio_gru_whole.py:307 AdamW
 * with two weight-decay groups; audio_bilstm_perm.py:250 Adam).  step is the 1-based count. */
int dep_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int decoupled, int step, void* stream);

/* ------------------------------------------------------------------ feature front-end */
/* wav2vlad of Classification/audio_features_whole.py:57-72: log-mel spectrogram (librosa.feature.melspectrogram defaults:
 * n_fft 2048, hop 512, periodic Hann, centred frames with reflect padding, power 2, 80 Slaney mel filters) followed by
 * loupe_keras.NetVLAD(feature_size 80, cluster_size 16, output_dim 256).  The five contractions run on dep_gemm_f32
 * (windowed frames x [cos | -sin] DFT basis, power x mel filters, frames x cluster weights, assignment^T x frames,
 * VLAD x hidden weights); these entry points are the passes between them.
 *   dep_frame_window   : out (n_frames, n_fft) = reflect-padded y framed at `hop`, times the Hann window
 *   dep_power_spectrum : reim (rows, ld) = [re(0..bins) | im(0..bins)] -> power (rows, bins) = re^2 + im^2
 *   dep_log_floor      : y = log(max(floor, x))                       (np.log(np.maximum(1e-6, melspec)))
 *   dep_row_softmax    : softmax over the last axis, C <= 64          (NetVLAD soft assignment)
 *   dep_vlad_normalize : vkf (K,F) = assignment^T x frames, a_sum (K), w2 (F,K) -> out (F*K) = l2norm_all(l2norm_f(vkf^T - a_sum*w2)) */
int dep_frame_window(const float* y, long n, int n_fft, int hop, int n_frames, float* out, void* stream);
int dep_power_spectrum(const float* reim, int rows, int bins, int ld, float* power, void* stream);
int dep_log_floor(const float* x, float* y, long n, float floor_value, void* stream);
int dep_row_softmax(const float* z, float* p, int rows, int C, void* stream);
int dep_vlad_normalize(const float* vkf, const float* a_sum, const float* w2, float* out, int F, int K, void* stream);

/* ------------------------------------------------------------------ profiling ------ */
/* Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the
 * roofline figure).  Categories: 0 GRU fwd sweep, 1 GRU bwd sweep, 2 LSTM fwd sweep, 3 LSTM bwd sweep,
 * 4 GEMM NT (input projection / Linear), 5 GEMM NN (dX), 6 GEMM TN (weight gradients; the split-K
 * reduce pass is not included).  dep_profile_read sums launch durations (ms) and counts per
 * category, then resets. */
int dep_profile_enable(int on);
int dep_profile_read(double* total_ms, int* counts, int ncat);

/* Launch-instance log (test infrastructure of the parity suite, tests/test_instance_coverage_gpu.py): while enabled, every
 * kernel launch of the library records its template instance (kernel expression + launcher signature), once per distinct
 * instance.  dep_instance_log_enable(1) clears and starts, (0) stops.  dep_instance_log_read copies the newline-separated
 * list into buf (NUL-terminated, truncated to cap; buf may be NULL) and returns the bytes the full list needs; reset != 0
 * empties the log afterwards.  Process-wide, thread-safe; one relaxed atomic load per launch when off. */
int dep_instance_log_enable(int on);
long dep_instance_log_read(char* buf, long cap, int reset);

/* Enqueue-order log (test infrastructure of the data-parallel path, tests/test_dp_gpu.py): while enabled, every kernel launch and every
 * collective the library enqueues ("K <kernel>" / "C <what> n=<floats>") is appended IN HOST ENQUEUE ORDER, together with the notes the
 * host adds through dep_order_log_note ("N <text>": the stream joins, the phases of a train step).  The two launches that need every CU
 * (the fused GRU forward / backward) must never have a collective enqueued in front of them that is not joined: the test asserts that on
 * this log.  dep_order_log_enable(1) clears and starts, (0) stops; dep_order_log_read as dep_instance_log_read (newline-separated). */
int dep_order_log_enable(int on);
int dep_order_log_note(const char* text);
long dep_order_log_read(char* buf, long cap, int reset);

/* ------------------------------------------------------------------ misc ----------- */
int dep_fill(float* p, long n, float value, void* stream);
/* y = a*x + b*y */
int dep_axpby(const float* x, float* y, long n, float a, float b, void* stream);
/* sigmoid gating of the regression fusion head: y = sigmoid(g) * x (Regression/fuse_net.py:345-351) */
int dep_sigmoid_gate(const float* g, const float* x, float* y, long n, void* stream);
/* Host-loop bookkeeping of train() / evaluate() as HIP kernels (round 4), so that the tensors really are storage only:
 *   dep_gather_rows  : dst[r] = src[idx[r]] for nrows <= 65535 rows of row_floats floats -- the mini-batch
 *                      X_train[lo:hi] of Classification/audio_gru_whole.py:170-172 out of the HBM-resident corpus
 *   dep_copy2d       : dst[r*ldd + c] = src[r*lds + c] -- torch.cat((text_feature, audio_feature), dim=1), fuse_net_whole.py:434
 *   dep_argmax_count : pred[b] = first arg-max of row b (output.data.max(1)[1]); *count += #(pred == label)
 *                      (pred.eq(y).sum(), audio_gru_whole.py:185-187); labels int32 or (labels_i64) int64; pred / count may be NULL */
/* Epoch-level loss bookkeeping in ONE launch per step (nn.LossSum): acc[0] += *loss in float64 (`total_loss += loss.item()`,
 * audio_gru_whole.py:195, without the per-step host sync), acc[1] = max(acc[1], *status), acc[2] = max(acc[2], *soft) -- the cluster
 * sweeps' status word (dep_rnn_status) and fallback word (dep_rnn_set_exclusive) of the step.  Any of loss / status / soft may be NULL. */
int dep_loss_accumulate(const float* loss, const unsigned* status, const unsigned* soft, double* acc, void* stream);
int dep_gather_rows(const float* src, const long long* idx, float* dst, long nrows, long row_floats, void* stream);
int dep_copy2d(const float* src, long lds, float* dst, long ldd, long rows, long cols, void* stream);
int dep_argmax_count(const float* p, const void* labels, int labels_i64, int B, int C, long long* count, long long* pred,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEP_RNN_H */
