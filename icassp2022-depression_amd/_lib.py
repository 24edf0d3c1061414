"""ctypes binding of libdep_rnn.so (include/dep_rnn.h).

PyTorch-ROCm tensors are storage only: every wrapper passes `tensor.data_ptr()` and the current
HIP stream.  There is NO fallback: if the library is missing or a call fails this module raises.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DEP_LIB_PATH') or os.path.join(HERE, 'libdep_rnn.so')       # DEP_LIB_PATH: A/B runs against another build of the same ABI

CELL_GRU, CELL_LSTM = 0, 1
POOL_NONE, POOL_MEAN, POOL_SUM = 0, 1, 2
LOSS_CE_ON_SOFTMAX, LOSS_L1_RELU, LOSS_SMOOTHL1_RELU, LOSS_CE_LOGITS, LOSS_SMOOTHL1 = 0, 1, 2, 3, 4
LOSS_LABELS_I64 = 0x100                     # OR into a CE kind: int64 labels read in place
SITE_FC0, SITE_FC1, SITE_FC2, SITE_FC3 = 1, 2, 3, 4


class DepError(RuntimeError):
    pass


class RnnDesc(C.Structure):
    _fields_ = [('cell', C.c_int32), ('B', C.c_int32), ('T', C.c_int32), ('F', C.c_int32), ('H', C.c_int32),
                ('L', C.c_int32), ('dirs', C.c_int32), ('training', C.c_int32), ('dropout_p', C.c_float),
                ('seed', C.c_uint64), ('pool', C.c_int32), ('impl', C.c_int32)]


_P = C.c_void_p


class GradSync(C.Structure):
    """dep_grad_sync (include/dep_rnn.h): per-layer ranges of the flat gradient buffer for the overlapped all-reduce."""
    _fields_ = [('comm', _P), ('comm_stream', _P), ('range_ptr', _P * 8), ('range_count', C.c_long * 8)]


_BWD_ARGS = [C.POINTER(RnnDesc), _P, C.POINTER(_P), _P, _P, _P, C.POINTER(_P), _P, _P, C.c_size_t, _P, C.c_size_t, _P]
_SIGS = {
    'dep_last_error': (C.c_char_p, []),
    'dep_version': (C.c_int, []),
    'dep_arch': (C.c_char_p, []),
    'dep_rnn_reserve_bytes': (C.c_size_t, [C.POINTER(RnnDesc)]),
    'dep_rnn_workspace_bytes': (C.c_size_t, [C.POINTER(RnnDesc)]),
    'dep_rnn_reserve_y_offset': (C.c_size_t, [C.POINTER(RnnDesc), C.c_int]),
    'dep_rnn_reserve_ydrop_offset': (C.c_size_t, [C.POINTER(RnnDesc), C.c_int]),
    'dep_rnn_status': (C.c_int, [C.POINTER(RnnDesc), _P, _P]),
    'dep_rnn_set_exclusive': (C.c_int, [C.c_int]),
    'dep_rnn_get_exclusive': (C.c_int, []),
    'dep_rnn_workspace_xbuf_offset': (C.c_size_t, [C.POINTER(RnnDesc)]),
    'dep_rnn_forward': (C.c_int, [C.POINTER(RnnDesc), _P, C.POINTER(_P), _P, _P, _P, _P, C.c_size_t, _P, C.c_size_t, _P]),
    'dep_rnn_backward': (C.c_int, _BWD_ARGS),
    'dep_rnn_backward_overlapped': (C.c_int, _BWD_ARGS + [C.POINTER(GradSync)]),
    'dep_comm_available': (C.c_int, []),
    'dep_comm_unique_id': (C.c_int, [_P, C.c_size_t]),
    'dep_comm_init': (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, _P, C.c_size_t, C.c_int]),
    'dep_comm_world': (C.c_int, [_P]),
    'dep_comm_rank': (C.c_int, [_P]),
    'dep_comm_allreduce': (C.c_int, [_P, _P, C.c_long, _P]),
    'dep_comm_allreduce_ranges': (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_long), C.c_int, _P]),
    'dep_comm_destroy': (C.c_int, [_P]),
    'dep_gemm_workspace_bytes': (C.c_size_t, [C.c_int] * 5),
    'dep_gemm_f32': (C.c_int, [C.c_int] * 5 + [_P, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_float, C.c_int, C.c_int,
                               _P, C.c_size_t, _P]),
    'dep_gemm_bf16x3': (C.c_int, [C.c_int] * 5 + [_P, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_float, C.c_int, C.c_int,
                                  _P, C.c_size_t, _P]),
    'dep_gemm': (C.c_int, [C.c_int] * 5 + [_P, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_float, C.c_int, C.c_int,
                           _P, C.c_size_t, _P]),
    'dep_set_gemm_mode': (C.c_int, [C.c_int, C.c_long]),
    'dep_get_gemm_mode': (C.c_int, []),
    'dep_layernorm_fwd': (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P]),
    'dep_ln_fold_fwd': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    'dep_ln_fold_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    'dep_layernorm_bwd_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'dep_layernorm_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    'dep_attn_fwd': (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    'dep_attn_bwd_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'dep_attn_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P,
                               C.c_size_t, _P]),
    'dep_dropout': (C.c_int, [_P, _P, C.c_long, C.c_float, C.c_uint64, C.c_uint32, _P]),
    'dep_dropout_mask': (C.c_int, [_P, C.c_long, C.c_float, C.c_uint64, C.c_uint32, _P]),
    'dep_relu_dropout_fwd': (C.c_int, [_P, _P, C.c_long, C.c_float, C.c_uint64, C.c_uint32, _P]),
    'dep_relu_dropout_bwd': (C.c_int, [_P, _P, _P, C.c_long, C.c_float, C.c_uint64, C.c_uint32, _P]),
    'dep_colsum': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    'dep_head_loss': (C.c_int, [C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P]),
    'dep_reduce_loss': (C.c_int, [_P, C.c_int, C.c_float, _P, C.c_int, _P]),
    'dep_gemm_set_xcds': (C.c_int, [C.c_int, C.c_int]),
    'dep_head_mlp_supported': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'dep_head_mlp_fwd': (C.c_int, [_P] * 9 + [C.c_int] * 4 + [C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _P]),
    'dep_head_mlp_bwd': (C.c_int, [_P] * 12 + [C.c_int] * 4 + [C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _P]),
    'dep_adam_step': (C.c_int, [_P, _P, _P, _P, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_int, C.c_int, _P]),
    'dep_frame_window': (C.c_int, [_P, C.c_long, C.c_int, C.c_int, C.c_int, _P, _P]),
    'dep_power_spectrum': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    'dep_log_floor': (C.c_int, [_P, _P, C.c_long, C.c_float, _P]),
    'dep_row_softmax': (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    'dep_vlad_normalize': (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P]),
    'dep_profile_enable': (C.c_int, [C.c_int]),
    'dep_profile_read': (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]),
    'dep_instance_log_enable': (C.c_int, [C.c_int]),
    'dep_instance_log_read': (C.c_long, [C.c_char_p, C.c_long, C.c_int]),
    'dep_order_log_enable': (C.c_int, [C.c_int]),
    'dep_order_log_note': (C.c_int, [C.c_char_p]),
    'dep_order_log_read': (C.c_long, [C.c_char_p, C.c_long, C.c_int]),
    'dep_fill': (C.c_int, [_P, C.c_long, C.c_float, _P]),
    'dep_axpby': (C.c_int, [_P, _P, C.c_long, C.c_float, C.c_float, _P]),
    'dep_sigmoid_gate': (C.c_int, [_P, _P, _P, C.c_long, _P]),
    'dep_loss_accumulate': (C.c_int, [_P, _P, _P, _P, _P]),
    'dep_gather_rows': (C.c_int, [_P, _P, _P, C.c_long, C.c_long, _P]),
    'dep_copy2d': (C.c_int, [_P, C.c_long, _P, C.c_long, C.c_long, C.c_long, _P]),
    'dep_argmax_count': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Load the shared library (once).  Raises DepError loudly if it is absent -- no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DepError(f'{LIB_PATH} not found: build it with `python icassp2022-depression_amd/build_ext.py` '
                       '(hipcc --offload-arch=gfx950). The HIP path has no fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)       # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    assert t.is_cuda, 'device tensor expected'
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise DepError(f'{what} failed ({rc}): {load().dep_last_error().decode()}')


def f32(t):
    assert t.dtype == torch.float32 and t.is_contiguous()
    return t


# ----------------------------------------------------------------------------- thin wrappers
def gemm(transA, transB, M, N, K, A, lda, B, ldb, Cm, ldc, bias=None, beta=0.0, seq_T=0, shiftB=0, ws=None):
    lib = load()
    wsb = ws.numel() * ws.element_size() if ws is not None else 0
    check(lib.dep_gemm_f32(transA, transB, M, N, K, _ptr(A), lda, _ptr(B), ldb, _ptr(Cm), ldc, _ptr(bias), beta,
                           seq_T, shiftB, _ptr(ws), wsb, stream()), 'dep_gemm_f32')


def gemm_split(transA, transB, M, N, K, A, lda, B, ldb, Cm, ldc, bias=None, beta=0.0, seq_T=0, shiftB=0, ws=None):
    """dep_gemm_bf16x3: the 3-term bf16 split-precision kernel (fp32 in / fp32 accumulate)."""
    lib = load()
    wsb = ws.numel() * ws.element_size() if ws is not None else 0
    check(lib.dep_gemm_bf16x3(transA, transB, M, N, K, _ptr(A), lda, _ptr(B), ldb, _ptr(Cm), ldc, _ptr(bias), beta,
                              seq_T, shiftB, _ptr(ws), wsb, stream()), 'dep_gemm_bf16x3')


def gemm_auto(transA, transB, M, N, K, A, lda, B, ldb, Cm, ldc, bias=None, beta=0.0, seq_T=0, shiftB=0, ws=None):
    """dep_gemm: the precision dep_rnn_* would use for this contraction under the current dep_set_gemm_mode."""
    lib = load()
    wsb = ws.numel() * ws.element_size() if ws is not None else 0
    check(lib.dep_gemm(transA, transB, M, N, K, _ptr(A), lda, _ptr(B), ldb, _ptr(Cm), ldc, _ptr(bias), beta,
                       seq_T, shiftB, _ptr(ws), wsb, stream()), 'dep_gemm')


def set_gemm_mode(mode, min_macs=-1):
    check(load().dep_set_gemm_mode(int(mode), int(min_macs)), 'dep_set_gemm_mode')


def get_gemm_mode():
    """0 = exact fp32 MFMA everywhere, 1 = 3-term bf16 split for the large contractions and the GRU cluster sweeps."""
    return int(load().dep_get_gemm_mode())


def gemm_ws(transA, transB, M, N, K, device):
    n = load().dep_gemm_workspace_bytes(transA, transB, M, N, K)
    return torch.empty(max(n, 16) // 4, dtype=torch.float32, device=device)


def linear_fwd(x, W, b, out=None):
    """out (M,N) = x (M,K) @ W (N,K)^T + b"""
    M, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    gemm(0, 1, M, N, K, x, x.stride(0), W, W.stride(0), out, out.stride(0), bias=b)
    return out


def layernorm_fwd(x2d, gamma, beta, eps=1e-5, save=True):
    rows, F = x2d.shape
    y = torch.empty_like(x2d)
    mr = torch.empty(rows, 2, dtype=torch.float32, device=x2d.device) if save else None
    check(load().dep_layernorm_fwd(_ptr(x2d), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mr), rows, F, eps, stream()),
          'dep_layernorm_fwd')
    return y, mr


def ln_fold_fwd(W, b, gamma, beta, Wf, bf):
    """Wf = W*gamma, bf = b + W beta (see include/dep_rnn.h: LayerNorm's affine folded into the next linear map)."""
    J, F = W.shape
    check(load().dep_ln_fold_fwd(_ptr(W), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(Wf), _ptr(bf), J, F, stream()),
          'dep_ln_fold_fwd')


def ln_fold_bwd(W, dWf, dbf, gamma, beta, dW, db, dgamma, dbeta):
    J, F = W.shape
    check(load().dep_ln_fold_bwd(_ptr(W), _ptr(dWf), _ptr(dbf), _ptr(gamma), _ptr(beta), _ptr(dW), _ptr(db), _ptr(dgamma),
                                 _ptr(dbeta), J, F, stream()), 'dep_ln_fold_bwd')


def layernorm_bwd(dy2d, x2d, gamma, mr, dgamma, dbeta, want_dx=False):
    rows, F = x2d.shape
    lib = load()
    ws = torch.empty(lib.dep_layernorm_bwd_workspace_bytes(rows, F) // 4, dtype=torch.float32, device=x2d.device)
    dx = torch.empty_like(x2d) if want_dx else None
    check(lib.dep_layernorm_bwd(_ptr(dy2d), _ptr(x2d), _ptr(gamma), _ptr(mr), _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                rows, F, _ptr(ws), ws.numel() * 4, stream()), 'dep_layernorm_bwd')
    return dx


def dropout(x, y, p, seed, site):
    check(load().dep_dropout(_ptr(x), _ptr(y), x.numel(), p, seed, site, stream()), 'dep_dropout')


def dropout_mask(n, p, seed, site, device):
    m = torch.empty(n, dtype=torch.float32, device=device)
    check(load().dep_dropout_mask(_ptr(m), n, p, seed, site, stream()), 'dep_dropout_mask')
    return m


def relu_dropout_fwd(z, a, p, seed, site):
    check(load().dep_relu_dropout_fwd(_ptr(z), _ptr(a), z.numel(), p, seed, site, stream()), 'dep_relu_dropout_fwd')


def relu_dropout_bwd(da, z, dz, p, seed, site):
    check(load().dep_relu_dropout_bwd(_ptr(da), _ptr(z), _ptr(dz), z.numel(), p, seed, site, stream()),
          'dep_relu_dropout_bwd')


def colsum(x2d, out):
    M, N = x2d.shape
    check(load().dep_colsum(_ptr(x2d), M, N, x2d.stride(0), _ptr(out), stream()), 'dep_colsum')


def head_loss(kind, z, target, out, loss_rows, dz, norm):
    B, Cc = z.shape
    check(load().dep_head_loss(kind, _ptr(z), _ptr(target), _ptr(out), _ptr(loss_rows), _ptr(dz), B, Cc, float(norm),
                               stream()), 'dep_head_loss')


def head_mlp_supported(Hin, H1, Cc):
    """Widths the fused head kernels cover (DEP_HEAD_FUSED=0 keeps the composed launches: A/B switch)."""
    return os.environ.get('DEP_HEAD_FUSED', '1') != '0' and bool(load().dep_head_mlp_supported(Hin, H1, Cc))


def head_mlp_fwd(x, W1, b1, W2, b2, a0, z1, a1, z2, p, seed, sites, first):
    B, Hin = x.shape
    H1 = W1.shape[0]
    Cc = 0 if W2 is None else W2.shape[0]
    check(load().dep_head_mlp_fwd(_ptr(x), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(a0), _ptr(z1), _ptr(a1), _ptr(z2),
                                  B, Hin, H1, Cc, float(p), int(seed), int(sites[0]), int(sites[1]), int(first), stream()),
          'dep_head_mlp_fwd')


def head_mlp_bwd(dz2, a0, z1, a1, W1, W2, dW1, db1, dW2, db2, dx, dz1, p, seed, sites, first):
    B, Hin = a0.shape
    H1, Cc = W1.shape[0], W2.shape[0]
    check(load().dep_head_mlp_bwd(_ptr(dz2), _ptr(a0), _ptr(z1), _ptr(a1), _ptr(W1), _ptr(W2), _ptr(dW1), _ptr(db1), _ptr(dW2),
                                  _ptr(db2), _ptr(dx), _ptr(dz1), B, Hin, H1, Cc, float(p), int(seed), int(sites[0]),
                                  int(sites[1]), int(first), stream()), 'dep_head_mlp_bwd')


def reduce_loss(loss_rows, norm, loss_out, accumulate=False):
    check(load().dep_reduce_loss(_ptr(loss_rows), loss_rows.numel(), float(norm), _ptr(loss_out), int(accumulate),
                                 stream()), 'dep_reduce_loss')


def adam_step(p, g, m, v, lr, b1, b2, eps, wd, decoupled, step):
    check(load().dep_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, b1, b2, eps, wd, int(decoupled),
                               int(step), stream()), 'dep_adam_step')


def fill(t, value):
    check(load().dep_fill(_ptr(t), t.numel(), float(value), stream()), 'dep_fill')


def axpby(x, y, a, b):
    check(load().dep_axpby(_ptr(x), _ptr(y), x.numel(), float(a), float(b), stream()), 'dep_axpby')


def sigmoid_gate(g, x, y):
    check(load().dep_sigmoid_gate(_ptr(g), _ptr(x), _ptr(y), x.numel(), stream()), 'dep_sigmoid_gate')


def loss_accumulate(loss, status, soft, acc):
    """acc (3,) float64 on the device: [sum of step losses, max status word, max fallback word] (dep_loss_accumulate)."""
    check(load().dep_loss_accumulate(_ptr(loss), _ptr(status), _ptr(soft), _ptr(acc), stream()), 'dep_loss_accumulate')


def gather_rows(src, idx, out=None):
    """src (N, ...) fp32 contiguous, idx (n,) int64 on the device -> (n, ...) rows (dep_gather_rows)."""
    n = idx.numel()
    assert src.dtype == torch.float32 and src.dim() >= 1
    row = int(src.numel() // src.shape[0]) if src.shape[0] else 0          # (an empty corpus has no src[0])
    if n and src.shape[0] == 0:
        raise IndexError('gather_rows: indices into an empty array')
    if os.environ.get('DEP_DEBUG_BOUNDS', '0') == '1' and n:               # debug: the range check index_select used to make (one host sync)
        lo_, hi_ = int(idx.min()), int(idx.max())
        if lo_ < 0 or hi_ >= src.shape[0]:
            raise IndexError(f'gather_rows: index {lo_ if lo_ < 0 else hi_} out of range for {src.shape[0]} rows')
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    if n:
        assert src.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous()
        for r0 in range(0, n, 65535):
            r1 = min(n, r0 + 65535)
            check(load().dep_gather_rows(_ptr(src), idx.data_ptr() + 8 * r0, out.data_ptr() + 4 * row * r0, r1 - r0, row, stream()),
                  'dep_gather_rows')
    return out


def concat2(a, b):
    """torch.cat((a, b), dim=1) of two (B, .) fp32 matrices as two strided copies (dep_copy2d)."""
    B, na = a.shape
    nb = b.shape[1]
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and b.shape[0] == B and (B == 0 or (a.stride(1) == 1 and b.stride(1) == 1)), \
        'concat2: two fp32 matrices with unit inner stride and equal row counts'
    out = torch.empty(B, na + nb, dtype=torch.float32, device=a.device)
    if B:
        lib = load()
        check(lib.dep_copy2d(_ptr(a), a.stride(0), _ptr(out), na + nb, B, na, stream()), 'dep_copy2d')
        check(lib.dep_copy2d(_ptr(b), b.stride(0), out.data_ptr() + 4 * na, na + nb, B, nb, stream()), 'dep_copy2d')
    return out


def argmax_count(probs, labels=None, count=None, want_pred=False):
    """First arg-max per row of `probs` (B, C) -> int64 (B, 1) when want_pred; `count` (0-dim int64 device tensor) += the
    number of rows whose arg-max equals `labels` (int32 / int64 device tensor).  One launch (dep_argmax_count)."""
    B, Cc = probs.shape
    pred = torch.empty(B, 1, dtype=torch.int64, device=probs.device) if want_pred else None
    if B:
        assert probs.is_contiguous() and probs.dtype == torch.float32
        i64 = 0
        if labels is not None:
            assert labels.is_cuda and labels.is_contiguous() and labels.dtype in (torch.int64, torch.int32)
            i64 = int(labels.dtype == torch.int64)
        check(load().dep_argmax_count(_ptr(probs), _ptr(labels), i64, B, Cc, _ptr(count) if labels is not None else None,
                                      _ptr(pred), stream()), 'dep_argmax_count')
    return pred


def attn_fwd(out, h_n, Wa, ba):
    B, T, H2 = out.shape
    H = H2 // 2
    K = h_n.shape[0]
    dev = out.device
    ctx = torch.empty(B, H, dtype=torch.float32, device=dev)
    alpha = torch.empty(B, T, dtype=torch.float32, device=dev)
    pre = torch.empty(B, H, dtype=torch.float32, device=dev)
    hsum = torch.empty(B, H, dtype=torch.float32, device=dev)
    check(load().dep_attn_fwd(_ptr(out), _ptr(h_n), K, _ptr(Wa), _ptr(ba), _ptr(ctx), _ptr(alpha), _ptr(pre),
                              _ptr(hsum), B, T, H, stream()), 'dep_attn_fwd')
    return ctx, (alpha, pre, hsum)


def attn_bwd(dctx, out, Wa, saved, K, dWa, dba):
    alpha, pre, hsum = saved
    B, T, H2 = out.shape
    H = H2 // 2
    lib = load()
    dev = out.device
    dout = torch.empty_like(out)
    dh_n = torch.empty(K, B, H, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.dep_attn_bwd_workspace_bytes(B, T, H) // 4 + 64, dtype=torch.float32, device=dev)
    check(lib.dep_attn_bwd(_ptr(dctx), _ptr(out), _ptr(Wa), _ptr(alpha), _ptr(pre), _ptr(hsum), K, _ptr(dout),
                           _ptr(dh_n), _ptr(dWa), _ptr(dba), B, T, H, _ptr(ws), ws.numel() * 4, stream()),
          'dep_attn_bwd')
    return dout, dh_n


_fallback_noted = [False]


def note_fallback():
    """The exclusive GRU forward gave way to foreign work at least once: results are right (the device redid the forward on
    the tolerant kernels), but every such step pays the hello time-out -- switch the attempt off for the rest of the process."""
    load().dep_rnn_set_exclusive(0)
    if not _fallback_noted[0]:
        _fallback_noted[0] = True
        import warnings
        warnings.warn('icassp2022_depression_amd: the GPU is shared with other kernels -- the fused GRU forward fell back to the '
                      'co-schedule-tolerant sweeps (results unaffected); using those directly from now on '
                      '(dep_rnn_set_exclusive(0); set DEP_EXCLUSIVE=0 to start that way)', RuntimeWarning, stacklevel=3)


class Rnn:
    """One dep_rnn_desc + its reserve/workspace buffers (allocated once per shape, reused per step)."""

    def __init__(self, cell, B, T, F, H, L, dirs, training, dropout_p, pool, device, impl=0):
        self.lib = load()
        self.desc = RnnDesc(cell, B, T, F, H, L, dirs, int(training), float(dropout_p), 0, pool, impl)
        self.device = device
        rb = self.lib.dep_rnn_reserve_bytes(C.byref(self.desc))
        wb = self.lib.dep_rnn_workspace_bytes(C.byref(self.desc))
        if rb == 0 or wb == 0:
            raise DepError(f'bad rnn descriptor: {cell=} {B=} {T=} {F=} {H=} {L=} {dirs=}')
        self.reserve = torch.empty(rb // 4, dtype=torch.float32, device=device)
        self.workspace = torch.empty(wb // 4, dtype=torch.float32, device=device)
        self.n_w = 4 * L * dirs
        self._warr = (_P * self.n_w)()
        self._garr = (_P * self.n_w)()

    def check(self):
        """Synchronise and raise if a cluster sweep gave up on a bounded spin (never silently wrong).  Also the place where the
        host learns that the exclusive forward had to fall back (soft word): it then stops attempting it in this process."""
        check(self.lib.dep_rnn_status(C.byref(self.desc), _ptr(self.workspace), stream()), 'dep_rnn_status')
        w = self.fallback_word()
        if w is not None and self.lib.dep_rnn_get_exclusive() and int(w.item()) != 0:
            note_fallback()

    def fallback_word(self):
        """Device view of the soft word (include/dep_rnn.h, dep_rnn_set_exclusive): non-zero when the forward that just ran on
        this workspace was redone by the co-schedule-tolerant kernels."""
        off = self.lib.dep_rnn_workspace_xbuf_offset(C.byref(self.desc))
        if off == C.c_size_t(-1).value:
            return None
        return self.workspace.view(torch.int32)[off // 4 + 1]

    def status_word(self):
        """Device view (0-dim int32) of the sweeps' status word inside the workspace, or None for layouts without one: lets a
        training loop fold the per-step status into a device-side flag instead of synchronising every step (nn.LossSum)."""
        off = self.lib.dep_rnn_workspace_xbuf_offset(C.byref(self.desc))
        if off == C.c_size_t(-1).value:
            return None
        return self.workspace.view(torch.int32)[off // 4]

    def layer_output(self, layer=None):
        """Zero-copy view of a layer's output sequence (B,T,H*dirs) inside the reserve."""
        d = self.desc
        layer = d.L - 1 if layer is None else layer
        off = self.lib.dep_rnn_reserve_y_offset(C.byref(d), layer) // 4
        n = d.B * d.T * d.H * d.dirs
        return self.reserve[off:off + n].view(d.B, d.T, d.H * d.dirs)

    def layer_output_dropped(self, layer):
        d = self.desc
        off = self.lib.dep_rnn_reserve_ydrop_offset(C.byref(d), layer)
        if off == C.c_size_t(-1).value:
            return None
        off //= 4
        n = d.B * d.T * d.H * d.dirs
        return self.reserve[off:off + n].view(d.B, d.T, d.H * d.dirs)

    def forward(self, x, weights, seed=0, pooled=None, h_n=None, y=None):
        for i, w in enumerate(weights):
            self._warr[i] = w.data_ptr()
        self.desc.seed = seed
        check(self.lib.dep_rnn_forward(C.byref(self.desc), _ptr(x), self._warr, _ptr(y), _ptr(pooled), _ptr(h_n),
                                       _ptr(self.reserve), self.reserve.numel() * 4, _ptr(self.workspace),
                                       self.workspace.numel() * 4, stream()), 'dep_rnn_forward')

    def backward(self, x, weights, dweights, dy=None, dpooled=None, dh_n=None, dx=None, grad_sync=None):
        """grad_sync: a GradSync (data parallel) -> dep_rnn_backward_overlapped: layer l's range of the flat gradient
        buffer is all-reduced on the communication stream beside the weight-gradient GEMMs of the layer below."""
        for i, (w, g) in enumerate(zip(weights, dweights)):
            self._warr[i] = w.data_ptr()
            self._garr[i] = g.data_ptr()
        args = (C.byref(self.desc), _ptr(x), self._warr, _ptr(dy), _ptr(dpooled), _ptr(dh_n), self._garr, _ptr(dx),
                _ptr(self.reserve), self.reserve.numel() * 4, _ptr(self.workspace), self.workspace.numel() * 4, stream())
        if grad_sync is None:
            check(self.lib.dep_rnn_backward(*args), 'dep_rnn_backward')
        else:
            check(self.lib.dep_rnn_backward_overlapped(*args, C.byref(grad_sync)), 'dep_rnn_backward_overlapped')


PROF_CATS = ('gru_fwd_sweep', 'gru_bwd_sweep', 'lstm_fwd_sweep', 'lstm_bwd_sweep', 'gemm_nt', 'gemm_nn', 'gemm_tn')


def instance_log_enable(on=True):
    load().dep_instance_log_enable(int(on))


def instance_log_read(reset=False):
    """Set of kernel template instances launched since the log was enabled (dep_instance_log_read)."""
    lib = load()
    n = lib.dep_instance_log_read(None, 0, 0)
    buf = C.create_string_buffer(n + 1)
    lib.dep_instance_log_read(buf, n + 1, int(reset))
    return set(l for l in buf.value.decode().split('\n') if l)


def profile_enable(on=True):
    load().dep_profile_enable(int(on))


def profile_read():
    """{category: (total_ms, launches)} since the last read (HIP events on the launch stream)."""
    n = len(PROF_CATS)
    ms = (C.c_double * n)(); cnt = (C.c_int * n)()
    load().dep_profile_read(ms, cnt, n)
    return {PROF_CATS[i]: (ms[i], cnt[i]) for i in range(n)}


_order_on = [False]


def order_log_enable(on=True):
    """Enqueue-order log (dep_order_log_*): kernel launches, collectives and host notes in host enqueue order."""
    load().dep_order_log_enable(int(on)); _order_on[0] = bool(on)


def order_note(text):
    if _order_on[0]:
        load().dep_order_log_note(text.encode())


def order_log_read(reset=False):
    lib = load()
    n = lib.dep_order_log_read(None, 0, 0)
    buf = C.create_string_buffer(n + 1)
    lib.dep_order_log_read(buf, n + 1, int(reset))
    return [l for l in buf.value.decode().split('\n') if l]
