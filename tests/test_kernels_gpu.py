"""GPU parity tests of the C-ABI operators (through ctypes) against the numpy oracle.
Run on the MI355X box:  python -m pytest tests -m gpu -q"""
import numpy as np
import pytest

from oracle import ref_numpy as R

torch = pytest.importorskip('torch')
pytestmark = [pytest.mark.gpu, pytest.mark.oracle]

if torch.cuda.is_available():
    from icassp2022_depression_amd import _lib as L
    DEV = torch.device('cuda:0')


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy().astype(np.float64)


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('tA,tB', [(0, 1), (0, 0), (1, 0)])
# small shapes run the 32x32-tile direct-operand kernel, (1100, 700, 96) and up the LDS-tiled 128x128 kernel
@pytest.mark.parametrize('M,N,K', [(300, 200, 64), (77, 45, 39), (128, 128, 32), (1, 5, 7), (513, 260, 100),
                                   (2, 256, 512), (1100, 700, 96), (1030, 770, 33)])
def test_gemm_shapes(tA, tB, M, N, K):
    rng = np.random.default_rng(M * 7 + N * 3 + K + tA * 2 + tB)
    A = rng.standard_normal((K, M) if tA else (M, K))
    B = rng.standard_normal((N, K) if tB else (K, N))
    bias = rng.standard_normal(N)
    C0 = rng.standard_normal((M, N))
    ref = (A.T if tA else A) @ (B.T if tB else B) + bias + 0.5 * C0
    a, b, c, bi = dev(A), dev(B), dev(C0), dev(bias)
    L.gemm(tA, tB, M, N, K, a, a.stride(0), b, b.stride(0), c, c.stride(0), bias=bi, beta=0.5)
    # compare against the fp32-rounded inputs
    A32, B32, b32, C32 = (x.astype(np.float32).astype(np.float64) for x in (A, B, bias, C0))
    ref = (A32.T if tA else A32) @ (B32.T if tB else B32) + b32 + 0.5 * C32
    assert relerr(host(c), ref) < 2e-6


@pytest.mark.parametrize('tA,tB', [(0, 1), (0, 0), (1, 0)])
@pytest.mark.parametrize('M,N,K', [(300, 200, 64), (77, 45, 39), (513, 260, 100), (256, 384, 2048)])
def test_gemm_bf16x3_split_accuracy(tA, tB, M, N, K):
    """3-term bf16 split: error relative to sum_k |a||b| must stay ~1e-5 (exact-f32 kernel: ~1e-7)."""
    rng = np.random.default_rng(M + N * 3 + K * 5 + tA * 2 + tB)
    A = rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32).astype(np.float64)
    B = rng.standard_normal((N, K) if tB else (K, N)).astype(np.float32).astype(np.float64)
    bias = rng.standard_normal(N).astype(np.float32).astype(np.float64)
    opA = A.T if tA else A; opB = B.T if tB else B
    ref = opA @ opB + bias
    scale = np.abs(opA) @ np.abs(opB)
    a, b, bi = dev(A), dev(B), dev(bias)
    c = torch.empty(M, N, device=DEV)
    ws = L.gemm_ws(tA, tB, M, N, K, DEV)
    L.gemm_split(tA, tB, M, N, K, a, a.stride(0), b, b.stride(0), c, N, bias=bi, ws=ws)
    err = np.abs(host(c) - ref) / scale
    assert err.max() < 1.5e-5, err.max()
    assert err.mean() < 2e-6


@pytest.mark.parametrize('tA,tB,M,N,K', [(0, 1, 1200, 768, 256), (0, 0, 1300, 260, 768), (1, 0, 768, 256, 4096)])
def test_gemm_bf16_products_mode_has_its_own_tolerance(tA, tB, M, N, K):
    """dep_set_gemm_mode(2): a_hi * b_hi only (the "bf16" throughput mode, never the parity path).  Error relative to
    sum_k |a||b| is ~1e-3 -- two orders above the 3-term split's bound, which proves the mode is really in effect -- and bounded."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32).astype(np.float64)
    B = rng.standard_normal((N, K) if tB else (K, N)).astype(np.float32).astype(np.float64)
    opA = A.T if tA else A; opB = B.T if tB else B
    ref = opA @ opB
    scale = np.abs(opA) @ np.abs(opB)
    a, b = dev(A), dev(B)
    c = torch.empty(M, N, device=DEV)
    ws = L.gemm_ws(tA, tB, M, N, K, DEV)
    L.set_gemm_mode(2, 0)                                   # threshold 0: these test sizes are below the default 2^28 multiply-adds
    try:
        L.gemm_auto(tA, tB, M, N, K, a, a.stride(0), b, b.stride(0), c, N, ws=ws)       # the mode-following entry (what dep_rnn_* runs)
        err = np.abs(host(c) - ref) / scale
        assert 2e-5 < err.max() < 1e-2, err.max()
        # the explicit three-term entry keeps its contract whatever the process mode is
        L.gemm_split(tA, tB, M, N, K, a, a.stride(0), b, b.stride(0), c, N, ws=ws)
        assert (np.abs(host(c) - ref) / scale).max() < 1.5e-5
    finally:
        L.set_gemm_mode(1, 1 << 28)
    L.gemm_split(tA, tB, M, N, K, a, a.stride(0), b, b.stride(0), c, N, ws=ws)
    assert (np.abs(host(c) - ref) / scale).max() < 1.5e-5          # and the default mode is back


# Round 6: with scratch for the weight's stage image the three-term NT projection takes the LDS-DMA kernel (gemm_bf16x3_nt_dma: M % 128 == 0,
# N % 256 == 0, N <= 1024, K % 16 == 0, K >= 48); without scratch the register-staged kernel runs.  Same products in the same order: bit-identical --
# and both inside the split's error bound against the float64 product.  Row-strided operands and outputs (lda > K, ldc > N), no bias, the K minimum.
@pytest.mark.parametrize('M,N,K,lda,ldc,bias', [(1280, 768, 256, 256, 768, True), (2560, 1024, 1024, 1024, 1024, True), (4096, 256, 48, 64, 260, False),
                                                (2048, 512, 1040, 1040, 512, True), (33024, 768, 256, 260, 768, True)])
def test_nt_projection_fed_by_lds_dma_is_bit_identical_to_the_staged_kernel(M, N, K, lda, ldc, bias):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, lda)).astype(np.float32); B = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
    bv = rng.standard_normal(N).astype(np.float32)
    a, b, bi = dev(A), dev(B), (dev(bv) if bias else None)
    c0 = torch.full((M, ldc), 7.0, device=DEV); c1 = torch.full((M, ldc), 7.0, device=DEV)
    L.gemm_split(0, 1, M, N, K, a, lda, b, K, c0, ldc, bias=bi)                                   # no scratch: the register-staged kernel
    i0 = ' '.join(L.instance_log_read())                                                          # (conftest.py switched the launch-instance log on for this test)
    ws = torch.empty(N * K + 64, device=DEV)
    L.gemm_split(0, 1, M, N, K, a, lda, b, K, c1, ldc, bias=bi, ws=ws)
    i1 = ' '.join(L.instance_log_read())
    assert 'nt_dma' not in i0 and 'gemm_bf16x3_nt_dma' in i1, (i0, i1)
    assert torch.equal(c0, c1)
    if ldc > N:
        assert bool((c1[:, N:] == 7.0).all())                                                    # nothing written beside the tile
    A64 = A[:, :K].astype(np.float64); B64 = B.astype(np.float64)
    ref = A64 @ B64.T + (bv.astype(np.float64) if bias else 0.0)
    scale = np.abs(A64) @ np.abs(B64).T
    assert (np.abs(host(c1[:, :N]) - ref) / scale).max() < 1.5e-5


def test_dma_fed_contractions_repeat_bit_for_bit():
    """The LDS-DMA kernels order their stage reads by counted vmcnt waits and barriers only: a misplaced count shows up as a RARE wrong tile.  The projection
    (cfg2's shape) 40 times and the GRU stack's backward (weight-gradient pairs over the PK image) 12 times on the same inputs: every repeat bit-identical."""
    rng = np.random.default_rng(3)
    M, N, K = 153600, 768, 256
    a = torch.randn(M, K, device=DEV); b = dev(rng.standard_normal((N, K)) / 16.0); bi = dev(rng.standard_normal(N))
    ws = torch.empty(N * K + 64, device=DEV)
    c0 = torch.empty(M, N, device=DEV); c = torch.empty(M, N, device=DEV)
    L.gemm_split(0, 1, M, N, K, a, K, b, K, c0, N, bias=bi, ws=ws)
    assert 'gemm_bf16x3_nt_dma' in ' '.join(L.instance_log_read())
    for _ in range(40):
        c.fill_(0.0)
        L.gemm_split(0, 1, M, N, K, a, K, b, K, c, N, bias=bi, ws=ws)
        assert torch.equal(c, c0)
    del a, c, c0
    B, T, F, H = 512, 300, 256, 256
    g = torch.Generator().manual_seed(5)
    W = []
    for l in range(2):
        for shp in ((3 * H, F if l == 0 else H), (3 * H, H), (3 * H,), (3 * H,)):
            W.append(((torch.rand(*shp, generator=g) * 2 - 1) / 16.0).to(DEV))
    x = torch.randn(B, T, F, generator=g).to(DEV); dpool = torch.randn(B, H, generator=g).to(DEV)
    rnn = L.Rnn(L.CELL_GRU, B, T, F, H, 2, 1, True, 0.5, L.POOL_MEAN, DEV)
    pooled = torch.empty(B, H, device=DEV)
    rnn.forward(x, W, seed=11, pooled=pooled)
    first = None
    for _ in range(12):
        Gd = [torch.full_like(w, float('nan')) for w in W]
        rnn.backward(x, W, Gd, dpooled=dpool)
        rnn.check()
        if first is None:
            first = Gd
            assert 'gemm_bf16x3_tn_dma' in ' '.join(L.instance_log_read())
        else:
            for u, v in zip(first, Gd):
                assert torch.equal(u, v)


def test_gemm_bf16x3_shift_and_splitk_match_exact_kernel():
    rng = np.random.default_rng(15)
    Bsz, T, M, N = 9, 300, 96, 64
    K = Bsz * T
    a, b = dev(rng.standard_normal((K, M))), dev(rng.standard_normal((K, N)))
    ws = L.gemm_ws(1, 0, M, N, K, DEV)
    for shift in (-1, 1):
        c0 = torch.empty(M, N, device=DEV); c1 = torch.empty(M, N, device=DEV)
        L.gemm(1, 0, M, N, K, a, M, b, N, c0, N, seq_T=T, shiftB=shift, ws=ws)
        L.gemm_split(1, 0, M, N, K, a, M, b, N, c1, N, seq_T=T, shiftB=shift, ws=ws)
        assert relerr(host(c1), host(c0)) < 2e-5


def test_gemm_splitk_and_shift():
    """dW_hh-style contraction: C (M,N) = A^T (M,K) * shift(B) (K,N) with K = B*T rows, zero rows at
    sequence starts, split-K with the deterministic two-pass reduction."""
    rng = np.random.default_rng(5)
    Bsz, T, M, N = 9, 300, 96, 64
    K = Bsz * T
    A = rng.standard_normal((K, M)); Bm = rng.standard_normal((K, N))
    a, b = dev(A), dev(Bm)
    for shift in (-1, 1):
        Bs = np.zeros_like(Bm)
        B3 = Bm.reshape(Bsz, T, N).astype(np.float32).astype(np.float64)
        if shift == -1:
            Bs.reshape(Bsz, T, N)[:, 1:] = B3[:, :-1]
        else:
            Bs.reshape(Bsz, T, N)[:, :-1] = B3[:, 1:]
        ref = A.astype(np.float32).astype(np.float64).T @ Bs
        c = torch.full((M, N), 7.0, device=DEV)
        ws = L.gemm_ws(1, 0, M, N, K, DEV)
        assert ws.numel() > 64          # split-K really engaged
        L.gemm(1, 0, M, N, K, a, M, b, N, c, N, seq_T=T, shiftB=shift, ws=ws)
        assert relerr(host(c), ref) < 5e-6
        c2 = torch.empty_like(c)
        L.gemm(1, 0, M, N, K, a, M, b, N, c2, N, seq_T=T, shiftB=shift, ws=ws)
        assert torch.equal(c, c2)       # deterministic


def test_gemm_strided_views():
    """sub-blocks with ld > width (the way dep_rnn_backward slices dGI / writes dW blocks)."""
    rng = np.random.default_rng(6)
    M, N, K, ld = 64, 48, 200, 160
    big = rng.standard_normal((K, ld)); Bm = rng.standard_normal((K, N))
    a, b = dev(big), dev(Bm)
    c = torch.zeros(M, N, device=DEV)
    A_view = a[:, 32:32 + M]
    L.gemm(1, 0, M, N, K, A_view, ld, b, N, c, N)
    ref = big[:, 32:32 + M].astype(np.float32).astype(np.float64).T @ Bm.astype(np.float32).astype(np.float64)
    assert relerr(host(c), ref) < 2e-6


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize('rows,F', [(37, 39), (600, 256), (9, 5), (130, 512), (66, 1024)])
def test_layernorm(rows, F):
    rng = np.random.default_rng(rows + F)
    x = rng.standard_normal((rows, F)) * 2 + 0.3
    g = rng.standard_normal(F); b = rng.standard_normal(F)
    dy = rng.standard_normal((rows, F))
    y, mr = L.layernorm_fwd(dev(x), dev(g), dev(b))
    x32 = x.astype(np.float32).astype(np.float64)
    yr, cache = R.layernorm_fwd(x32, g.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64))
    assert np.abs(host(y) - yr).max() < 2e-5
    dg = torch.empty(F, device=DEV); db = torch.empty(F, device=DEV)
    dx = L.layernorm_bwd(dev(dy), dev(x), dev(g), mr, dg, db, want_dx=True)
    dxr, dgr, dbr = R.layernorm_bwd(dy.astype(np.float32).astype(np.float64), g.astype(np.float32).astype(np.float64), cache)
    assert relerr(host(dx), dxr) < 2e-5
    assert relerr(host(dg), dgr) < 2e-5
    assert relerr(host(db), dbr) < 2e-5


@pytest.mark.parametrize('rows,J,F', [(50, 12, 7), (300, 96, 64), (64, 768, 256)])
def test_ln_fold_matches_explicit_layernorm_chain(rows, J, F):
    """LayerNorm -> linear with the affine folded into the linear map (dep_ln_fold_*) against the oracle's explicit
    chain xn = LN(x); g = xn W^T + b: same forward values, same dW, db, dgamma, dbeta."""
    rng = np.random.default_rng(rows + J + F)
    f64 = lambda a: a.astype(np.float32).astype(np.float64)
    x = f64(rng.standard_normal((rows, F)) * 1.5 + 0.2)
    W = f64(rng.standard_normal((J, F)) * 0.3); b = f64(rng.standard_normal(J))
    g = f64(rng.standard_normal(F) + 1.0); be = f64(rng.standard_normal(F) * 0.5)
    dG = f64(rng.standard_normal((rows, J)))
    # oracle: explicit chain
    xn, cache = R.layernorm_fwd(x, g, be)
    G_ref = xn @ W.T + b
    dxn = dG @ W
    _, dg_ref, dbe_ref = R.layernorm_bwd(dxn, g, cache)
    dW_ref = dG.T @ xn; db_ref = dG.sum(0)
    # device: x-hat, folded pair
    xhat, _ = L.layernorm_fwd(dev(x), None, None, save=False)
    Wf = torch.empty(J, F, device=DEV); bf = torch.empty(J, device=DEV)
    L.ln_fold_fwd(dev(W), dev(b), dev(g), dev(be), Wf, bf)
    G_dev = host(xhat) @ host(Wf).T + host(bf)
    assert relerr(G_dev, G_ref) < 2e-5
    P = dG.T @ host(xhat); q = dG.sum(0)            # what the stack's backward returns for (Wf, bf)
    dW = torch.empty(J, F, device=DEV); db = torch.empty(J, device=DEV)
    dg = torch.empty(F, device=DEV); dbe = torch.empty(F, device=DEV)
    L.ln_fold_bwd(dev(W), dev(P), dev(q), dev(g), dev(be), dW, db, dg, dbe)
    assert relerr(host(dW), dW_ref) < 2e-5
    assert relerr(host(db), db_ref) < 2e-5
    assert relerr(host(dg), dg_ref) < 2e-5
    assert relerr(host(dbe), dbe_ref) < 2e-5


# ----------------------------------------------------------------------------- RNN stacks
def make_rnn_params(rng, cell, F, H, L, dirs):
    G = 3 if cell == 'gru' else 4
    P = {}; names = []
    prefix = 'lstm_net_audio' if cell == 'gru' else 'lstm_net'
    k = 1.0 / np.sqrt(H)
    for l in range(L):
        for d in range(dirs):
            sfx = f'l{l}' + ('_reverse' if d else '')
            inp = F if l == 0 else H * dirs
            for nm, shp in (('weight_ih', (G * H, inp)), ('weight_hh', (G * H, H)), ('bias_ih', (G * H,)), ('bias_hh', (G * H,))):
                key = f'{prefix}.{nm}_{sfx}'
                P[key] = rng.uniform(-k, k, shp).astype(np.float32).astype(np.float64)
                names.append(key)
    return P, names, prefix


RNN_CASES = [
    # cell, B, T, F, H, impl     (impl 1 = generic kernels, 2 = MFMA kernels)
    ('gru', 4, 6, 5, 8, 1),
    ('gru', 6, 20, 24, 16, 1),
    ('gru', 6, 20, 24, 16, 2),
    ('gru', 8, 50, 39, 128, 2),         # BASELINE configs[0] shape
    ('gru', 37, 9, 64, 256, 2),         # ragged batch tile, 2 hidden tiles per wave (cfg2 H)
    ('gru', 5, 7, 12, 48, 2),           # 3 tiles on one wave
    ('gru', 8, 50, 39, 128, 3),         # cluster-parallel sweeps: 4 members per tile
    ('gru', 37, 9, 64, 256, 3),         # 8 members, ragged tile, padded tile count
    ('gru', 130, 12, 32, 256, 3),       # 9 tiles -> 16 padded -> 128 workgroups
    ('gru', 16, 300, 256, 256, 3),      # full T of the benchmark
    ('lstm', 4, 6, 5, 8, 1),
    ('lstm', 6, 20, 24, 16, 1),
    ('lstm', 6, 20, 24, 16, 2),
    ('lstm', 19, 11, 40, 128, 2),       # cfg3 H
    ('lstm', 3, 5, 16, 32, 2),
    ('lstm', 19, 11, 40, 128, 3),       # cluster-parallel BiLSTM sweeps (cfg3 H), ragged tile
    ('lstm', 70, 300, 64, 128, 3),      # 5 tiles -> 8 padded x 2 directions x 4 members, full T
    ('gru', 530, 4, 16, 256, 3),        # more utterances than one co-resident launch holds: chunks of 512 + 18
    ('gru', 1040, 3, 8, 128, 3),        # H = 128: chunks of 1024 + 16
    ('lstm', 530, 3, 8, 128, 3),        # BiLSTM: chunks of 512 + 18
    ('gru', 1, 1, 8, 256, 3),           # degenerate: one utterance, one step (no exchange at all)
    ('gru', 3, 2, 8, 128, 3),           # two steps: a single exchange
    ('lstm', 2, 1, 8, 128, 3),
    ('gru', 2, 1, 5, 16, 2), ('lstm', 1, 1, 3, 8, 1),
    ('gru', 400, 7, 16, 256, 3),        # burst-stream backward: every dirty-step phase (tiles 0..24), T not a multiple of the burst
    ('gru', 300, 13, 8, 128, 3),        # same, H = 128 members, ragged last tile
    ('gru', 200, 5, 8, 256, 3),         # T = burst + 1: the prologue's partial burst and the final flush
    ('lstm', 400, 7, 16, 128, 3),       # burst-stream BiLSTM sweeps: every dirty-step phase, T not a multiple of the burst
    ('lstm', 40, 5, 8, 128, 3),         # T = burst + 1
    ('gru', 20, 9, 24, 64, 3),          # H = 64: two members per tile
    ('gru', 2100, 3, 8, 64, 3),         # H = 64: chunks of 2048 + 52
    ('gru', 40, 6, 32, 512, 3),         # H = 512: sixteen members per tile (round-1 backward schedule: no room for service waves)
    ('gru', 300, 5, 16, 512, 3),        # H = 512: chunks of 256 + 44
    # round 5: the fused two-layer all-gather backward (H = 256, default) at the edges of its pipeline: layer 0 runs two fused steps behind layer 1,
    # the streaming group one; T = 2 / 3 / 4 (even T: PK write-out in step pairs, odd T: fp32 rows), a ragged tile and a single-step pair
    ('gru', 5, 2, 8, 256, 3), ('gru', 20, 3, 8, 256, 3), ('gru', 33, 4, 16, 256, 3), ('gru', 17, 6, 256, 256, 3),
    # round 6: mid-size stacks whose contractions take the LDS-DMA kernels (B T = 2048: a multiple of 128, split-K weight gradients) in tilings the full-size
    # cases do not have: 16 row tiles x 3 column tiles in the projection, 8 K chunks of 256 rows; H = 512: 6 x 2 tiles of the weight gradient; the BiLSTM's
    # K = 256 projection, its 256 x 128-tile dW_hh pairs and the NN form of dX
    ('gru', 64, 32, 256, 256, 3), ('gru', 64, 32, 64, 512, 3), ('lstm', 64, 32, 1024, 128, 3),
]


@pytest.fixture(params=['f32', 'bf16x3'])
def gemm_mode(request):
    """Run a test under both precision modes of the time-parallel contractions (forced on every size)."""
    L.set_gemm_mode(0 if request.param == 'f32' else 1, 0)
    yield request.param
    L.set_gemm_mode(1, 1 << 28)


# call forms (tests/test_fullsize_gpu.py): 'full' = dy + dpooled | dh_n in, dX out; 'model' = what the training step passes
# (GRU: dpooled only, no dX -> the HASDY = false instances of the fused backward; BiLSTM: dy + dh_n, no dX).  The cluster kernels
# (impl 3: what the models run) take both forms at every shape; the generic / single-workgroup kernels have one instance each.
RNN_FORM_CASES = [c + ('full',) for c in RNN_CASES] + [c + ('model',) for c in RNN_CASES if c[5] == 3]


@pytest.mark.parametrize('cell,B,T,F,H,impl,form', RNN_FORM_CASES)
def test_rnn_stack_fwd_bwd(cell, B, T, F, H, impl, form, gemm_mode):
    rng = np.random.default_rng(B * 1000 + T * 100 + F + H + impl)
    Lyr = 2
    dirs = 1 if cell == 'gru' else 2
    P, names, prefix = make_rnn_params(rng, cell, F, H, Lyr, dirs)
    x = rng.standard_normal((B, T, F)).astype(np.float32).astype(np.float64)
    Wd = [dev(P[n]) for n in names]
    Gd = [torch.full_like(w, float('nan')) for w in Wd]
    xd = dev(x)
    pool = L.POOL_MEAN if cell == 'gru' else L.POOL_NONE
    rnn = L.Rnn(L.CELL_GRU if cell == 'gru' else L.CELL_LSTM, B, T, F, H, Lyr, dirs, True, 0.0, pool, DEV, impl=impl)
    pooled = torch.full((B, H), float('nan'), device=DEV) if cell == 'gru' else None
    h_n = torch.full((Lyr * dirs, B, H), float('nan'), device=DEV)
    rnn.forward(xd, Wd, pooled=pooled, h_n=h_n)
    y = host(rnn.layer_output())
    if cell == 'gru':
        yr, caches = R.gru_stack_fwd(x, P, prefix, Lyr)
        assert np.abs(y - yr).max() < 1e-4
        assert np.abs(host(pooled) - yr.mean(1)).max() < 1e-4
        assert np.abs(host(h_n)[-1] - yr[:, -1]).max() < 1e-4
        dpool = rng.standard_normal((B, H))
        dyv = rng.standard_normal((B, T, H)) * 0.3
        dxd = torch.full((B, T, F), float('nan'), device=DEV) if form == 'full' else None
        if form == 'full':
            rnn.backward(xd, Wd, Gd, dy=dev(dyv), dpooled=dev(dpool), dx=dxd)
            dy_full = dyv.astype(np.float32).astype(np.float64) + dpool.astype(np.float32).astype(np.float64)[:, None, :] / T
        else:
            rnn.backward(xd, Wd, Gd, dpooled=dev(dpool), dx=None)
            dy_full = np.repeat(dpool.astype(np.float32).astype(np.float64)[:, None, :] / T, T, axis=1)
        dxr, Gr = R.gru_stack_bwd(dy_full, P, prefix, Lyr, caches)
    else:
        yr, hnr, caches = R.bilstm_stack_fwd(x, P, prefix, Lyr)
        assert np.abs(y - yr).max() < 1e-4
        assert np.abs(host(h_n) - hnr).max() < 1e-4
        dyv = rng.standard_normal((B, T, 2 * H)) * 0.3
        dhn = rng.standard_normal((Lyr * 2, B, H)) * 0.3
        dxd = torch.full((B, T, F), float('nan'), device=DEV) if form == 'full' else None
        rnn.backward(xd, Wd, Gd, dy=dev(dyv), dh_n=dev(dhn), dx=dxd)
        dxr, Gr = R.bilstm_stack_bwd(dyv.astype(np.float32).astype(np.float64), dhn.astype(np.float32).astype(np.float64),
                                     P, prefix, Lyr, caches)
    rnn.check()
    if dxd is not None:
        assert relerr(host(dxd), dxr) < 1e-4, 'dx'
    for n, g in zip(names, Gd):
        assert relerr(host(g), Gr[n]) < 1e-4, n


@pytest.mark.parametrize('cell,impl,H,form,T', [
    ('gru', 2, 16, 'full', 9), ('gru', 1, 16, 'full', 9), ('lstm', 2, 16, 'full', 9), ('lstm', 1, 8, 'full', 9), ('gru', 3, 128, 'full', 9),
    ('lstm', 3, 128, 'full', 9),
    # H = 256: the fused two-layer launches (forward and backward) with the mask draw, in both call forms and at the pipeline's edges
    # (T = 2 / 3 / 4 / 6: layer 0 two fused steps behind layer 1; even T = PK step pairs, odd T = fp32 rows; model form = three input slots)
    ('gru', 3, 256, 'full', 9), ('gru', 3, 256, 'model', 9), ('gru', 3, 256, 'model', 2), ('gru', 3, 256, 'model', 3), ('gru', 3, 256, 'model', 4),
    ('gru', 3, 256, 'model', 6), ('gru', 3, 256, 'full', 6), ('gru', 3, 256, 'model', 20), ('gru', 3, 128, 'model', 9), ('lstm', 3, 128, 'model', 8)])
def test_rnn_interlayer_dropout_matches_oracle_with_same_masks(cell, impl, H, form, T, gemm_mode):
    """nn.GRU/LSTM(dropout=p) training mode: the oracle is fed the masks the HIP path drew.  form 'model' = the gradient inputs the
    training step passes (GRU: dpooled of the mean pool only; BiLSTM: dy + dh_n)."""
    rng = np.random.default_rng(77 + impl + H + T)
    B, F, Lyr, p, seed = (37 if H == 256 else 5), 10, 2, 0.5, 1234
    dirs = 1 if cell == 'gru' else 2
    P, names, prefix = make_rnn_params(rng, cell, F, H, Lyr, dirs)
    x = rng.standard_normal((B, T, F)).astype(np.float32).astype(np.float64)
    Wd = [dev(P[n]) for n in names]; Gd = [torch.zeros_like(w) for w in Wd]
    model_form = form == 'model'
    pool = L.POOL_MEAN if (model_form and cell == 'gru') else L.POOL_NONE
    rnn = L.Rnn(L.CELL_GRU if cell == 'gru' else L.CELL_LSTM, B, T, F, H, Lyr, dirs, True, p, pool, DEV, impl=impl)
    xd = dev(x)
    pooled = torch.empty(B, H, device=DEV) if pool != L.POOL_NONE else None
    rnn.forward(xd, Wd, seed=seed, pooled=pooled)
    y0 = host(rnn.layer_output(0)); y0d = host(rnn.layer_output_dropped(0))
    mask = host(L.dropout_mask(B * T * H * dirs, p, seed, 16, DEV)).reshape(B, T, H * dirs)   # site = DEP_SITE_RNN0 + 0
    assert set(np.unique(mask)).issubset({0.0, 2.0})
    assert 0.35 < (mask == 0).mean() < 0.65
    assert np.abs(y0d - y0 * mask).max() < 1e-6
    dyv = rng.standard_normal((B, T, H * dirs)).astype(np.float32).astype(np.float64)
    if cell == 'gru':
        yr, caches = R.gru_stack_fwd(x, P, prefix, Lyr, masks=[mask])
        if model_form:
            dpool = rng.standard_normal((B, H)).astype(np.float32).astype(np.float64)
            rnn.backward(xd, Wd, Gd, dpooled=dev(dpool), dx=None)
            assert np.abs(host(pooled) - yr.mean(1)).max() < 1e-4
            dyv = np.repeat(dpool[:, None, :] / T, T, axis=1)
        else:
            rnn.backward(xd, Wd, Gd, dy=dev(dyv))
        _, Gr = R.gru_stack_bwd(dyv, P, prefix, Lyr, caches, masks=[mask])
    else:
        yr, hnr, caches = R.bilstm_stack_fwd(x, P, prefix, Lyr, masks=[mask])
        dhn = np.zeros((Lyr * 2, B, H))
        if model_form:
            dhn = (rng.standard_normal((Lyr * 2, B, H)) * 0.3).astype(np.float32).astype(np.float64)
            rnn.backward(xd, Wd, Gd, dy=dev(dyv), dh_n=dev(dhn), dx=None)
        else:
            rnn.backward(xd, Wd, Gd, dy=dev(dyv))
        _, Gr = R.bilstm_stack_bwd(dyv, dhn, P, prefix, Lyr, caches, masks=[mask])
    assert np.abs(host(rnn.layer_output()) - yr).max() < 1e-4
    for n, g in zip(names, Gd):
        assert relerr(host(g), Gr[n]) < 1e-4, n
    # a different seed draws a different mask; the same seed reproduces it bit for bit
    y_a = rnn.layer_output().clone()
    rnn.forward(xd, Wd, seed=seed)
    assert torch.equal(y_a, rnn.layer_output())
    rnn.forward(xd, Wd, seed=seed + 1)
    assert not torch.equal(y_a, rnn.layer_output())


def test_rnn_generic_and_mfma_agree_on_cfg1_shape():
    rng = np.random.default_rng(3)
    B, T, F, H = 8, 50, 39, 128
    P, names, prefix = make_rnn_params(rng, 'gru', F, H, 2, 1)
    x = dev(rng.standard_normal((B, T, F)))
    Wd = [dev(P[n]) for n in names]
    outs = []
    for impl in (1, 2):
        rnn = L.Rnn(L.CELL_GRU, B, T, F, H, 2, 1, False, 0.0, L.POOL_SUM, DEV, impl=impl)
        pooled = torch.empty(B, H, device=DEV)
        rnn.forward(x, Wd, pooled=pooled)
        outs.append((host(rnn.layer_output()), host(pooled)))
    assert np.abs(outs[0][0] - outs[1][0]).max() < 2e-5
    assert np.abs(outs[0][1] - outs[1][1]).max() < 2e-4


# ----------------------------------------------------------------------------- attention
# H in {64,128,256}: attention.hip (rows cached in LDS up to T H ~ 39 K floats, re-read beyond: T = 330 at H = 128, 170 at 256);
# other widths: the first-generation kernels
@pytest.mark.parametrize('B,T,H', [(3, 6, 8), (7, 50, 128), (2, 300, 16), (5, 300, 128), (3, 330, 128), (4, 33, 64), (5, 60, 256), (6, 90, 256),
                                   (2, 170, 256), (2, 1, 128), (3, 17, 128)])
def test_attention(B, T, H):
    rng = np.random.default_rng(B + T + H)
    out = rng.standard_normal((B, T, 2 * H)).astype(np.float32).astype(np.float64)
    hn = rng.standard_normal((4, B, H)).astype(np.float32).astype(np.float64)
    Wa = (rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float32).astype(np.float64)
    ba = rng.standard_normal(H).astype(np.float32).astype(np.float64)
    dctx = rng.standard_normal((B, H)).astype(np.float32).astype(np.float64)
    ctx, saved = L.attn_fwd(dev(out), dev(hn), dev(Wa), dev(ba))
    ctxr, cache = R.attention_fwd(out, hn, Wa, ba)
    assert np.abs(host(ctx) - ctxr).max() < 1e-4
    assert np.abs(host(saved[0]) - cache[5]).max() < 1e-5
    dWa = torch.empty(H, H, device=DEV); dba = torch.empty(H, device=DEV)
    dout, dhn = L.attn_bwd(dev(dctx), dev(out), dev(Wa), saved, 4, dWa, dba)
    doutr, dhnr, dWar, dbar = R.attention_bwd(dctx, Wa, cache)
    assert relerr(host(dout), doutr) < 1e-4
    assert relerr(host(dhn), dhnr) < 1e-4
    assert relerr(host(dWa), dWar) < 1e-4
    assert relerr(host(dba), dbar) < 1e-4


# ----------------------------------------------------------------------------- heads / losses
def test_head_loss_kinds():
    rng = np.random.default_rng(9)
    B = 37
    z = rng.standard_normal((B, 2)).astype(np.float32).astype(np.float64) * 2
    y = rng.integers(0, 2, B)
    zd = dev(z); yd = torch.from_numpy(y.astype(np.int32)).to(DEV)
    out = torch.empty(B, 2, device=DEV); rows = torch.empty(B, device=DEV); dz = torch.empty(B, 2, device=DEV)
    loss = torch.zeros(1, device=DEV)
    # CE on softmax output (double softmax)
    L.head_loss(L.LOSS_CE_ON_SOFTMAX, zd, yd, out, rows, dz, B)
    L.reduce_loss(rows, B, loss)
    p = R.softmax(z); lr_, dp = R.ce_on_probs(p, y)
    assert np.abs(host(out) - p).max() < 1e-6
    assert abs(host(loss)[0] - lr_) < 1e-6
    assert relerr(host(dz), R.softmax_bwd(p, dp)) < 1e-5
    # the same with torch.long labels read in place (DEP_LOSS_LABELS_I64): bit-identical
    y64 = torch.from_numpy(y.astype(np.int64)).to(DEV)
    dz64 = torch.empty_like(dz); rows64 = torch.empty_like(rows)
    L.head_loss(L.LOSS_CE_ON_SOFTMAX | L.LOSS_LABELS_I64, zd, y64, out, rows64, dz64, B)
    assert torch.equal(dz64, dz) and torch.equal(rows64, rows)
    with pytest.raises(L.DepError):
        L.head_loss(L.LOSS_L1_RELU | L.LOSS_LABELS_I64, zd, y64, out, rows64, dz64, B)
    # CE on logits
    L.head_loss(L.LOSS_CE_LOGITS, zd, yd, out, rows, dz, B)
    L.reduce_loss(rows, B, loss)
    lr_, dzr = R.ce_logits(z, y)
    assert abs(host(loss)[0] - lr_) < 1e-6 and relerr(host(dz), dzr) < 1e-5
    # regression heads
    z1 = (rng.standard_normal((B, 1)) * 3 + 1).astype(np.float32).astype(np.float64)
    yt = rng.uniform(-1, 3, (B, 1)).astype(np.float32).astype(np.float64)
    z1d, ytd = dev(z1), dev(yt)
    out1 = torch.empty(B, 1, device=DEV); dz1 = torch.empty(B, 1, device=DEV)
    for kind, fn, relu in ((L.LOSS_L1_RELU, R.l1_loss, True), (L.LOSS_SMOOTHL1_RELU, R.smooth_l1_loss, True),
                           (L.LOSS_SMOOTHL1, R.smooth_l1_loss, False)):
        L.head_loss(kind, z1d, ytd, out1, rows, dz1, B)
        L.reduce_loss(rows, B, loss)
        o = np.maximum(z1, 0) if relu else z1
        lr_, g = fn(o, yt)
        if relu:
            g = g * (z1 > 0)
        assert np.abs(host(out1) - o).max() < 1e-6
        assert abs(host(loss)[0] - lr_) < 1e-5
        assert np.abs(host(dz1) - g).max() < 1e-6


def test_relu_dropout_colsum_adam():
    rng = np.random.default_rng(10)
    M, N = 50, 70
    z = rng.standard_normal((M, N)).astype(np.float32)
    zd = dev(z); a = torch.empty_like(zd)
    L.relu_dropout_fwd(zd, a, 0.3, 99, L.SITE_FC1)
    mask = host(L.dropout_mask(M * N, 0.3, 99, L.SITE_FC1, DEV)).reshape(M, N)
    assert np.abs(host(a) - np.maximum(z, 0) * mask).max() < 1e-6
    da = rng.standard_normal((M, N)).astype(np.float32); dz = torch.empty_like(zd)
    L.relu_dropout_bwd(dev(da), zd, dz, 0.3, 99, L.SITE_FC1)
    assert np.abs(host(dz) - da * mask * (z > 0)).max() < 1e-6
    cs = torch.empty(N, device=DEV)
    L.colsum(zd, cs)
    assert np.abs(host(cs) - z.astype(np.float64).sum(0)).max() < 1e-4
    # Adam / AdamW, 3 steps
    n = 1000
    for decoupled, wd in ((True, 1e-2), (False, 0.0), (False, 1e-3)):
        p = rng.standard_normal(n).astype(np.float32).astype(np.float64); m = np.zeros(n); v = np.zeros(n)
        pd = dev(p); md = torch.zeros(n, device=DEV); vd = torch.zeros(n, device=DEV)
        for step in range(1, 4):
            g = rng.standard_normal(n).astype(np.float32).astype(np.float64)
            L.adam_step(pd, dev(g), md, vd, 1e-3, 0.9, 0.999, 1e-8, wd, decoupled, step)
            p, m, v = R.adam_step(p, g, m, v, step, 1e-3, wd=wd, decoupled=decoupled)
        assert np.abs(host(pd) - p).max() < 1e-6


@pytest.mark.parametrize('B,Hin,H1,C', [(5, 32, 8, 2), (512, 256, 256, 2), (77, 128, 128, 1), (64, 256, 64, 0), (3, 64, 256, 16)])
@pytest.mark.parametrize('p,first', [(0.0, 0), (0.5, 1), (0.3, 0)])
def test_fused_mlp_head_against_float64_with_the_same_masks(B, Hin, H1, C, p, first):
    """csrc/head.hip: [Dropout] -> Linear -> ReLU -> Dropout -> [Linear] in one launch, its backward in two; the masks are
    the draws of the stand-alone dropout kernels (dep_dropout_mask at the same seed / site)."""
    assert L.head_mlp_supported(Hin, H1, C)
    rng = np.random.default_rng(B * 7 + Hin + C)
    x = rng.standard_normal((B, Hin)).astype(np.float32)
    W1 = (rng.standard_normal((H1, Hin)) * 0.1).astype(np.float32); b1 = rng.standard_normal(H1).astype(np.float32)
    W2 = (rng.standard_normal((max(C, 1), H1)) * 0.1).astype(np.float32); b2 = rng.standard_normal(max(C, 1)).astype(np.float32)
    seed, sites = 1234, (L.SITE_FC0, L.SITE_FC1)
    m0 = host(L.dropout_mask(B * Hin, p, seed, sites[0], DEV)).reshape(B, Hin) if (first and p > 0) else np.ones((B, Hin))
    m1 = host(L.dropout_mask(B * H1, p, seed, sites[1], DEV)).reshape(B, H1) if p > 0 else np.ones((B, H1))
    a0r = x.astype(np.float64) * m0
    z1r = a0r @ W1.T.astype(np.float64) + b1
    a1r = np.maximum(z1r, 0) * m1
    xd, W1d, b1d = dev(x), dev(W1), dev(b1)
    W2d, b2d = (dev(W2), dev(b2)) if C else (None, None)
    use0 = bool(first and p > 0)
    a0 = torch.empty_like(xd) if use0 else xd
    z1 = torch.empty(B, H1, device=DEV); a1 = torch.empty_like(z1)
    z2 = torch.empty(B, C, device=DEV) if C else None
    L.head_mlp_fwd(xd, W1d, b1d, W2d, b2d, a0 if use0 else None, z1, a1, z2, p, seed, sites, use0)
    assert np.abs(host(a0) - a0r).max() < 1e-6
    assert relerr(host(z1), z1r) < 2e-6 and relerr(host(a1), a1r) < 2e-6
    assert ((host(a1) == 0) == (a1r == 0)).all()                      # same mask, same ReLU pattern
    if not C:
        return
    z2r = a1r @ W2.T.astype(np.float64) + b2
    assert relerr(host(z2), z2r) < 2e-6
    dz2 = rng.standard_normal((B, C)).astype(np.float32)
    dW1 = torch.full((H1, Hin), 7.0, device=DEV); db1 = torch.full((H1,), 7.0, device=DEV)      # must be overwritten
    dW2 = torch.full((C, H1), 7.0, device=DEV); db2 = torch.full((C,), 7.0, device=DEV)
    dx = torch.empty_like(xd); dz1 = torch.empty_like(z1)
    L.head_mlp_bwd(dev(dz2), a0, z1, a1, W1d, W2d, dW1, db1, dW2, db2, dx, dz1, p, seed, sites, use0)
    da1 = dz2.astype(np.float64) @ W2
    dz1r = da1 * (host(z1) > 0) * m1
    assert relerr(host(dz1), dz1r) < 2e-6
    assert relerr(host(dW2), dz2.astype(np.float64).T @ a1r) < 5e-6 and relerr(host(db2), dz2.astype(np.float64).sum(0)) < 5e-6
    assert relerr(host(dW1), dz1r.T @ a0r) < 5e-6 and relerr(host(db1), dz1r.sum(0)) < 5e-6
    assert relerr(host(dx), (dz1r @ W1) * m0) < 5e-6
    # the composed launches (DEP_HEAD_FUSED=0 path of models._MLPHead) agree
    zc = L.linear_fwd(a0, W1d, b1d)
    assert relerr(host(zc), host(z1)) < 2e-6


def test_fused_mlp_head_refuses_shapes_it_does_not_cover():
    assert not L.load().dep_head_mlp_supported(300, 256, 2) and not L.load().dep_head_mlp_supported(40, 256, 2)
    assert not L.load().dep_head_mlp_supported(256, 256, 17)
    x = torch.zeros(4, 40, device=DEV); W1 = torch.zeros(8, 40, device=DEV); b1 = torch.zeros(8, device=DEV)
    z1 = torch.zeros(4, 8, device=DEV)
    with pytest.raises(L.DepError):
        L.head_mlp_fwd(x, W1, b1, None, None, None, z1, torch.empty_like(z1), None, 0.0, 0, (L.SITE_FC0, L.SITE_FC1), False)


def test_bad_arguments_fail_loudly():
    with pytest.raises(L.DepError):
        L.gemm(1, 1, 4, 4, 4, torch.zeros(16, device=DEV), 4, torch.zeros(16, device=DEV), 4, torch.zeros(16, device=DEV), 4)
    with pytest.raises(L.DepError):
        L.Rnn(L.CELL_GRU, 4, 4, 4, 4, 2, 2, False, 0.0, L.POOL_NONE, DEV)     # bidirectional GRU is not in the path


def test_backward_refuses_a_reserve_from_the_other_precision_mode():
    """The packed W_hh images in the reserve are precision-mode specific (include/dep_rnn.h): flipping the mode between a
    forward and its backward must fail loudly (DEP_ERR_ARG), not produce silently wrong gradients."""
    rng = np.random.default_rng(21)
    B, T, F, H = 8, 5, 16, 128
    P, names, prefix = make_rnn_params(rng, 'gru', F, H, 2, 1)
    Wd = [dev(P[n]) for n in names]; Gd = [torch.zeros_like(w) for w in Wd]
    x = dev(rng.standard_normal((B, T, F)))
    rnn = L.Rnn(L.CELL_GRU, B, T, F, H, 2, 1, True, 0.0, L.POOL_NONE, DEV, impl=3)
    dy = dev(rng.standard_normal((B, T, H)))
    try:
        L.set_gemm_mode(1)
        rnn.forward(x, Wd)
        L.set_gemm_mode(0)
        with pytest.raises(L.DepError, match='mode'):
            rnn.backward(x, Wd, Gd, dy=dy)
        L.set_gemm_mode(1)
        rnn.backward(x, Wd, Gd, dy=dy)          # same mode again: accepted
        rnn.check()
    finally:
        L.set_gemm_mode(1, 1 << 28)


def test_sweep_status_is_sticky_over_a_step():
    """A give-up in any sweep of a step must still be visible after the whole step: the status word is only
    cleared by dep_rnn_forward; a raised word makes the later sweeps leave at entry and dep_rnn_status report it."""
    rng = np.random.default_rng(22)
    B, T, F, H = 8, 5, 16, 128
    P, names, prefix = make_rnn_params(rng, 'gru', F, H, 2, 1)
    Wd = [dev(P[n]) for n in names]; Gd = [torch.zeros_like(w) for w in Wd]
    x = dev(rng.standard_normal((B, T, F)))
    rnn = L.Rnn(L.CELL_GRU, B, T, F, H, 2, 1, True, 0.0, L.POOL_NONE, DEV, impl=3)
    dy = dev(rng.standard_normal((B, T, H)))
    rnn.forward(x, Wd); rnn.check()
    off = L.load().dep_rnn_workspace_xbuf_offset(__import__('ctypes').byref(rnn.desc)) // 4
    rnn.workspace[off:off + 1].view(torch.int32).fill_(3)       # as if a forward sweep had given up
    rnn.backward(x, Wd, Gd, dy=dy)                              # must not clear it
    with pytest.raises(L.DepError, match='gave up'):
        rnn.check()
    rnn.forward(x, Wd)                                          # the next step starts clean
    rnn.backward(x, Wd, Gd, dy=dy)
    rnn.check()


# ----------------------------------------------------------------------------- host-loop helpers (round 4): bit-exact integer / copy work
@pytest.mark.parametrize('N,n,shape', [(40, 17, (5, 8)), (9, 9, (3, 7)), (300, 64, (30, 256)), (6, 1, (1,))])
def test_gather_rows_is_an_exact_index_select(N, n, shape):
    rng = np.random.default_rng(N + n)
    X = rng.standard_normal((N,) + shape).astype(np.float32)
    idx = rng.integers(0, N, n)
    got = L.gather_rows(dev(X), torch.as_tensor(idx, dtype=torch.int64, device=DEV))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), X[idx])
    # the feeder built on it: rows(a, b) == features[idxs][a:b], runs and scattered index lists
    from icassp2022_depression_amd import _common
    for idxs in (list(idx), list(range(2, min(N, 7)))):
        _common.invalidate_device_features()
        fd = _common.FeatureFeeder(X, idxs, DEV, role='t')
        assert np.array_equal(fd.rows(0, len(idxs)).cpu().numpy(), X[idxs])
    _common.invalidate_device_features()


def test_concat_and_argmax_count_equal_torch_bit_for_bit():
    from icassp2022_depression_amd import _common
    rng = np.random.default_rng(3)
    a = rng.standard_normal((37, 128)).astype(np.float32); b = rng.standard_normal((37, 256)).astype(np.float32)
    big = dev(np.concatenate([b, b], 1))
    cat = _common.concat_features(dev(a), big[:, 256:])              # a strided second operand
    assert np.array_equal(cat.cpu().numpy(), np.concatenate([a, b], 1))
    for B, C in ((37, 2), (512, 2), (5, 7), (1, 3)):
        p = rng.random((B, C)).astype(np.float32)
        p[B // 2] = p[B // 2, 0]                                       # a tie: the first index wins, like torch.max on the reference's CPU path
        y = rng.integers(0, C, B)
        ref_pred = torch.from_numpy(p).max(1, keepdim=True)[1].numpy()
        pd = dev(p)
        assert np.array_equal(_common.predict(pd).cpu().numpy(), ref_pred)
        for dt in (torch.int64, torch.int32):
            cnt = torch.full((), 5, dtype=torch.int64, device=DEV)
            _common.count_correct(pd, torch.as_tensor(y, dtype=dt, device=DEV), cnt)
            assert int(cnt.item()) == 5 + int((ref_pred[:, 0] == y).sum())
        out = torch.empty(B + 3, 1, dtype=torch.int64, device=DEV)
        _common.predict(pd, out=out[2:2 + B])
        assert np.array_equal(out[2:2 + B].cpu().numpy(), ref_pred)
    buf = _common.prediction_buffer(11, DEV)
    _common.store_predictions(buf, 4, dev(np.arange(3, dtype=np.float32).reshape(3, 1) + 1))
    assert np.array_equal(buf.cpu().numpy(), np.array([0, 0, 0, 0, 1, 2, 3, 0, 0, 0, 0], np.float32))
