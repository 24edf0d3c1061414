"""CPU oracle for the GRU / BiLSTM / late-fusion hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement (float64 by default) of the arithmetic the reference
scripts delegate to PyTorch (torch.nn.GRU / LSTM / LayerNorm / Linear / Softmax / losses /
Adam[W] + autograd).  It is the *checker* for the HIP path: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import anything under oracle/.  The product package
(icassp2022-depression_amd/) never imports it and fails loudly when its HIP library is missing.

Parity pin: the reference tree holds no tests or golden vectors ("parity unpinned" by the
reference itself).  The pin is created by this repo: tests/golden/make_golden.py
imports the reference's own classes from /root/reference (AST extraction, build container only),
runs them on seeded inputs and freezes inputs -> outputs/loss/grads/post-step params as .npz
fixtures; tests/test_oracle_golden.py checks every function below against those fixtures.

All citations are relative to /root/reference/DepressionCollected/.
Gate conventions are PyTorch's (verified against the fixtures):
  GRU  rows of weight_ih/hh ordered r,z,n ; LSTM rows ordered i,f,g,o.
"""
from __future__ import annotations

import numpy as np

F64 = np.float64


# ----------------------------------------------------------------------------- elementwise
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def log_softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    s = x - m
    return s - np.log(np.exp(s).sum(axis=axis, keepdims=True))


# ----------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, w, b, eps=1e-5):
    """nn.LayerNorm(F) over the last axis (Classification/audio_gru_whole.py:62,104)."""
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mu) * rstd
    return xhat * w + b, (xhat, rstd)


def layernorm_bwd(dy, w, cache):
    xhat, rstd = cache
    dw = (dy * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    db = dy.reshape(-1, xhat.shape[-1]).sum(0)
    g = dy * w
    dx = rstd * (g - g.mean(-1, keepdims=True) - xhat * (g * xhat).mean(-1, keepdims=True))
    return dx, dw, db


# ----------------------------------------------------------------------------- GRU
def gru_layer_fwd(x, w_ih, w_hh, b_ih, b_hh):
    """One unidirectional GRU layer, batch-first x (B,T,I), h0 = 0.
    torch.nn.GRU as used at Classification/audio_gru_whole.py:59-60,105.
      r = s(gi_r+gh_r) ; z = s(gi_z+gh_z) ; n = tanh(gi_n + r*gh_n) ; h = (1-z)*n + z*h_prev
    """
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.T + b_ih                     # (B,T,3H)
    h = np.zeros((B, H), x.dtype)
    ys = np.zeros((B, T, H), x.dtype)
    R = np.zeros_like(ys); Z = np.zeros_like(ys); N = np.zeros_like(ys); HN = np.zeros_like(ys)
    for t in range(T):
        gh = h @ w_hh.T + b_hh
        r = sigmoid(gi[:, t, :H] + gh[:, :H])
        z = sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        ys[:, t] = h
        R[:, t] = r; Z[:, t] = z; N[:, t] = n; HN[:, t] = gh[:, 2 * H:]
    return ys, (x, ys, R, Z, N, HN)


def gru_layer_bwd(dy, w_ih, w_hh, cache):
    """BPTT for gru_layer_fwd.  dy: (B,T,H) gradient w.r.t. every h_t."""
    x, ys, R, Z, N, HN = cache
    B, T, H = ys.shape
    dgi = np.zeros((B, T, 3 * H), x.dtype)
    dgh = np.zeros((B, T, 3 * H), x.dtype)
    dh = np.zeros((B, H), x.dtype)
    for t in range(T - 1, -1, -1):
        hp = ys[:, t - 1] if t > 0 else np.zeros((B, H), x.dtype)
        d = dy[:, t] + dh
        r, z, n, hn = R[:, t], Z[:, t], N[:, t], HN[:, t]
        dn = d * (1.0 - z) * (1.0 - n * n)
        dz = d * (hp - n) * z * (1.0 - z)
        dr = dn * hn * r * (1.0 - r)
        dgi[:, t] = np.concatenate([dr, dz, dn], 1)
        dgh[:, t] = np.concatenate([dr, dz, dn * r], 1)
        dh = d * z + dgh[:, t] @ w_hh
    hprev = np.concatenate([np.zeros((B, 1, H), x.dtype), ys[:, :-1]], 1)
    dW_ih = np.einsum('btg,bti->gi', dgi, x)
    dW_hh = np.einsum('btg,bth->gh', dgh, hprev)
    db_ih = dgi.sum((0, 1)); db_hh = dgh.sum((0, 1))
    dx = dgi @ w_ih
    return dx, dW_ih, dW_hh, db_ih, db_hh


def gru_stack_fwd(x, P, prefix, L, masks=None):
    """L stacked GRU layers with inter-layer dropout masks (already scaled by 1/(1-p)); masks[l]
    multiplies the output of layer l (l < L-1) exactly like nn.GRU(dropout=p) in training mode."""
    caches = []
    inp = x
    for l in range(L):
        y, c = gru_layer_fwd(inp, P[f'{prefix}.weight_ih_l{l}'], P[f'{prefix}.weight_hh_l{l}'],
                             P[f'{prefix}.bias_ih_l{l}'], P[f'{prefix}.bias_hh_l{l}'])
        caches.append(c)
        inp = y
        if l < L - 1 and masks is not None and masks[l] is not None:
            inp = y * masks[l]
    return inp, caches


def gru_stack_bwd(dy, P, prefix, L, caches, masks=None):
    G = {}
    d = dy
    for l in range(L - 1, -1, -1):
        if l < L - 1 and masks is not None and masks[l] is not None:
            d = d * masks[l]
        d, dWi, dWh, dbi, dbh = gru_layer_bwd(d, P[f'{prefix}.weight_ih_l{l}'],
                                              P[f'{prefix}.weight_hh_l{l}'], caches[l])
        G[f'{prefix}.weight_ih_l{l}'] = dWi; G[f'{prefix}.weight_hh_l{l}'] = dWh
        G[f'{prefix}.bias_ih_l{l}'] = dbi; G[f'{prefix}.bias_hh_l{l}'] = dbh
    return d, G


# ----------------------------------------------------------------------------- LSTM
def lstm_dir_fwd(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one LSTM layer; x is (B,T,I) (batch-first view of the reference's
    time-first tensor, Classification/text_bilstm_whole.py:103-105), h0 = c0 = 0.
      g = x W_ih^T + b_ih + h W_hh^T + b_hh ; c = s(f)*c + s(i)*tanh(g_g) ; h = s(o)*tanh(c)"""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.T + b_ih + b_hh
    h = np.zeros((B, H), x.dtype); c = np.zeros((B, H), x.dtype)
    ys = np.zeros((B, T, H), x.dtype); cs = np.zeros((B, T, H), x.dtype)
    gates = np.zeros((B, T, 4 * H), x.dtype)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = gi[:, t] + h @ w_hh.T
        i = sigmoid(g[:, :H]); f = sigmoid(g[:, H:2 * H]); gg = np.tanh(g[:, 2 * H:3 * H]); o = sigmoid(g[:, 3 * H:])
        c = f * c + i * gg
        h = o * np.tanh(c)
        ys[:, t] = h; cs[:, t] = c
        gates[:, t] = np.concatenate([i, f, gg, o], 1)
    return ys, h, (x, ys, cs, gates, reverse)


def lstm_dir_bwd(dy, dhn, w_ih, w_hh, cache):
    """dy (B,T,H): grad of every h_t ; dhn (B,H) or None: extra grad on the final hidden state."""
    x, ys, cs, gates, reverse = cache
    B, T, H = ys.shape
    dg = np.zeros((B, T, 4 * H), x.dtype)
    dh = np.zeros((B, H), x.dtype) if dhn is None else dhn.copy()
    dc = np.zeros((B, H), x.dtype)
    order = list(range(T - 1, -1, -1)) if reverse else list(range(T))
    hprev_seq = np.zeros_like(ys)
    for k in range(T - 1, -1, -1):          # walk the sweep backwards
        t = order[k]
        tp = order[k - 1] if k > 0 else None
        hp = ys[:, tp] if tp is not None else np.zeros((B, H), x.dtype)
        cp = cs[:, tp] if tp is not None else np.zeros((B, H), x.dtype)
        hprev_seq[:, t] = hp
        i = gates[:, t, :H]; f = gates[:, t, H:2 * H]; gg = gates[:, t, 2 * H:3 * H]; o = gates[:, t, 3 * H:]
        d = dy[:, t] + dh
        tc = np.tanh(cs[:, t])
        do = d * tc * o * (1 - o)
        dct = d * o * (1 - tc * tc) + dc
        di = dct * gg * i * (1 - i)
        df = dct * cp * f * (1 - f)
        dgg = dct * i * (1 - gg * gg)
        dc = dct * f
        dg[:, t] = np.concatenate([di, df, dgg, do], 1)
        dh = dg[:, t] @ w_hh
    dW_ih = np.einsum('btg,bti->gi', dg, x)
    dW_hh = np.einsum('btg,bth->gh', dg, hprev_seq)
    db = dg.sum((0, 1))
    dx = dg @ w_ih
    return dx, dW_ih, dW_hh, db, db.copy()


def _sfx(l, d):
    return f'l{l}' + ('_reverse' if d else '')


def bilstm_stack_fwd(x, P, prefix, L, masks=None):
    """2-direction L-layer LSTM. Returns out (B,T,2H) [fwd | bwd] and h_n (2L,B,H) ordered
    [l0_fwd, l0_bwd, l1_fwd, l1_bwd] like torch (text_bilstm_whole.py:105)."""
    caches = []; hn = []
    inp = x
    for l in range(L):
        outs = []
        for d in (0, 1):
            s = _sfx(l, d)
            y, hl, c = lstm_dir_fwd(inp, P[f'{prefix}.weight_ih_{s}'], P[f'{prefix}.weight_hh_{s}'],
                                    P[f'{prefix}.bias_ih_{s}'], P[f'{prefix}.bias_hh_{s}'], bool(d))
            outs.append(y); hn.append(hl); caches.append(c)
        out = np.concatenate(outs, -1)
        inp = out
        if l < L - 1 and masks is not None and masks[l] is not None:
            inp = out * masks[l]
    return out, np.stack(hn, 0), caches


def bilstm_stack_bwd(dout, dhn, P, prefix, L, caches, masks=None):
    """dout (B,T,2H) grad of top-layer output ; dhn (2L,B,H) grad of h_n."""
    G = {}
    H = dhn.shape[-1]
    d = dout
    for l in range(L - 1, -1, -1):
        if l < L - 1 and masks is not None and masks[l] is not None:
            d = d * masks[l]
        dx_sum = 0.0
        for dd in (0, 1):
            s = _sfx(l, dd)
            dx, dWi, dWh, dbi, dbh = lstm_dir_bwd(d[..., dd * H:(dd + 1) * H], dhn[2 * l + dd],
                                                  P[f'{prefix}.weight_ih_{s}'], P[f'{prefix}.weight_hh_{s}'],
                                                  caches[2 * l + dd])
            G[f'{prefix}.weight_ih_{s}'] = dWi; G[f'{prefix}.weight_hh_{s}'] = dWh
            G[f'{prefix}.bias_ih_{s}'] = dbi; G[f'{prefix}.bias_hh_{s}'] = dbh
            dx_sum = dx_sum + dx
        d = dx_sum
    return d, G


# ----------------------------------------------------------------------------- attention
def attention_fwd(out, hn, Wa, ba):
    """attention_net_with_w (Classification/text_bilstm_whole.py:74-99).
    out (B,T,2H), hn (K,B,H) -> ctx (B,H); also returns alpha (B,T)."""
    H = out.shape[-1] // 2
    h = out[..., :H] + out[..., H:]
    hs = hn.sum(0)                                  # torch.sum(lstm_hidden, dim=1) on (B,K,H)
    pre = hs @ Wa.T + ba
    q = np.maximum(pre, 0.0)
    m = np.tanh(h)
    sc = np.einsum('bj,btj->bt', q, m)
    al = softmax(sc, -1)
    ctx = np.einsum('bt,btj->bj', al, h)
    return ctx, (h, hs, pre, q, m, al, hn.shape[0])


def attention_bwd(dctx, Wa, cache):
    h, hs, pre, q, m, al, K = cache
    dal = np.einsum('bj,btj->bt', dctx, h)
    dh = al[..., None] * dctx[:, None, :]
    dsc = al * (dal - (al * dal).sum(-1, keepdims=True))
    dq = np.einsum('bt,btj->bj', dsc, m)
    dh = dh + dsc[..., None] * q[:, None, :] * (1 - m * m)
    dpre = dq * (pre > 0)
    dWa = dpre.T @ hs; dba = dpre.sum(0)
    dhs = dpre @ Wa
    dout = np.concatenate([dh, dh], -1)
    dhn = np.broadcast_to(dhs, (K,) + dhs.shape).copy()
    return dout, dhn, dWa, dba


# ----------------------------------------------------------------------------- losses
def ce_on_probs(p, y):
    """nn.CrossEntropyLoss applied to *softmax outputs* (audio_gru_whole.py:72,188,308): a second
    log_softmax is taken over the probabilities.  Returns loss, dL/dp."""
    B = p.shape[0]
    ls = log_softmax(p, -1)
    loss = -ls[np.arange(B), y].mean()
    d = softmax(p, -1)
    d[np.arange(B), y] -= 1.0
    return loss, d / B


def ce_logits(z, y):
    B = z.shape[0]
    ls = log_softmax(z, -1)
    loss = -ls[np.arange(B), y].mean()
    d = softmax(z, -1)
    d[np.arange(B), y] -= 1.0
    return loss, d / B


def softmax_bwd(p, dp):
    return p * (dp - (dp * p).sum(-1, keepdims=True))


def l1_loss(o, y):
    """nn.L1Loss (Regression/audio_bilstm_perm.py:251), mean reduction."""
    d = o - y
    return np.abs(d).mean(), np.sign(d) / d.size


def smooth_l1_loss(o, y):
    """nn.SmoothL1Loss beta=1 (Regression/text_bilstm_perm.py:247)."""
    d = o - y
    a = np.abs(d)
    loss = np.where(a < 1.0, 0.5 * d * d, a - 0.5).mean()
    g = np.where(a < 1.0, d, np.sign(d)) / d.size
    return loss, g


# ----------------------------------------------------------------------------- heads
def mlp_head_fwd(x, W1, b1, W2, b2, m0=None, m1=None):
    """Dropout -> Linear -> ReLU -> Dropout -> Linear (audio_gru_whole.py:65-70). m0/m1 are
    pre-scaled dropout masks or None."""
    a0 = x if m0 is None else x * m0
    z1 = a0 @ W1.T + b1
    a1 = np.maximum(z1, 0.0)
    a1d = a1 if m1 is None else a1 * m1
    z2 = a1d @ W2.T + b2
    return z2, (a0, z1, a1d, m0, m1)


def mlp_head_bwd(dz2, W1, W2, cache):
    a0, z1, a1d, m0, m1 = cache
    dW2 = dz2.T @ a1d; db2 = dz2.sum(0)
    da1 = dz2 @ W2
    if m1 is not None:
        da1 = da1 * m1
    dz1 = da1 * (z1 > 0)
    dW1 = dz1.T @ a0; db1 = dz1.sum(0)
    dx = dz1 @ W1
    if m0 is not None:
        dx = dx * m0
    return dx, dW1, db1, dW2, db2


# ----------------------------------------------------------------------------- whole models
def audio_forward(P, x, cfg, variant, masks=None):
    """variant 'clf': Classification/audio_gru_whole.py:103-108 (LN, mean pool, Softmax)
       variant 'reg': Regression/audio_bilstm_perm.py:122-127 (no LN, sum pool, ReLU).
    masks: dict with optional 'rnn' (list per layer), 'fc0', 'fc1' pre-scaled dropout masks."""
    masks = masks or {}
    L = cfg['rnn_layers']
    cache = {}
    if variant == 'clf':
        xin, cache['ln'] = layernorm_fwd(x, P['ln.weight'], P['ln.bias'])
    else:
        xin = x
    y, cache['rnn'] = gru_stack_fwd(xin, P, 'lstm_net_audio', L, masks.get('rnn'))
    T = x.shape[1]
    pooled = y.mean(1) if variant == 'clf' else y.sum(1)
    z, cache['head'] = mlp_head_fwd(pooled, P['fc_audio.1.weight'], P['fc_audio.1.bias'],
                                    P['fc_audio.4.weight'], P['fc_audio.4.bias'],
                                    masks.get('fc0'), masks.get('fc1'))
    out = softmax(z, -1) if variant == 'clf' else np.maximum(z, 0.0)
    cache.update(z=z, out=out, T=T, variant=variant, masks=masks, L=L)
    return out, cache


def audio_backward(P, dout, cache):
    variant = cache['variant']; L = cache['L']; masks = cache['masks']
    if variant == 'clf':
        dz = softmax_bwd(cache['out'], dout)
    else:
        dz = dout * (cache['z'] > 0)
    dpool, dW1, db1, dW2, db2 = mlp_head_bwd(dz, P['fc_audio.1.weight'], P['fc_audio.4.weight'], cache['head'])
    T = cache['T']
    dy = np.repeat(dpool[:, None, :], T, 1) * ((1.0 / T) if variant == 'clf' else 1.0)
    dx, G = gru_stack_bwd(dy, P, 'lstm_net_audio', L, cache['rnn'], masks.get('rnn'))
    G['fc_audio.1.weight'] = dW1; G['fc_audio.1.bias'] = db1
    G['fc_audio.4.weight'] = dW2; G['fc_audio.4.bias'] = db2
    if variant == 'clf':
        dx, dlw, dlb = layernorm_bwd(dx, P['ln.weight'], cache['ln'])
        G['ln.weight'] = dlw; G['ln.bias'] = dlb
    return dx, G


def text_forward(P, x, cfg, variant, masks=None, fc_idx=None):
    """variant 'clf': Classification/text_bilstm_whole.py:101-114 (fc_out.0 / fc_out.3, Softmax)
       variant 'reg': Regression/text_bilstm_perm.py:112-124 (Dropout first: fc_out.1 / fc_out.4, ReLU)."""
    masks = masks or {}
    L = cfg['rnn_layers']
    i1, i2 = fc_idx if fc_idx is not None else ((0, 3) if variant == 'clf' else (1, 4))
    out, hn, rc = bilstm_stack_fwd(x, P, 'lstm_net', L, masks.get('rnn'))
    ctx, ac = attention_fwd(out, hn, P['attention_layer.0.weight'], P['attention_layer.0.bias'])
    z, hc = mlp_head_fwd(ctx, P[f'fc_out.{i1}.weight'], P[f'fc_out.{i1}.bias'],
                         P[f'fc_out.{i2}.weight'], P[f'fc_out.{i2}.bias'],
                         masks.get('fc0'), masks.get('fc1'))
    o = softmax(z, -1) if variant == 'clf' else np.maximum(z, 0.0)
    cache = dict(rnn=rc, att=ac, head=hc, z=z, out=o, variant=variant, masks=masks, L=L, idx=(i1, i2),
                 alpha=ac[5], ctx=ctx)
    return o, cache


def text_backward(P, dout, cache):
    variant = cache['variant']; L = cache['L']; masks = cache['masks']; i1, i2 = cache['idx']
    dz = softmax_bwd(cache['out'], dout) if variant == 'clf' else dout * (cache['z'] > 0)
    dctx, dW1, db1, dW2, db2 = mlp_head_bwd(dz, P[f'fc_out.{i1}.weight'], P[f'fc_out.{i2}.weight'], cache['head'])
    dseq, dhn, dWa, dba = attention_bwd(dctx, P['attention_layer.0.weight'], cache['att'])
    dx, G = bilstm_stack_bwd(dseq, dhn, P, 'lstm_net', L, cache['rnn'], masks.get('rnn'))
    G[f'fc_out.{i1}.weight'] = dW1; G[f'fc_out.{i1}.bias'] = db1
    G[f'fc_out.{i2}.weight'] = dW2; G[f'fc_out.{i2}.bias'] = db2
    G['attention_layer.0.weight'] = dWa; G['attention_layer.0.bias'] = dba
    return dx, G


def fusion_features(P, x_audio, x_text, cfg, variant='clf', masks=None):
    """fusion_net.pretrained_feature (Classification/fuse_net_whole.py:336-366 ; Regression/fuse_net.py:313-343):
    text encoder -> Dropout,Linear,ReLU,Dropout (fc_out.1) ; audio: [LN only for clf] -> GRU -> SUM over T ->
    Dropout,Linear,ReLU,Dropout (fc_audio.1)."""
    masks = masks or {}
    L = cfg['rnn_layers']
    out, hn, _ = bilstm_stack_fwd(x_text, P, 'lstm_net', L, masks.get('rnn_text'))
    ctx, _ = attention_fwd(out, hn, P['attention_layer.0.weight'], P['attention_layer.0.bias'])
    a0 = ctx if masks.get('t0') is None else ctx * masks['t0']
    tf = np.maximum(a0 @ P['fc_out.1.weight'].T + P['fc_out.1.bias'], 0.0)
    if masks.get('t1') is not None:
        tf = tf * masks['t1']
    xa = x_audio
    if variant == 'clf':
        xa, _ = layernorm_fwd(x_audio, P['ln.weight'], P['ln.bias'])
    y, _ = gru_stack_fwd(xa, P, 'lstm_net_audio', L, masks.get('rnn_audio'))
    pooled = y.sum(1)
    a0 = pooled if masks.get('a0') is None else pooled * masks['a0']
    af = np.maximum(a0 @ P['fc_audio.1.weight'].T + P['fc_audio.1.bias'], 0.0)
    if masks.get('a1') is not None:
        af = af * masks['a1']
    return tf, af


def fusion_clf_forward(W, tf, af):
    """fusion_net.forward (fuse_net_whole.py:368-374): Softmax(cat(text,audio) W^T)."""
    return softmax(np.concatenate([tf, af], 1) @ W.T, -1)


def fusion_clf_loss(W, tf, af, y):
    """MyLoss (fuse_net_whole.py:376-395): CE(text W[:, :Ht]^T) + CE(audio W[:, Ht:]^T); grad to W only."""
    Ht = tf.shape[1]
    l1, d1 = ce_logits(tf @ W[:, :Ht].T, y)
    l2, d2 = ce_logits(af @ W[:, Ht:].T, y)
    dW = np.concatenate([d1.T @ tf, d2.T @ af], 1)
    return l1 + l2, dW


def fusion_reg_forward(W, M, tf, af):
    """Regression/fuse_net.py:345-351: ReLU((sigmoid(x M^T) * x) W^T)."""
    x = np.concatenate([tf, af], 1)
    return np.maximum((sigmoid(x @ M.T) * x) @ W.T, 0.0)


def fusion_reg_loss(W, tf, af, y):
    """Regression/fuse_net.py:353-366: SmoothL1(text W[:, :Ht]^T, y) + SmoothL1(audio W[:, Ht:]^T, y)."""
    Ht = tf.shape[1]
    y = y.reshape(-1, 1)
    l1, d1 = smooth_l1_loss(tf @ W[:, :Ht].T, y)
    l2, d2 = smooth_l1_loss(af @ W[:, Ht:].T, y)
    dW = np.concatenate([d1.T @ tf, d2.T @ af], 1)
    return l1 + l2, dW


# ----------------------------------------------------------------------------- optimizer
def adam_step(p, g, m, v, step, lr, wd=0.0, decoupled=False, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam / AdamW single-tensor update (audio_gru_whole.py:307 AdamW;
    audio_bilstm_perm.py:250 Adam).  `step` is the 1-based step count.  Returns new (p, m, v)."""
    if decoupled:
        p = p * (1.0 - lr * wd)
    elif wd != 0.0:
        g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def to_f64(P):
    return {k: np.asarray(v, dtype=F64) for k, v in P.items()}
