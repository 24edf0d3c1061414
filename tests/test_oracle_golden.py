"""Pin the numpy oracle (oracle/ref_numpy.py) against fixtures captured from the REFERENCE's own
classes (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import ref_numpy as R

CFG = {'rnn_layers': 2}
# reference is fp32 (CPU torch); oracle is fp64: agreement is limited by the reference's own rounding
ATOL_OUT = 2e-5
RTOL_GRAD = 2e-4


def _check_grads(G, ref, scale_tol=RTOL_GRAD):
    assert set(ref.keys()) == set(G.keys()), (sorted(set(ref) ^ set(G)))
    for k, g in ref.items():
        tol = scale_tol * max(np.abs(g).max(), 1e-6) + 1e-7
        assert np.abs(G[k] - g).max() <= tol, (k, np.abs(G[k] - g).max(), tol)


def _adam_all(P, G, nodecay, lr, decoupled, steps_state, step):
    for k in G:
        m, v = steps_state.setdefault(k, (np.zeros_like(P[k]), np.zeros_like(P[k])))
        wd = 0.0
        if decoupled:
            wd = 0.0 if k in nodecay else 1e-5
        P[k], m, v = R.adam_step(P[k], G[k], m, v, step, lr, wd=wd, decoupled=decoupled)
        steps_state[k] = (m, v)


@pytest.mark.parametrize('name,variant', [('audio_clf_tiny', 'clf'), ('audio_clf_mid', 'clf'),
                                          ('audio_clf_cfg1', 'clf'), ('audio_reg_tiny', 'reg'),
                                          ('audio_reg_mid', 'reg')])
def test_audio_model(name, variant):
    g = load_golden(name)
    P = R.to_f64(g['sd']); x = g['x'].astype(np.float64); y = g['y']
    out, cache = R.audio_forward(P, x, CFG, variant)
    assert np.abs(out - g['out_eval']).max() < ATOL_OUT
    # intermediate: top-layer GRU output
    xin = R.layernorm_fwd(x, P['ln.weight'], P['ln.bias'])[0] if variant == 'clf' else x
    yseq, _ = R.gru_stack_fwd(xin, P, 'lstm_net_audio', 2)
    assert np.abs(yseq - g['gru_out']).max() < ATOL_OUT
    lr = float(g['lr']); state = {}
    nodecay = set(g.get('nodecay_names', np.array([])).tolist())
    steps = 3
    for s in range(1, steps + 1):
        out, cache = R.audio_forward(P, x, CFG, variant)
        if variant == 'clf':
            loss, dout = R.ce_on_probs(out, y)
        else:
            loss, dout = R.l1_loss(out, y.reshape(-1, 1).astype(np.float64))
        assert abs(loss - g['losses'][s - 1]) < 5e-5 * max(1.0, abs(loss)), (s, loss, g['losses'][s - 1])
        dx, G = R.audio_backward(P, dout, cache)
        if s == 1:
            _check_grads(G, g['grads'])
            assert np.abs(dx - g['dx']).max() <= RTOL_GRAD * np.abs(g['dx']).max() + 1e-7
            # dead parameters of the reference get no gradient
            assert not any(k.startswith('attention_layer') or k.startswith('bn') for k in g['grads'])
        _adam_all(P, G, nodecay, lr, variant == 'clf', state, s)
        if s == 1 and 'after1' in g:
            for k, v in g['after1'].items():
                assert np.abs(P[k] - v).max() < 3e-6 + 1e-5 * np.abs(v).max(), k
    for k, v in g['after3'].items():
        if k in P:
            # L1 gradients are sign() of a residual: not sensitive; Adam's first steps are ~lr*sign(g)
            assert np.abs(P[k] - v).max() < 2e-5 + 1e-4 * np.abs(v).max(), (k, np.abs(P[k] - v).max())


@pytest.mark.parametrize('name,variant', [('text_clf_tiny', 'clf'), ('text_clf_mid', 'clf'),
                                          ('text_reg_tiny', 'reg'), ('text_reg_mid', 'reg')])
def test_text_model(name, variant):
    g = load_golden(name)
    P = R.to_f64(g['sd']); x = g['x'].astype(np.float64); y = g['y']
    out, cache = R.text_forward(P, x, CFG, variant)
    assert np.abs(out - g['out_eval']).max() < ATOL_OUT
    lo, hn, _ = R.bilstm_stack_fwd(x, P, 'lstm_net', 2)
    assert np.abs(lo - g['lstm_out']).max() < ATOL_OUT
    assert np.abs(hn - g['h_n']).max() < ATOL_OUT
    assert np.abs(cache['ctx'] - g['ctx']).max() < ATOL_OUT
    lr = float(g['lr']); state = {}
    nodecay = set(g.get('nodecay_names', np.array([])).tolist())
    for s in range(1, 4):
        out, cache = R.text_forward(P, x, CFG, variant)
        if variant == 'clf':
            loss, dout = R.ce_on_probs(out, y)
        else:
            loss, dout = R.smooth_l1_loss(out, y.reshape(-1, 1).astype(np.float64))
        assert abs(loss - g['losses'][s - 1]) < 5e-5 * max(1.0, abs(loss))
        dx, G = R.text_backward(P, dout, cache)
        if s == 1:
            _check_grads(G, g['grads'])
            assert np.abs(dx - g['dx']).max() <= RTOL_GRAD * np.abs(g['dx']).max() + 1e-7
            assert not any(k.startswith('ln1') or k.startswith('ln2') for k in g['grads'])
        _adam_all(P, G, nodecay, lr, variant == 'clf', state, s)
        if s == 1:
            for k, v in g['after1'].items():
                assert np.abs(P[k] - v).max() < 3e-6 + 1e-5 * np.abs(v).max(), k
    for k, v in g['after3'].items():
        if k in P:
            assert np.abs(P[k] - v).max() < 2e-5 + 1e-4 * np.abs(v).max(), (k, np.abs(P[k] - v).max())


@pytest.mark.parametrize('name,variant', [('fuse_clf', 'clf'), ('fuse_reg', 'reg')])
def test_fusion(name, variant):
    g = load_golden(name)
    P = R.to_f64(g['sd'])
    xa = g['xa'].astype(np.float64); xt = g['xt'].astype(np.float64); y = g['y']
    tf, af = R.fusion_features(P, xa, xt, CFG, variant)
    assert np.abs(tf - g['text_feature']).max() < ATOL_OUT
    assert np.abs(af - g['audio_feature']).max() < ATOL_OUT * max(1.0, np.abs(af).max())
    W = P['fc_final.0.weight']
    if variant == 'clf':
        out = R.fusion_clf_forward(W, tf, af)
    else:
        out = R.fusion_reg_forward(W, P['modal_attn.weight'], tf, af)
    assert np.abs(out - g['out']).max() < ATOL_OUT * max(1.0, np.abs(out).max())
    m = np.zeros_like(W); v = np.zeros_like(W)
    for s in range(1, 4):
        if variant == 'clf':
            loss, dW = R.fusion_clf_loss(W, tf, af, y)
        else:
            loss, dW = R.fusion_reg_loss(W, tf, af, y.astype(np.float64))
        assert abs(loss - g['losses'][s - 1]) < 5e-5 * max(1.0, abs(loss))
        if s == 1:
            assert np.abs(dW - g['gW']).max() <= RTOL_GRAD * np.abs(g['gW']).max() + 1e-7
        W, m, v = R.adam_step(W, dW, m, v, s, float(g['lr']))
    assert np.abs(W - g['W3']).max() < 2e-5


def test_attention_alpha_sums_to_one():
    g = load_golden('text_clf_tiny')
    P = R.to_f64(g['sd'])
    _, cache = R.text_forward(P, g['x'].astype(np.float64), CFG, 'clf')
    assert np.allclose(cache['alpha'].sum(-1), 1.0)


def test_numeric_gradient_gru_lstm():
    """Independent check of the hand-written BPTT: central differences on a tiny problem."""
    rng = np.random.default_rng(0)
    B, T, F, H = 2, 4, 3, 3
    x = rng.standard_normal((B, T, F))
    for kind, G in (('gru', 3), ('lstm', 4)):
        w_ih = rng.standard_normal((G * H, F)) * 0.5; w_hh = rng.standard_normal((G * H, H)) * 0.5
        b_ih = rng.standard_normal(G * H) * 0.1; b_hh = rng.standard_normal(G * H) * 0.1
        wsel = rng.standard_normal((B, T, H))

        def f(w_hh_):
            if kind == 'gru':
                y, _ = R.gru_layer_fwd(x, w_ih, w_hh_, b_ih, b_hh)
            else:
                y, _, _ = R.lstm_dir_fwd(x, w_ih, w_hh_, b_ih, b_hh, True)
            return (y * wsel).sum()
        if kind == 'gru':
            _, c = R.gru_layer_fwd(x, w_ih, w_hh, b_ih, b_hh)
            _, _, dWh, _, _ = R.gru_layer_bwd(wsel, w_ih, w_hh, c)
        else:
            _, _, c = R.lstm_dir_fwd(x, w_ih, w_hh, b_ih, b_hh, True)
            _, _, dWh, _, _ = R.lstm_dir_bwd(wsel, None, w_ih, w_hh, c)
        num = np.zeros_like(w_hh)
        for i in range(w_hh.shape[0]):
            for j in range(w_hh.shape[1]):
                e = np.zeros_like(w_hh); e[i, j] = 1e-6
                num[i, j] = (f(w_hh + e) - f(w_hh - e)) / 2e-6
        assert np.abs(num - dWh).max() < 1e-6, kind
