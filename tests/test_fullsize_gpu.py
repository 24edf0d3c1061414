"""Parity at the benchmark's real sizes (BASELINE.json configs[1..4] per-GPU shapes).

The oracle cannot run B = 512, T = 300 in seconds, but utterances are independent: the HIP path runs the FULL
batch (full grid: 256 / 512 co-resident workgroups over 8 XCDs, every cluster exchanging concurrently) and the
oracle runs 16 sampled utterances spread over the first / middle / last tiles.  Backward: the incoming gradients
are non-zero on the sampled utterances only, so the full-size weight gradients must equal the oracle's gradients
over the sample, every other utterance's dX must be exactly zero (nothing leaks between utterances), and the
sampled rows' dX must match.
"""
import numpy as np
import pytest

from oracle import ref_numpy as R

torch = pytest.importorskip('torch')
pytestmark = [pytest.mark.gpu, pytest.mark.oracle]

if torch.cuda.is_available():
    from icassp2022_depression_amd import _lib as L
    DEV = torch.device('cuda:0')

# first tile, a tile in the middle of the first chunk, XCD-group boundaries, the last tile (ragged ends included)
SAMPLE = np.array([0, 1, 15, 16, 17, 130, 255, 256, 257, 300, 383, 384, 495, 496, 510, 511])


def f64(a):
    return np.asarray(a).astype(np.float32).astype(np.float64)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy().astype(np.float64)


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def make_rnn_params(rng, cell, F, H, Lyr, dirs):
    G = 3 if cell == 'gru' else 4
    P = {}; names = []
    prefix = 'lstm_net_audio' if cell == 'gru' else 'lstm_net'
    k = 1.0 / np.sqrt(H)
    for l in range(Lyr):
        for d in range(dirs):
            sfx = f'l{l}' + ('_reverse' if d else '')
            inp = F if l == 0 else H * dirs
            for nm, shp in (('weight_ih', (G * H, inp)), ('weight_hh', (G * H, H)), ('bias_ih', (G * H,)), ('bias_hh', (G * H,))):
                key = f'{prefix}.{nm}_{sfx}'
                P[key] = f64(rng.uniform(-k, k, shp))
                names.append(key)
    return P, names, prefix


@pytest.fixture(params=['f32', 'bf16x3'])
def gemm_mode(request):
    L.set_gemm_mode(0 if request.param == 'f32' else 1, 0)
    yield request.param
    L.set_gemm_mode(1, 1 << 28)


FULL_CASES = [
    # cell, B, T, F, H, dropout
    ('gru', 512, 300, 256, 256, 0.0),      # BASELINE configs[1]: the headline benchmark shape
    ('gru', 512, 300, 256, 256, 0.5),      # ... with the benchmark's inter-layer dropout (masks fed to the oracle)
    ('lstm', 512, 300, 1024, 128, 0.0),    # configs[2]: BiLSTM-128 x2 on ELMo-like F = 1024
    ('lstm', 512, 300, 1024, 128, 0.5),
    ('gru', 512, 300, 39, 256, 0.0),       # configs[4] audio leg: F = 39 (unaligned rows: scalar GEMM loaders), T = 300
    ('gru', 520, 40, 39, 128, 0.5),        # two launch chunks (512 + 8), H = 128 members, ragged last tile
]


# 'full'  : every gradient input at once (dy + dpooled | dh_n) and a dX output -- the widest operator contract;
# 'model' : exactly the call the training step makes (models.py AudioBiLSTM.backward: dpooled only, no dy, no dX;
#           TextBiLSTM.backward: dy + dh_n, no dX).  For the GRU-256 stack this is ANOTHER kernel instance
#           (gru2_bwd_fused<.., HASDY = false, ..>: three input slots, prefetch distance 2) -- the one bench.py times.
@pytest.mark.parametrize('form', ['full', 'model'])
@pytest.mark.parametrize('cell,B,T,F,H,p', FULL_CASES)
def test_full_size_stack_against_sampled_oracle(cell, B, T, F, H, p, gemm_mode, form):
    rng = np.random.default_rng(B + T + F + H + int(p * 10))
    Lyr = 2
    dirs = 1 if cell == 'gru' else 2
    P, names, prefix = make_rnn_params(rng, cell, F, H, Lyr, dirs)
    S = SAMPLE[SAMPLE < B] if B == 512 else np.array([0, 5, 16, 255, 256, 300, 511, 512, 513, 519])
    x32 = rng.standard_normal((B, T, F)).astype(np.float32)
    xs = x32[S].astype(np.float64)
    xd = torch.from_numpy(x32).to(DEV)
    Wd = [dev(P[n]) for n in names]
    Gd = [torch.full_like(w, float('nan')) for w in Wd]
    pool = L.POOL_MEAN if cell == 'gru' else L.POOL_NONE
    seed = 4242
    rnn = L.Rnn(L.CELL_GRU if cell == 'gru' else L.CELL_LSTM, B, T, F, H, Lyr, dirs, True, p, pool, DEV, impl=3)
    pooled = torch.full((B, H), float('nan'), device=DEV) if cell == 'gru' else None
    h_n = torch.full((Lyr * dirs, B, H), float('nan'), device=DEV)
    rnn.forward(xd, Wd, seed=seed, pooled=pooled, h_n=h_n)
    rnn.check()
    masks = None
    if p > 0:
        m = L.dropout_mask(B * T * H * dirs, p, seed, 16, DEV).view(B, T, H * dirs)      # site = DEP_SITE_RNN0 + 0
        masks = [host(m[torch.from_numpy(S).to(DEV)])]
        del m
    Sd = torch.from_numpy(S).to(DEV)
    y = host(rnn.layer_output()[Sd])
    assert np.isfinite(host(rnn.layer_output().sum()))                 # every utterance was written
    # gradients: non-zero on the sample only
    dy_s = f64(rng.standard_normal((len(S), T, H * dirs)) * 0.3)
    dyd = torch.zeros(B, T, H * dirs, device=DEV); dyd[Sd] = dev(dy_s)
    dxd = torch.full((B, T, F), float('nan'), device=DEV) if form == 'full' else None
    if cell == 'gru':
        yr, caches = R.gru_stack_fwd(xs, P, prefix, Lyr, masks=masks)
        assert np.abs(y - yr).max() < 1e-4
        assert np.abs(host(pooled[Sd]) - yr.mean(1)).max() < 1e-4
        assert np.abs(host(h_n[-1][Sd]) - yr[:, -1]).max() < 1e-4
        dp_s = f64(rng.standard_normal((len(S), H)))
        dpd = torch.zeros(B, H, device=DEV); dpd[Sd] = dev(dp_s)
        if form == 'full':
            rnn.backward(xd, Wd, Gd, dy=dyd, dpooled=dpd, dx=dxd)
            dxr, Gr = R.gru_stack_bwd(dy_s + dp_s[:, None, :] / T, P, prefix, Lyr, caches, masks=masks)
        else:
            rnn.backward(xd, Wd, Gd, dpooled=dpd, dx=None)
            dxr, Gr = R.gru_stack_bwd(np.repeat(dp_s[:, None, :] / T, T, axis=1), P, prefix, Lyr, caches, masks=masks)
    else:
        yr, hnr, caches = R.bilstm_stack_fwd(xs, P, prefix, Lyr, masks=masks)
        assert np.abs(y - yr).max() < 1e-4
        assert np.abs(host(h_n[:, Sd]) - hnr).max() < 1e-4
        dhn_s = f64(rng.standard_normal((Lyr * 2, len(S), H)) * 0.3)
        dhd = torch.zeros(Lyr * 2, B, H, device=DEV); dhd[:, Sd] = dev(dhn_s)
        rnn.backward(xd, Wd, Gd, dy=dyd, dh_n=dhd, dx=dxd)
        dxr, Gr = R.bilstm_stack_bwd(dy_s, dhn_s, P, prefix, Lyr, caches, masks=masks)
    rnn.check()
    if dxd is not None:
        assert relerr(host(dxd[Sd]), dxr) < 1e-4, 'dx (sampled utterances)'
        rest = torch.ones(B, dtype=torch.bool, device=DEV); rest[Sd] = False
        assert float(dxd[rest].abs().max()) == 0.0, 'gradient leaked into an utterance whose dy is zero'
    for n, g in zip(names, Gd):
        assert relerr(host(g), Gr[n]) < 1e-4, n


def _sd_from_model(model):
    return {k: host(v) for k, v in model.state_dict().items() if torch.is_tensor(v) and v.is_cuda}


def test_full_size_audio_model_forward_against_sampled_oracle():
    """The benchmark's model (LayerNorm fold + GRU-256 x2 + mean pool + head) at (512, 300, 256): outputs of the
    sampled utterances vs the oracle (the stack's full-size gradients are covered by the test above)."""
    from icassp2022_depression_amd import audio_gru_whole as m
    B, T, F, H = 512, 300, 256, 256
    cfg = dict(m.config); cfg.update(embedding_size=F, hidden_dims=H, dropout=0.0)
    model = m.AudioBiLSTM(cfg, seed=3)
    rng = np.random.default_rng(11)
    # non-trivial LayerNorm affine so that the fold is exercised
    sd = model.state_dict()
    sd['ln.weight'].copy_(dev(1.0 + 0.2 * rng.standard_normal(F))); sd['ln.bias'].copy_(dev(0.1 * rng.standard_normal(F)))
    x32 = rng.standard_normal((B, T, F)).astype(np.float32)
    model.eval()
    out = model(torch.from_numpy(x32))
    P = R.to_f64({k: v for k, v in _sd_from_model(model).items()})
    o, _ = R.audio_forward(P, x32[SAMPLE].astype(np.float64), {'rnn_layers': 2}, 'clf')
    assert np.abs(host(out.data)[SAMPLE] - o).max() < 1e-4
    assert np.isfinite(host(out.data)).all()


def test_full_size_text_model_forward_against_sampled_oracle():
    from icassp2022_depression_amd import text_bilstm_whole as m
    B, T, F, H = 512, 300, 1024, 128
    cfg = dict(m.config); cfg.update(embedding_size=F, hidden_dims=H, dropout=0.0)
    model = m.TextBiLSTM(cfg, seed=5)
    rng = np.random.default_rng(12)
    x32 = (rng.standard_normal((B, T, F)) * 0.5).astype(np.float32)
    model.eval()
    out = model(torch.from_numpy(x32))
    P = R.to_f64(_sd_from_model(model))
    o, _ = R.text_forward(P, x32[SAMPLE].astype(np.float64), {'rnn_layers': 2}, 'clf')
    assert np.abs(host(out.data)[SAMPLE] - o).max() < 1e-4
    assert np.isfinite(host(out.data)).all()


def test_full_size_fusion_step_against_sampled_oracle():
    """BASELINE configs[3] per-GPU step: frozen GRU-256 + BiLSTM-128x2 encoders on paired (512,300,256|1024) features,
    concat, linear head, MyLoss, Adam.  Encoder features of sampled pairs vs the oracle (eval mode: no dropout);
    the loss / weight gradient / Adam update on ALL 512 pairs vs the oracle fed the device features."""
    from icassp2022_depression_amd import fuse_net_whole as m, nn
    B, T, Fa, Ft, Ha, Ht = 512, 300, 256, 1024, 256, 128
    cfg = dict(m.config); cfg.update(audio_embed_size=Fa, audio_hidden_dims=Ha, text_embed_size=Ft, text_hidden_dims=Ht)
    model = m.fusion_net(Ft, Ht, cfg['rnn_layers'], 0.0, cfg['num_classes'], Ha, Fa, seed=7)     # dropout 0: features are deterministic
    rng = np.random.default_rng(13)
    xa = rng.standard_normal((B, T, Fa)).astype(np.float32)
    xt = (rng.standard_normal((B, T, Ft)) * 0.5).astype(np.float32)
    y = rng.integers(0, 2, B)
    model.eval()
    tf, af = model.pretrained_feature((torch.from_numpy(xa), torch.from_numpy(xt)))
    P = R.to_f64(_sd_from_model(model))
    tfr, afr = R.fusion_features(P, xa[SAMPLE].astype(np.float64), xt[SAMPLE].astype(np.float64),
                                 {'rnn_layers': 2}, 'clf')
    assert np.abs(host(tf)[SAMPLE] - tfr).max() < 1e-4
    # the audio feature sits behind a SUM pool over T = 300 steps: values (and rounding) scale with T, so the 1e-4 bar
    # is taken relative to the feature scale
    assert np.abs(host(af)[SAMPLE] - afr).max() < 1e-4 * max(1.0, np.abs(afr).max())
    out = model(torch.cat((tf, af), dim=1))
    W0 = P['fc_final.0.weight'].copy()
    assert np.abs(host(out.data) - R.fusion_clf_forward(W0, host(tf), host(af))).max() < 1e-5
    # one train step of the head on the full batch (the frozen encoders' features are held fixed: dropout off)
    model.train()
    opt = nn.Adam(model.parameters(), lr=cfg['learning_rate'])
    crit = m.MyLoss()
    tf2, af2 = model.pretrained_feature((torch.from_numpy(xa), torch.from_numpy(xt)))
    assert torch.equal(tf2, tf) and torch.equal(af2, af)          # p = 0: train-mode kernels reproduce the eval features
    loss = crit(tf2, af2, y, model)
    loss.backward(); opt.step()
    lr_, gW = R.fusion_clf_loss(W0, host(tf), host(af), y)
    assert abs(loss.item() - lr_) < 1e-5 * max(1.0, abs(lr_))
    g = dict(model.named_parameters())['fc_final.0.weight'].grad
    assert relerr(host(g), gW) < 1e-4
    Wn, _, _ = R.adam_step(W0, gW, np.zeros_like(W0), np.zeros_like(W0), 1, cfg['learning_rate'])
    assert np.abs(host(model.state_dict()['fc_final.0.weight']) - Wn).max() < 1e-6
