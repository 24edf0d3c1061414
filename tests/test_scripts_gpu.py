"""The script-level functions that round 1 left unexecuted, pinned to fixtures the
reference's own functions produced (tests/golden/make_golden.py --round2-only): text classifier / regressor
train() + evaluate(), regression fusion train() / evaluate() / evaluate_audio() / evaluate_text(), the fusion
permutation pairing, the H = 128 TextBiLSTM (cluster BiLSTM kernels against the reference directly), and the text /
fusion model checkers."""
import contextlib
import io
import os
import re

import numpy as np
import pytest

from conftest import load_golden

torch = pytest.importorskip('torch')
pytestmark = [pytest.mark.gpu, pytest.mark.oracle]

if torch.cuda.is_available():
    from icassp2022_depression_amd import (_common, audio_bilstm_perm, audio_gru_whole, fuse_net_whole, model_checking, nn,
                                           text_bilstm_perm, text_bilstm_whole)
    from icassp2022_depression_amd import fuse_net as fuse_net_reg

ATOL = 1e-4


def relerr(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-12)


def capture(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        r = fn(*a, **k)
    return r, buf.getvalue()


def numbers(text):
    return [float(v) for v in re.findall(r'-?\d+\.\d+(?:e-?\d+)?', text)]


def close_params(sd, ref, what=''):
    for k, v in ref.items():
        assert np.abs(sd[k].cpu().numpy() - v).max() < 5e-5 + 2e-4 * np.abs(v).max(), (what, k)


def test_text_bilstm_h128_matches_reference_fixture():
    """TextBiLSTM at (8,50,64,128): the cluster BiLSTM sweeps (H = 128) pinned to the reference's own module
    (text_bilstm_whole.py:101-114), not only to the oracle."""
    g = load_golden('text_clf_h128')
    B, T, F, H = [int(v) for v in g['shape']]
    m = text_bilstm_whole
    cfg = dict(m.config); cfg.update(embedding_size=F, hidden_dims=H, dropout=0.0)
    model = m.TextBiLSTM(cfg, seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()}, strict=True)
    model.eval()
    assert np.abs(model(g['x']).numpy() - g['out_eval']).max() < ATOL
    # the reference's intermediate tensor: the BiLSTM output sequence (through the operator API, zero-copy view)
    assert np.abs(model._saved[2].cpu().numpy() - g['lstm_out']).max() < ATOL
    model.train()
    opt = nn.AdamW(m.get_param_group(model), lr=float(g['lr']))
    crit = nn.CrossEntropyLoss()
    for s in range(1, 4):
        opt.zero_grad()
        out = model(g['x'])
        l = crit(out, g['y'])
        l.backward()
        assert abs(l.item() - g['losses'][s - 1]) < ATOL
        if s == 1:
            live = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
            assert set(live) == set(g['grads'])
            for k, gr in g['grads'].items():
                assert relerr(live[k].cpu().numpy(), gr) < 1e-3, k
        opt.step()
    close_params(model.state_dict(), g['after3'])


def test_text_clf_train_evaluate_functions():
    """text_bilstm_whole.train / evaluate (reference lines 154-235) with the module-global protocol."""
    g = load_golden('text_clf_train_eval')
    N, T, F, H = [int(v) for v in g['shape']]
    m = text_bilstm_whole
    saved = dict(m.config)
    try:
        m.config.update(embedding_size=F, hidden_dims=H, dropout=0.0, batch_size=int(g['batch_size']), learning_rate=float(g['lr']))
        m.text_features = g['feats']; m.text_targets = g['targs']
        m.model = m.TextBiLSTM(m.config, seed=0)
        m.model.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()})
        m.optimizer = nn.AdamW(m.get_param_group(m.model), lr=m.config['learning_rate'])
        m.criterion = nn.CrossEntropyLoss()
        m.max_f1 = m.max_acc = m.max_rec = m.max_prec = 2.0
        tr = g['train_idxs'].tolist(); te = g['test_idxs'].tolist()
        _, p1 = capture(m.train, 1, tr); acc1 = m.train_acc
        _, p2 = capture(m.train, 2, tr); acc2 = m.train_acc
        tl, p3 = capture(m.evaluate, m.model, te, 1, tr)
        assert [acc1, acc2] == g['train_acc'].tolist()
        assert abs(tl - float(g['eval_loss'])) < ATOL
        close_params(m.model.state_dict(), g['after'])
        m.model.eval()
        probs = m.model(g['feats'][te].astype(np.float32)).numpy()
        assert np.abs(probs - g['probs']).max() < ATOL
        assert (m.standard_confusion_matrix(torch.from_numpy(g['targs'][te]), probs.argmax(1)) == g['conf']).all()
        # what the functions print (epoch losses, accuracy / precision / recall / F1) equals the reference's printout
        ref_nums = numbers(str(g['printed'])); got = numbers(p1 + p2 + p3)
        assert len(ref_nums) == len(got) and np.allclose(ref_nums, got, atol=2e-4), (ref_nums, got)
    finally:
        m.config.clear(); m.config.update(saved)


def test_text_reg_train_evaluate_functions():
    """text_bilstm_perm.train / evaluate (Regression, reference lines 131-210)."""
    g = load_golden('text_reg_train_eval')
    N, T, F, H = [int(v) for v in g['shape']]
    m = text_bilstm_perm
    saved = dict(m.config)
    try:
        m.config.update(embedding_size=F, hidden_dims=H, dropout=0.0, batch_size=int(g['batch_size']), learning_rate=float(g['lr']))
        m.text_features = g['feats']; m.text_targets = g['targs']
        m.model = m.TextBiLSTM(m.config, seed=0)
        m.model.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()})
        m.optimizer = nn.Adam(m.model.parameters(), lr=m.config['learning_rate'])
        m.criterion = nn.SmoothL1Loss()
        m.train_dep_idxs = [0, 1, 2, 3]; m.train_non_idxs = [4, 5, 6, 7, 8, 9]
        m.test_dep_idxs = [10, 11]; m.test_non_idxs = [12, 13, 14]
        m.min_mae = -1.0; m.min_rmse = -1.0
        mae1, p1 = capture(m.train, 1)
        mae2, p2 = capture(m.train, 2)
        tl, p3 = capture(m.evaluate, 0, m.model, mae2)
        assert np.allclose([mae1, mae2], g['train_mae'], atol=2e-3)
        assert abs(tl - float(g['eval_loss'])) < 1e-3
        close_params(m.model.state_dict(), g['after'])
        ref_nums = numbers(str(g['printed'])); got = numbers(p1 + p2 + p3)
        assert len(ref_nums) == len(got) and np.allclose(ref_nums, got, atol=2e-3), (ref_nums, got)
    finally:
        m.config.clear(); m.config.update(saved)


def test_fuse_reg_train_evaluate_and_single_modality_checks():
    """Regression/fuse_net.py train / evaluate (lines 373-456) and evaluate_audio / evaluate_text (458-524)."""
    g = load_golden('fuse_reg_train_eval')
    N, T, Fa, Ft, Ha, Ht = [int(v) for v in g['dims']]
    m = fuse_net_reg
    saved = dict(m.config)
    try:
        m.config.update(audio_embed_size=Fa, text_embed_size=Ft, audio_hidden_dims=Ha, text_hidden_dims=Ht, dropout=0.0,
                        batch_size=4, learning_rate=float(g['lr']))
        model = m.fusion_net(Ft, Ht, m.config['rnn_layers'], 0.0, m.config['num_classes'], Ha, Fa, seed=0)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()}, strict=True)
        m.model = model
        m.fuse_features = [[g['xa'][i], g['xt'][i]] for i in range(N)]; m.fuse_targets = g['y']
        m.optimizer = nn.Adam(model.parameters(), lr=m.config['learning_rate'])
        m.criterion = m.MyLoss()
        m.train_dep_idxs = [0, 1, 2]; m.train_non_idxs = [3, 4, 5, 6]
        m.test_dep_idxs = [7, 8]; m.test_non_idxs = [9, 10]
        m.min_mae = -1.0; m.min_rmse = -1.0
        mae1, p1 = capture(m.train, model, 1)
        mae2, p2 = capture(m.train, model, 2)
        tl, p3 = capture(m.evaluate, model, 0, mae2)
        assert np.allclose([mae1, mae2], g['train_mae'], atol=2e-3)
        assert abs(tl - float(g['eval_loss'])) < 1e-3 * max(1.0, abs(float(g['eval_loss'])))
        assert np.abs(model.state_dict()['fc_final.0.weight'].cpu().numpy() - g['W_after']).max() < 5e-5
        ref_nums = numbers(str(g['printed'])); got = numbers(p1 + p2 + p3)
        assert len(ref_nums) == len(got) and np.allclose(ref_nums, got, rtol=1e-4, atol=2e-3), (ref_nums, got)
        # single-modality evaluators on the reference's regressors
        ca = dict(audio_bilstm_perm.config); ca.update(embedding_size=Fa, hidden_dims=Ha, dropout=0.0)
        ct = dict(text_bilstm_perm.config); ct.update(embedding_size=Ft, hidden_dims=Ht, dropout=0.0)
        am = audio_bilstm_perm.AudioBiLSTM(ca, seed=0); tm = text_bilstm_perm.TextBiLSTM(ct, seed=0)
        am.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd_audio'].items()})
        tm.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd_text'].items()})
        m.criterion = nn.L1Loss()
        r, pa = capture(m.evaluate_audio, am)
        assert r is None
        _, pt = capture(m.evaluate_text, tm)
        assert np.allclose(numbers(pa), numbers(str(g['printed_audio'])), atol=2e-3), (pa, str(g['printed_audio']))
        assert np.allclose(numbers(pt), numbers(str(g['printed_text'])), atol=2e-3), (pt, str(g['printed_text']))
    finally:
        m.config.clear(); m.config.update(saved)


def test_fusion_permutation_pairing_equals_reference_loop():
    """fuse_net_whole.augment_pairs against the reference's own augmentation statements (fuse_net_whole.py:531-564,
    extracted by AST into the fixture generator): same new indices, same appended pairs in the same order, same labels."""
    g = load_golden('fuse_augment')
    m = fuse_net_whole
    N = g['xa'].shape[0]
    old = (m.fuse_features, m.fuse_targets, m.fuse_dep_idxs, m.fuse_non_idxs)
    try:
        m.fuse_features = [[g['xa'][i], g['xt'][i]] for i in range(N)]
        m.fuse_targets = g['targets'].copy()
        m.fuse_dep_idxs = np.where(g['targets'] == 1)[0]; m.fuse_non_idxs = np.where(g['targets'] == 0)[0]
        tr, te = m.augment_pairs(g['train_idxs_tmp'], g['test_idxs_tmp'].tolist())
        assert tr == g['train_idxs'].tolist() and te == g['test_idxs'].tolist()
        assert np.array_equal(np.asarray(m.fuse_targets), g['targets_after'])
        added = m.fuse_features[N:]
        assert np.array_equal(np.stack([a[0] for a in added]), g['added_audio'])
        assert np.array_equal(np.stack([a[1] for a in added]), g['added_text'])
    finally:
        m.fuse_features, m.fuse_targets, m.fuse_dep_idxs, m.fuse_non_idxs = old


@pytest.fixture()
def dataset(tmp_path):
    rng = np.random.default_rng(3)
    N, T, Fa, Ft = 30, 3, 12, 20
    root = tmp_path
    os.makedirs(root / 'Features/AudioWhole'); os.makedirs(root / 'Features/TextWhole')
    y = (rng.random(N) < 0.4).astype(np.int64); y[:4] = [0, 1, 0, 1]
    xa = rng.standard_normal((N, T, 1, Fa)) + y[:, None, None, None] * 0.8      # (N,3,1,F): the loader squeezes axis 2
    xt = rng.standard_normal((N, T, Ft)) + y[:, None, None] * 0.8
    np.savez(root / 'Features/AudioWhole/whole_samples_clf_256.npz', xa)
    np.savez(root / 'Features/AudioWhole/whole_labels_clf_256.npz', y)
    np.savez(root / 'Features/TextWhole/whole_samples_clf_avg.npz', xt)
    np.savez(root / 'Features/TextWhole/whole_labels_clf_avg.npz', y)
    perm = rng.permutation(N)
    folds = []
    for k in range(3):
        test = perm[k * 10:(k + 1) * 10]
        tr = np.array(sorted(set(range(N)) - set(test.tolist())))
        name = f'train_idxs_0.6{k}_{k + 1}.npy'
        np.save(root / 'Features/TextWhole' / name, tr); folds.append(name)
    return dict(root=str(root), folds=tuple(folds), Fa=Fa, Ft=Ft, N=N)


def test_text_model_checker_equals_direct_evaluation(dataset):
    """check_text_classifier (TextModelChecking.py:266-395): reload three checkpoints written by save(), rebuild the folds
    with the test-side permutations, mini-batched evaluate; must equal evaluating the same weights directly."""
    m = text_bilstm_whole
    cfg = dict(m.config); cfg.update(embedding_size=dataset['Ft'], hidden_dims=16, batch_size=4, dropout=0.0)
    names, sds = [], []
    for k in range(3):
        model = m.TextBiLSTM(cfg, seed=50 + k)
        capture(m.save, model, os.path.join(dataset['root'], 'Model/ClassificationWhole/Text', f'tck_{k}'))
        names.append(f'tck_{k}.pt'); sds.append({kk: v.clone() for kk, v in model.state_dict().items()})
    (p_, r_, f_), printed = capture(model_checking.check_text_classifier, dataset['root'], dataset['folds'], tuple(names), cfg)
    assert printed.count('Confusion Matrix:') == 3
    m.load_features(dataset['root'])
    ps, rs = [], []
    for k in range(3):
        tr = np.load(os.path.join(dataset['root'], 'Features/TextWhole', dataset['folds'][k]), allow_pickle=True)
        _, te = m.fold_split(tr, train_keep=(0, 1, 2, 3, 4, 5) if k == 0 else (0, 1, 4, 5))
        feats, targs = m.text_features, m.text_targets
        model = m.TextBiLSTM(cfg, seed=0); model.load_state_dict(sds[k])
        model.eval()
        pred = model(np.ascontiguousarray(feats[te], dtype=np.float32)).data.max(1)[1].cpu().numpy()     # ONE full batch
        _, p, r, _ = _common.prf(_common.standard_confusion_matrix(targs[te], pred))
        ps.append(p); rs.append(r)
    assert np.allclose(np.mean(ps), p_, equal_nan=True) and np.allclose(np.mean(rs), r_, equal_nan=True)


def test_fusion_model_checker_equals_direct_evaluation(dataset):
    """check_fusion_classifier (FuseModelChecking.py:22-105) on fusion checkpoints built by transplanting text / audio
    checkpoints (fuse_net_whole.py:566-588), against a direct full-batch evaluation of the same weights."""
    m = fuse_net_whole
    saved = dict(m.config)
    try:
        m.config.update(audio_embed_size=dataset['Fa'], text_embed_size=dataset['Ft'], audio_hidden_dims=16, text_hidden_dims=16,
                        dropout=0.0, batch_size=4)
        tcfg = dict(text_bilstm_whole.config); tcfg.update(embedding_size=dataset['Ft'], hidden_dims=16, dropout=0.0)
        acfg = dict(audio_gru_whole.config); acfg.update(embedding_size=dataset['Fa'], hidden_dims=16, dropout=0.0)
        names, sds = [], []
        for k in range(3):
            tm = text_bilstm_whole.TextBiLSTM(tcfg, seed=60 + k); am = audio_gru_whole.AudioBiLSTM(acfg, seed=70 + k)
            fm = m.fusion_net(dataset['Ft'], 16, 2, 0.0, 2, 16, dataset['Fa'], seed=80 + k)
            m.transplant(fm, tm.state_dict(), am.state_dict())
            capture(m.save, fm, os.path.join(dataset['root'], 'Model/ClassificationWhole/Fuse', f'fck_{k}'))
            names.append(f'fck_{k}.pt'); sds.append({kk: v.clone() for kk, v in fm.state_dict().items()})
        (p_, r_, f_), printed = capture(model_checking.check_fusion_classifier, dataset['root'], dataset['folds'], tuple(names))
        m.load_features(dataset['root'])
        ps, rs = [], []
        for k in range(3):
            tr = np.load(os.path.join(dataset['root'], 'Features/TextWhole', dataset['folds'][k]), allow_pickle=True)
            te_tmp = list(set(list(m.fuse_dep_idxs) + list(m.fuse_non_idxs)) - set(tr))
            _, te = m.augment_pairs(tr, te_tmp)
            fm = m.fusion_net(dataset['Ft'], 16, 2, 0.0, 2, 16, dataset['Fa'], seed=0); fm.load_state_dict(sds[k])
            fm.eval()
            tf, af = fm.pretrained_feature([m.fuse_features[i] for i in te])
            pred = fm(torch.cat((tf, af), dim=1)).data.max(1)[1].cpu().numpy()
            _, p, r, _ = _common.prf(_common.standard_confusion_matrix(np.asarray([m.fuse_targets[i] for i in te]), pred))
            ps.append(p); rs.append(r)
        assert np.allclose(np.mean(ps), p_, equal_nan=True) and np.allclose(np.mean(rs), r_, equal_nan=True)
    finally:
        m.config.clear(); m.config.update(saved)


def test_out_of_range_label_raises_like_torch():
    cfg = dict(audio_gru_whole.config); cfg.update(embedding_size=8, hidden_dims=16, dropout=0.0)
    model = audio_gru_whole.AudioBiLSTM(cfg, seed=0)
    out = model(np.zeros((3, 2, 8), np.float32))
    with pytest.raises(IndexError):
        nn.CrossEntropyLoss()(out, np.array([0, 2, 1]))                 # class 2 of 2
    with pytest.raises(IndexError):
        nn.CrossEntropyLoss()(out, np.array([0, -100, 1]))              # torch's ignore_index is not part of the path


def test_models_built_without_a_seed_differ_and_follow_the_global_seed():
    """every fold's model must start from different weights (the reference consumes the global RNG), and
    torch.manual_seed must make the sequence reproducible."""
    cfg = dict(audio_gru_whole.config); cfg.update(embedding_size=8, hidden_dims=16)
    torch.manual_seed(123)
    a = audio_gru_whole.AudioBiLSTM(cfg).state_dict()['fc_audio.1.weight'].clone()
    b = audio_gru_whole.AudioBiLSTM(cfg).state_dict()['fc_audio.1.weight'].clone()
    torch.manual_seed(123)
    a2 = audio_gru_whole.AudioBiLSTM(cfg).state_dict()['fc_audio.1.weight'].clone()
    assert not torch.equal(a, b) and torch.equal(a, a2)


def test_streamed_feature_feeder_equals_resident_mode(monkeypatch):
    """_common.FeatureFeeder: arrays beyond DEP_FEATURES_HBM_GB are fed from a pinned fp32 host copy by a copy stream, one
    mini-batch ahead of the step that consumes it.  Forced here with a zero budget: two epochs of audio_gru_whole.train() must leave
    bit-identical parameters and the same accuracy count as the HBM-resident mode (same kernels, same inputs)."""
    g = load_golden('audio_clf_train_eval')
    N, T, F, H = [int(v) for v in g['shape']]
    m = audio_gru_whole
    saved_cfg = dict(m.config)
    saved = (m.audio_features, m.audio_targets, m.model, m.optimizer, m.criterion)
    res = {}
    try:
        for mode, budget in (('resident', '64'), ('streamed', '0')):
            monkeypatch.setenv('DEP_FEATURES_HBM_GB', budget)
            _common.invalidate_device_features()
            m.config.update(embedding_size=F, hidden_dims=H, dropout=0.0, batch_size=4, learning_rate=float(g['lr']))
            m.audio_features = g['feats']; m.audio_targets = g['targs']
            m.model = m.AudioBiLSTM(m.config, seed=0)
            m.model.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()})
            m.optimizer = nn.AdamW(m.get_param_group(m.model), lr=m.config['learning_rate']); m.criterion = nn.CrossEntropyLoss()
            idx = [5, 0, 3, 11, 7, 2, 9, 1, 14, 6, 4]                     # not a run: the resident mode gathers, the streamed one copies slices
            feed = _common.FeatureFeeder(m.audio_features, idx, m.model.device, role='audio_features')
            assert (feed.Xd is None) == (mode == 'streamed')
            capture(m.train, 1, idx); capture(m.train, 2, idx)
            res[mode] = ({k: v.cpu().numpy().copy() for k, v in m.model.state_dict().items()}, int(m.train_acc))
        assert res['resident'][1] == res['streamed'][1]
        for k, v in res['resident'][0].items():
            assert np.array_equal(v, res['streamed'][0][k]), k
    finally:
        _common.invalidate_device_features()
        m.config.clear(); m.config.update(saved_cfg)
        m.audio_features, m.audio_targets, m.model, m.optimizer, m.criterion = saved
