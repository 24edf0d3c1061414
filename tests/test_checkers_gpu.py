"""the post-hoc checkers and the threshold-gated checkpoint against what the REFERENCE's own functions
produced (tests/golden/checker_*.npz, save_gate.npz; generator: tests/golden/make_golden.py::round3 -- the `evaluate`
functions of Classification/{Audio,Text,Fuse}ModelChecking.py and Regression/AudioModelChecking.py driven by their own fold-loop
statements on reference modules with seeded weights).  Every model in those fixtures was made decisive (no sample within 5e-3 of a
tie), so the HIP path has to reproduce each argmax: confusion matrices, P / R / F1, row numbers of the evaluated permutations
and the printed text must be identical."""
import contextlib
import io
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from icassp2022_depression_amd import (_common, audio_bilstm_perm, audio_gru_whole, fuse_net_whole, model_checking, nn,
                                           text_bilstm_whole)


def _npz(name):
    return np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))


def _sd(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def capture(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        r = fn(*a, **k)
    return r, buf.getvalue()


def test_audio_checker_equals_reference_evaluate():
    z = _npz('checker_audio_clf')
    N, T, F, H = [int(v) for v in z['shape']]
    m = audio_gru_whole
    saved = (m.audio_features, m.audio_targets, m.audio_dep_idxs_tmp, m.audio_non_idxs)
    try:
        targs = z['targs']
        m.audio_features, m.audio_targets = z['feats'].copy(), targs.copy()
        m.audio_dep_idxs_tmp, m.audio_non_idxs = np.where(targs == 1)[0], np.where(targs == 0)[0]
        cfg = dict(m.config); cfg.update(embedding_size=F, hidden_dims=H, dropout=0.0)
        models = []
        for k in range(3):
            mod = m.AudioBiLSTM(cfg, seed=k); mod.load_state_dict(_sd(z, f'sd{k}/')); models.append(mod)
        (ps, rs, fs, tests), printed = capture(model_checking.check_audio_folds, models, [z[f'fold{k}'] for k in range(3)], 4)
        for k in range(3):
            assert list(tests[k]) == z[f'test{k}'].tolist()
        assert np.allclose(ps, z['ps'], atol=1e-12) and np.allclose(rs, z['rs'], atol=1e-12) and np.allclose(fs, z['fs'], atol=1e-12)
        assert printed == str(z['printed'])
        assert len(m.audio_features) == int(z['n_after']) and np.array_equal(m.audio_targets, z['targs_after'])
    finally:
        m.audio_features, m.audio_targets, m.audio_dep_idxs_tmp, m.audio_non_idxs = saved


def test_text_checker_equals_reference_evaluate_including_the_resample_leak():
    z = _npz('checker_text_clf')
    N, T, F, H = [int(v) for v in z['shape']]
    m = text_bilstm_whole
    saved = (m.text_features, m.text_targets, m.text_dep_idxs_tmp, m.text_non_idxs)
    try:
        targs = z['targs']
        m.text_features, m.text_targets = z['feats'].copy(), targs.copy()
        m.text_dep_idxs_tmp, m.text_non_idxs = np.where(targs == 1)[0], np.where(targs == 0)[0]
        cfg = dict(m.config); cfg.update(embedding_size=F, hidden_dims=H, dropout=0.0)
        models = []
        for k in range(3):
            mod = m.TextBiLSTM(cfg, seed=k); mod.load_state_dict(_sd(z, f'sd{k}/')); models.append(mod)
        (ps, rs, fs, tests), printed = capture(model_checking.check_text_folds, models, [z[f'fold{k}'] for k in range(3)],
                                               int(z['batch_size']))
        for k in range(3):
            assert list(tests[k]) == z[f'test{k}'].tolist()              # row numbers depend on the 6-then-4 training-side appends
        assert len(m.text_features) == int(z['n_after'])
        assert np.allclose(ps, z['ps'], atol=1e-12) and np.allclose(rs, z['rs'], atol=1e-12) and np.allclose(fs, z['fs'], atol=1e-12)
        assert printed == str(z['printed'])
    finally:
        m.text_features, m.text_targets, m.text_dep_idxs_tmp, m.text_non_idxs = saved


def test_fusion_checker_equals_reference_evaluate():
    z = _npz('checker_fuse_clf')
    N, T, Fa, Ft, Ha, Ht = [int(v) for v in z['dims']]
    m = fuse_net_whole
    saved = (m.fuse_features, m.fuse_targets, m.fuse_dep_idxs, m.fuse_non_idxs)
    try:
        y = z['targs']
        m.fuse_features = [[z['xa'][i], z['xt'][i]] for i in range(N)]; m.fuse_targets = y.copy()
        m.fuse_dep_idxs, m.fuse_non_idxs = np.where(y == 1)[0], np.where(y == 0)[0]
        models = []
        for k in range(3):
            fm = m.fusion_net(Ft, Ht, 2, 0.0, 2, Ha, Fa, seed=k); fm.load_state_dict(_sd(z, f'sd{k}/')); models.append(fm)
        (ps, rs, fs, tests), printed = capture(model_checking.check_fusion_folds, models, [z[f'fold{k}'] for k in range(3)], 4)
        for k in range(3):
            assert list(tests[k]) == z[f'test{k}'].tolist()
        assert len(m.fuse_features) == int(z['n_after'])
        assert np.allclose(ps, z['ps'], atol=1e-12) and np.allclose(rs, z['rs'], atol=1e-12) and np.allclose(fs, z['fs'], atol=1e-12)
        assert printed == str(z['printed'])
    finally:
        m.fuse_features, m.fuse_targets, m.fuse_dep_idxs, m.fuse_non_idxs = saved


def test_regression_checker_from_disk_equals_reference(tmp_path):
    """check_audio_regressor end to end: .npz / .npy loaders, a checkpoint in the reference's own format (a pickled module,
    written by export_reference_checkpoint), strict load, fold split, full-batch MAE / RMSE -- the reference printed the same line."""
    z = _npz('checker_audio_reg')
    Nr, T, F, H = [int(v) for v in z['shape']]
    root = tmp_path
    os.makedirs(root / 'Features/AudioWhole'); os.makedirs(root / 'Model/Regression/Audio3')
    np.savez(root / 'Features/AudioWhole/whole_samples_reg_256.npz', z['feats'][:, :, None, :])      # the loader squeezes axis 2
    np.savez(root / 'Features/AudioWhole/whole_labels_reg_256.npz', z['targs'])
    np.save(root / 'Features/AudioWhole/dep_idxs.npy', np.arange(0, 36)); np.save(root / 'Features/AudioWhole/non_idxs.npy', np.arange(36, 170))
    m = audio_bilstm_perm
    cfg = dict(m.config); cfg.update(embedding_size=F, hidden_dims=H, dropout=0.0)
    src = m.AudioBiLSTM(cfg, seed=0); src.load_state_dict(_sd(z, 'sd/'))
    _common.export_reference_checkpoint(src, str(root / 'Model/Regression/Audio3/gru_vlad256_256_8.25'))
    (mae, rmse), printed = capture(model_checking.check_audio_regressor, str(root), 'Model/Regression/Audio3/gru_vlad256_256_8.25.pt',
                                   int(z['fold']), cfg)
    assert printed == str(z['printed'])
    idx = z['test_idx']
    assert list(m.test_dep_idxs) + list(m.test_non_idxs) == idx.tolist()
    assert list(m.train_dep_idxs) == z['train_dep_idxs'].tolist() and len(m.audio_features) == int(z['n_after'])
    y = z['targs'][idx]
    assert abs(mae - np.mean(np.abs(y - z['pred']))) < 1e-3 and rmse >= mae


def test_classification_save_gate_fires_like_the_reference(tmp_path):
    """audio_gru_whole.evaluate (reference 204-245): closed while train_acc <= 0.9 len(train_idxs), open afterwards -- checkpoint
    name, the train_idxs_*.npy side file and every printed line as the reference produced them."""
    z = _npz('save_gate')
    n, T, F, H = [int(v) for v in z['clf/shape']]
    m = audio_gru_whole
    saved_cfg = dict(m.config)
    saved = (m.audio_features, m.audio_targets, m.prefix, m.model, m.optimizer, m.criterion, m.max_f1, m.max_acc, m.max_rec, m.max_prec, m.train_acc)
    try:
        m.config.update(embedding_size=F, hidden_dims=H, dropout=0.0)
        m.prefix = str(tmp_path); os.makedirs(tmp_path / 'Features/TextWhole')
        m.audio_features, m.audio_targets = z['clf/feats'], z['clf/targs']
        m.model = m.AudioBiLSTM(m.config, seed=0); m.model.load_state_dict(_sd(z, 'clf/sd/'))
        m.optimizer = nn.AdamW(m.get_param_group(m.model), lr=1e-3); m.criterion = nn.CrossEntropyLoss()
        m.max_f1 = m.max_acc = m.max_rec = m.max_prec = -1
        test_idxs = list(range(n)); train_idxs = list(range(10)); tmp_idx = z['clf/train_idxs_tmp']
        m.train_acc = 9
        _, p1 = capture(m.evaluate, m.model, test_idxs, 2, tmp_idx, train_idxs)
        assert not os.path.exists(tmp_path / 'Model') and os.listdir(tmp_path / 'Features/TextWhole') == []       # gate closed
        m.train_acc = 10
        tl, p2 = capture(m.evaluate, m.model, test_idxs, 2, tmp_idx, train_idxs)
        names = [str(s) for s in z['clf/save_names']]
        assert os.path.exists(os.path.join(str(tmp_path), names[0] + '.pt'))
        assert sorted(os.listdir(tmp_path / 'Features/TextWhole')) == [str(s) for s in z['clf/idx_files']]
        assert np.array_equal(np.load(tmp_path / 'Features/TextWhole' / str(z['clf/idx_files'][0]), allow_pickle=True), z['clf/saved_idx'])
        assert m.max_f1 == float(z['clf/max_f1']) and m.max_acc == float(z['clf/max_acc'])
        assert abs(tl - float(z['clf/eval_loss'])) < 1e-4
        ours = (p1 + p2).replace('Saved as %s\n' % os.path.join(str(tmp_path), names[0] + '.pt'), '')       # the recorder that replaced the reference's save() printed nothing
        assert ours == str(z['clf/printed'])
        # the checkpoint loads back into a fresh module, strictly
        back = m.AudioBiLSTM(m.config, seed=5)
        back.load_state_dict(_common.load_checkpoint_state_dict(os.path.join(str(tmp_path), names[0] + '.pt')), strict=True)
        assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(back.state_dict().values(), m.model.state_dict().values()))
    finally:
        m.config.clear(); m.config.update(saved_cfg)
        (m.audio_features, m.audio_targets, m.prefix, m.model, m.optimizer, m.criterion, m.max_f1, m.max_acc, m.max_rec, m.max_prec, m.train_acc) = saved


def test_regression_save_gate_fires_like_the_reference(tmp_path):
    """audio_bilstm_perm.evaluate (reference 175-213): mae <= min_mae and mae < 8.5 and train_mae < 13."""
    z = _npz('save_gate')
    m = audio_bilstm_perm
    F, H = int(z['reg/feats'].shape[2]), 16
    saved_cfg = dict(m.config)
    saved = (m.audio_features, m.audio_targets, m.prefix, m.model, m.optimizer, m.criterion, m.min_mae, m.min_rmse, m.test_dep_idxs, m.test_non_idxs)
    try:
        m.config.update(embedding_size=F, hidden_dims=H, dropout=0.0)
        m.prefix = str(tmp_path)
        m.audio_features, m.audio_targets = z['reg/feats'], z['reg/targs']
        m.model = m.AudioBiLSTM(m.config, seed=0); m.model.load_state_dict(_sd(z, 'reg/sd/'))
        m.optimizer = nn.Adam(m.model.parameters(), lr=1e-3); m.criterion = nn.L1Loss()
        m.test_dep_idxs = [0, 1, 2, 3]; m.test_non_idxs = [4, 5, 6, 7, 8, 9, 10, 11]
        m.min_mae = 100; m.min_rmse = 100
        _, p1 = capture(m.evaluate, 1, m.model, 13.0)
        assert not os.path.exists(tmp_path / 'Model')
        _, p2 = capture(m.evaluate, 1, m.model, 12.5)
        name = str(z['reg/save_names'][0])
        assert os.path.exists(os.path.join(str(tmp_path), name + '.pt'))
        assert abs(m.min_mae - float(z['reg/min_mae'])) < 1e-4 and abs(m.min_rmse - float(z['reg/min_rmse'])) < 1e-4
        ours = (p1 + p2).replace('Saved as %s\n' % os.path.join(str(tmp_path), name + '.pt'), '')
        ref = str(z['reg/printed'])
        # 'model saved: mae: {}' prints the un-rounded float: compare that line numerically, the rest verbatim
        strip = lambda s: '\n'.join(l for l in s.split('\n') if not l.startswith('model saved'))
        assert strip(ours) == strip(ref)
    finally:
        m.config.clear(); m.config.update(saved_cfg)
        (m.audio_features, m.audio_targets, m.prefix, m.model, m.optimizer, m.criterion, m.min_mae, m.min_rmse, m.test_dep_idxs, m.test_non_idxs) = saved
