"""End-to-end on a synthetic on-disk dataset shaped like EATD-Corpus features: the .npz/.npy loaders, the 3-fold
drivers with permutation augmentation, checkpoint save/load, and the model-checking evaluators."""
import contextlib
import io
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from icassp2022_depression_amd import (_common, audio_bilstm_perm, audio_gru_whole, model_checking,
                                           text_bilstm_whole)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.fixture()
def dataset(tmp_path):
    rng = np.random.default_rng(0)
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
    Nr = 170                                                  # 3 folds x (10 depressed + 44 others) test volunteers
    np.savez(root / 'Features/AudioWhole/whole_samples_reg_256.npz', rng.standard_normal((Nr, T, 1, Fa)))
    np.savez(root / 'Features/AudioWhole/whole_labels_reg_256.npz', rng.uniform(30, 70, Nr))
    perm = rng.permutation(N)
    folds = []
    for k in range(3):
        test = perm[k * 10:(k + 1) * 10]
        tr = np.array(sorted(set(range(N)) - set(test.tolist())))
        name = f'train_idxs_0.6{k}_{k + 1}.npy'
        np.save(root / 'Features/TextWhole' / name, tr); folds.append(name)
    np.save(root / 'Features/AudioWhole/dep_idxs.npy', np.arange(0, 36)); np.save(root / 'Features/AudioWhole/non_idxs.npy', np.arange(36, 170))
    return dict(root=str(root), folds=tuple(folds), Fa=Fa, Ft=Ft, N=N)


def test_audio_classifier_fold_driver_and_checker(dataset):
    m = audio_gru_whole
    saved = dict(m.config)
    try:
        m.config.update(embedding_size=dataset['Fa'], hidden_dims=16, batch_size=8, learning_rate=5e-3, dropout=0.0, epochs=4)
        m.load_features(dataset['root'])
        assert m.audio_features.shape == (dataset['N'], 3, dataset['Fa'])
        quiet(m.main, fold_files=dataset['folds'])           # 3 folds x 3 epochs, augmentation included
        assert m.audio_features.shape[0] > dataset['N']        # depressed volunteers were permuted in
        assert 0 <= m.train_acc
        # checkpoints for the checker: save the last model as every fold's checkpoint
        paths = []
        for k in range(3):
            p = os.path.join(dataset['root'], 'Model/ClassificationWhole/Audio', f'ck_{k}')
            quiet(m.save, m.model, p); paths.append(f'ck_{k}.pt')
        ref_sd = {k: v.clone() for k, v in m.model.state_dict().items()}
        p_, r_, f_ = quiet(model_checking.check_audio_classifier, dataset['root'], dataset['folds'], tuple(paths), dict(m.config))
        # direct evaluation of the same weights on the same rebuilt folds must agree
        m.load_features(dataset['root'])
        model = m.AudioBiLSTM(m.config); model.load_state_dict(ref_sd)
        ps = []
        for k in range(3):
            tr = np.load(os.path.join(dataset['root'], 'Features/TextWhole', dataset['folds'][k]), allow_pickle=True)
            _, te = m.fold_split(tr)
            ps.append(quiet(model_checking.evaluate_classifier, model, m.audio_features, m.audio_targets, te, m.config['batch_size'])[0])
        assert np.allclose(np.nanmean(ps), p_, equal_nan=True)
    finally:
        m.config.clear(); m.config.update(saved)


def test_text_classifier_learns_on_separable_data(dataset):
    m = text_bilstm_whole
    saved = dict(m.config)
    try:
        m.config.update(embedding_size=dataset['Ft'], hidden_dims=16, batch_size=8, learning_rate=1e-2, dropout=0.0, epochs=2)
        m.load_features(dataset['root'])
        from icassp2022_depression_amd import nn
        m.model = m.TextBiLSTM(m.config, seed=1)
        m.optimizer = nn.AdamW(m.get_param_group(m.model), lr=m.config['learning_rate'])
        m.criterion = nn.CrossEntropyLoss()
        m.max_f1 = m.max_acc = m.max_rec = m.max_prec = 2.0
        idx = list(range(dataset['N']))
        m.model.eval()
        l0 = m.criterion(m.model(m.text_features[idx].astype(np.float32)), m.text_targets[idx]).item()
        for ep in range(1, 30):
            quiet(m.train, ep, idx)
        m.model.eval()
        l1 = m.criterion(m.model(m.text_features[idx].astype(np.float32)), m.text_targets[idx]).item()
        assert l1 < l0 - 0.02, (l0, l1)                         # CE-on-softmax saturates slowly; it must go down
        assert m.train_acc >= dataset['N'] * 0.6
    finally:
        m.config.clear(); m.config.update(saved)


def test_regression_driver_and_checker(dataset):
    m = audio_bilstm_perm
    saved = dict(m.config)
    try:
        m.config.update(embedding_size=dataset['Fa'], hidden_dims=16, batch_size=4, learning_rate=1e-2, dropout=0.0, epochs=2)
        m.load_features(dataset['root'])
        quiet(m.main, epochs=2)
        p = os.path.join(dataset['root'], 'Model/Regression/Audio1/ck')
        quiet(m.save, m.model, p)
        mae, rmse = quiet(model_checking.check_audio_regressor, dataset['root'], 'Model/Regression/Audio1/ck.pt', 0, dict(m.config))
        assert np.isfinite(mae) and rmse >= mae
    finally:
        m.config.clear(); m.config.update(saved)
