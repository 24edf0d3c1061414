"""Post-hoc evaluators: the reference's `*ModelChecking.py` scripts reload saved checkpoints, rebuild the
folds (train- and test-side permutation augmentation) and print precision / recall / F1 per fold and averaged
(Classification/AudioModelChecking.py:127-208, TextModelChecking.py:266-395, FuseModelChecking.py:22-105) or MAE / RMSE
(Regression/AudioModelChecking.py:129-208).  Same flow, same prints, same index lists here on the HIP-backed modules -- pinned
to fixtures the reference's own `evaluate` functions and fold-loop statements produced (tests/golden/checker_*.npz).

Checkpoints may be this package's `{state_dict, ...}` files or reference-made pickles of a whole torch module.

Reference quirks that are results-visible and therefore kept:
  * every checker ALSO appends the training-side permutations (6 per depressed training volunteer), although it evaluates only
    the test rows: the appended rows decide the row numbers of the test-side permutations;
  * TextModelChecking.py keeps `resample_idxs` in a module global that the first fold's test loop leaves at [0, 1, 4, 5]
    (line 340): from the second fold on only 4 permutations per depressed TRAINING volunteer are appended;
  * evaluation is mini-batched (config['batch_size']) in the three classification checkers, full-batch in the regression one.
"""
import os

import numpy as np
import torch

from . import _common, audio_bilstm_perm, audio_gru_whole, fuse_net_whole, text_bilstm_whole


def _load_into(model, path, strict=True):
    sd = _common.load_checkpoint_state_dict(path)
    model.load_state_dict(sd, strict=strict)
    return model


def _report(conf_matrix):
    """The metric block every classification checker prints after model_performance (AudioModelChecking.py:150-160)."""
    print('Calculating additional test metrics...')
    accuracy, precision, recall, f1_score = _common.prf(conf_matrix)
    print("Accuracy: {}".format(accuracy)); print("Precision: {}".format(precision))
    print("Recall: {}".format(recall)); print("F1-Score: {}\n".format(f1_score)); print('=' * 89)
    return precision, recall, f1_score


def evaluate_classifier(model, features, targets, test_idxs, batch_size):
    """Mini-batched evaluation -> (precision, recall, f1): AudioModelChecking.evaluate (127-161) / TextModelChecking.evaluate
    (266-306)."""
    model.eval()
    X_test = features[test_idxs]; Y_test = targets[test_idxs]
    preds = []
    for lo, hi in _common.minibatches(X_test.shape[0], batch_size):
        out = model(np.ascontiguousarray(X_test[lo:hi], dtype=np.float32))
        preds.append(_common.predict(out).cpu().numpy())
    pred = np.concatenate(preds) if preds else np.zeros((0, 1), np.int64)
    conf_matrix = _common.standard_confusion_matrix(Y_test, pred)
    print("Confusion Matrix:"); print(conf_matrix)
    return _report(conf_matrix)


def evaluate_fusion(model, fuse_features, fuse_targets, test_idxs, batch_size):
    """FuseModelChecking.evaluate (22-60): mini-batched pretrained_feature -> concat(text, audio) -> head -> argmax."""
    model.eval()
    X = [fuse_features[i] for i in test_idxs]; Y = [fuse_targets[i] for i in test_idxs]
    preds = []
    for lo, hi in _common.minibatches(len(X), batch_size):
        tf, af = model.pretrained_feature(X[lo:hi])
        preds.append(_common.predict(model(_common.concat_features(tf, af))).cpu().numpy())
    conf_matrix = _common.standard_confusion_matrix(np.asarray(Y), np.concatenate(preds) if preds else np.zeros((0, 1), np.int64))
    print("Confusion Matrix:"); print(conf_matrix)
    return _report(conf_matrix)


def _mean_report(ps, rs, fs):
    print('precison: {} \n recall: {} \n f1 score: {}'.format(np.mean(ps), np.mean(rs), np.mean(fs)))
    return float(np.mean(ps)), float(np.mean(rs)), float(np.mean(fs))


def check_audio_folds(models, folds, batch_size):
    """Fold loop of Classification/AudioModelChecking.py:170-208 on the features already held by audio_gru_whole: `models[k]`
    is evaluated on fold k's test rows.  Returns ([p], [r], [f], [test_idxs])."""
    m = audio_gru_whole
    ps, rs, fs, tests = [], [], [], []
    for k, train_idxs_tmp in enumerate(folds):
        _, test_idxs = m.fold_split(train_idxs_tmp)
        p, r, f = evaluate_classifier(models[k], m.audio_features, m.audio_targets, test_idxs, batch_size)
        ps.append(p); rs.append(r); fs.append(f); tests.append(test_idxs)
    return ps, rs, fs, tests


def check_text_folds(models, folds, batch_size):
    """Fold loop of Classification/TextModelChecking.py:316-394 (with its leaking `resample_idxs`, see the module docstring)."""
    m = text_bilstm_whole
    ps, rs, fs, tests = [], [], [], []
    for k, train_idxs_tmp in enumerate(folds):
        _, test_idxs = m.fold_split(train_idxs_tmp, train_keep=(0, 1, 2, 3, 4, 5) if k == 0 else (0, 1, 4, 5))
        p, r, f = evaluate_classifier(models[k], m.text_features, m.text_targets, test_idxs, batch_size)
        ps.append(p); rs.append(r); fs.append(f); tests.append(test_idxs)
    return ps, rs, fs, tests


def check_fusion_folds(models, folds, batch_size):
    """Fold loop of Classification/FuseModelChecking.py:63-104 on fuse_net_whole's pair list."""
    m = fuse_net_whole
    ps, rs, fs, tests = [], [], [], []
    for k, train_idxs_tmp in enumerate(folds):
        te_tmp = list(set(list(m.fuse_dep_idxs) + list(m.fuse_non_idxs)) - set(train_idxs_tmp))
        _, test_idxs = m.augment_pairs(train_idxs_tmp, te_tmp)
        p, r, f = evaluate_fusion(models[k], m.fuse_features, m.fuse_targets, test_idxs, batch_size)
        ps.append(p); rs.append(r); fs.append(f); tests.append(test_idxs)
    return ps, rs, fs, tests


def check_audio_classifier(root, idxs_paths=('train_idxs_0.63_1.npy', 'train_idxs_0.65_2.npy', 'train_idxs_0.60_3.npy'),
                           model_paths=('BiLSTM_gru_vlad256_256_0.67_1.pt', 'BiLSTM_gru_vlad256_256_0.67_2.pt',
                                        'BiLSTM_gru_vlad256_256_0.63_3.pt'), config=None):
    m = audio_gru_whole
    m.load_features(root)
    cfg = dict(m.config if config is None else config)
    folds = [np.load(os.path.join(m.prefix, 'Features/TextWhole', f), allow_pickle=True) for f in idxs_paths]
    models = [_load_into(m.AudioBiLSTM(cfg), os.path.join(m.prefix, 'Model/ClassificationWhole/Audio', f)) for f in model_paths]
    ps, rs, fs, _ = check_audio_folds(models, folds, cfg['batch_size'])
    return _mean_report(ps, rs, fs)


def check_text_classifier(root, idxs_paths=('train_idxs_0.63_1.npy', 'train_idxs_0.60_2.npy', 'train_idxs_0.60_3.npy'),
                          model_paths=('BiLSTM_128_0.64_1.pt', 'BiLSTM_128_0.66_2.pt', 'BiLSTM_128_0.66_3.pt'), config=None):
    m = text_bilstm_whole
    m.load_features(root)
    cfg = dict(m.config if config is None else config)
    folds = [np.load(os.path.join(m.prefix, 'Features/TextWhole', f), allow_pickle=True) for f in idxs_paths]
    models = [_load_into(m.TextBiLSTM(cfg), os.path.join(m.prefix, 'Model/ClassificationWhole/Text', f)) for f in model_paths]
    ps, rs, fs, _ = check_text_folds(models, folds, cfg['batch_size'])
    return _mean_report(ps, rs, fs)


def check_fusion_classifier(root, idxs_paths=('train_idxs_0.63_1.npy', 'train_idxs_0.65_2.npy', 'train_idxs_0.60_3.npy'),
                            model_paths=('fuse_0.69_1.pt', 'fuse_0.68_2.pt', 'fuse_0.62_3.pt')):
    """FuseModelChecking.py: mini-batched fusion evaluate over the three saved fusion checkpoints."""
    m = fuse_net_whole
    m.load_features(root)
    folds = [np.load(os.path.join(m.prefix, 'Features/TextWhole', f), allow_pickle=True) for f in idxs_paths]
    c = m.config
    models = [_load_into(m.fusion_net(c['text_embed_size'], c['text_hidden_dims'], c['rnn_layers'], c['dropout'], c['num_classes'],
                                      c['audio_hidden_dims'], c['audio_embed_size']),
                         os.path.join(m.prefix, 'Model/ClassificationWhole/Fuse', f)) for f in model_paths]
    ps, rs, fs, _ = check_fusion_folds(models, folds, c['batch_size'])
    return _mean_report(ps, rs, fs)


def check_audio_regressor(root, model_path, fold=2, config=None):
    """Regression/AudioModelChecking.py:157-208: strict load of one regression checkpoint, the fold's split (training-side
    permutations appended as the script does, unused by the evaluation), full-batch MAE / RMSE on the fold's 10 depressed + 44
    other test volunteers (`fold = 2` is what the script hard-codes)."""
    m = audio_bilstm_perm
    m.load_features(root)
    cfg = dict(m.config if config is None else config)
    model = _load_into(m.AudioBiLSTM(cfg), os.path.join(m.prefix, model_path), strict=True)
    m.fold_split(fold)
    idx = list(m.test_dep_idxs) + list(m.test_non_idxs)
    model.eval()
    pred = model(np.ascontiguousarray(m.audio_features[idx], dtype=np.float32)).data.flatten().cpu().numpy()
    y = np.asarray(m.audio_targets[idx], np.float64)
    mae = float(np.mean(np.abs(y - pred))); rmse = float(np.sqrt(np.mean((y - pred) ** 2)))
    print('MAE: {:.4f}\t RMSE: {:.4f}\n'.format(mae, rmse))
    print('=' * 89)
    return mae, rmse
