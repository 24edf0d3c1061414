"""Post-hoc evaluators: the reference's `*ModelChecking.py` scripts reload saved checkpoints,
rebuild the folds (with the test-side permutation augmentation) and print precision / recall / F1 averaged over the
three folds (Classification/AudioModelChecking.py:127-208, TextModelChecking.py:266-395, FuseModelChecking.py:22-105)
or MAE / RMSE (Regression/AudioModelChecking.py:129-208).  Same flow here on the HIP-backed modules; checkpoints may
be this package's `{state_dict, ...}` files or reference-made pickles of a whole torch module."""
import os

import numpy as np
import torch

from . import _common, audio_bilstm_perm, audio_gru_whole, fuse_net_whole, text_bilstm_whole


def _load_into(model, path, strict=True):
    sd = _common.load_checkpoint_state_dict(path)
    model.load_state_dict(sd, strict=strict)
    return model


def evaluate_classifier(model, features, targets, test_idxs, batch_size):
    """Mini-batched evaluation -> (precision, recall, f1) as AudioModelChecking.evaluate (lines 127-161)."""
    model.eval()
    X_test = features[test_idxs]; Y_test = targets[test_idxs]
    preds = []
    for lo, hi in _common.minibatches(X_test.shape[0], batch_size):
        out = model(np.ascontiguousarray(X_test[lo:hi], dtype=np.float32))
        preds.append(out.data.max(1, keepdim=True)[1].cpu())
    pred = torch.cat(preds).numpy()
    conf_matrix = _common.standard_confusion_matrix(Y_test, pred)
    print("Confusion Matrix:"); print(conf_matrix)
    print('Calculating additional test metrics...')
    accuracy, precision, recall, f1_score = _common.prf(conf_matrix)
    print("Accuracy: {}".format(accuracy)); print("Precision: {}".format(precision))
    print("Recall: {}".format(recall)); print("F1-Score: {}\n".format(f1_score)); print('=' * 89)
    return precision, recall, f1_score


def _folds_clf(features, targets, dep_idxs, non_idxs, train_idxs_tmp):
    dep = set(np.asarray(dep_idxs).tolist())
    test_idxs_tmp = list(set(list(dep_idxs) + list(non_idxs)) - set(train_idxs_tmp))
    features, targets, _ = _common.permutation_augment(features, targets, train_idxs_tmp, lambda i: i in dep,
                                                       (0, 1, 2, 3, 4, 5), label=1)
    features, targets, test_idxs = _common.permutation_augment(features, targets, test_idxs_tmp, lambda i: i in dep,
                                                               (0, 1, 4, 5), label=1)
    return features, targets, test_idxs


def check_audio_classifier(root, idxs_paths=('train_idxs_0.63_1.npy', 'train_idxs_0.65_2.npy', 'train_idxs_0.60_3.npy'),
                           model_paths=('BiLSTM_gru_vlad256_256_0.67_1.pt', 'BiLSTM_gru_vlad256_256_0.67_2.pt',
                                        'BiLSTM_gru_vlad256_256_0.63_3.pt'), config=None):
    m = audio_gru_whole
    m.load_features(root)
    cfg = dict(m.config if config is None else config)
    feats, targs = m.audio_features, m.audio_targets
    ps, rs, fs = [], [], []
    for fold in range(3):
        tr = np.load(os.path.join(m.prefix, 'Features/TextWhole', idxs_paths[fold]), allow_pickle=True)
        feats, targs, test_idxs = _folds_clf(feats, targs, m.audio_dep_idxs_tmp, m.audio_non_idxs, tr)
        model = _load_into(m.AudioBiLSTM(cfg), os.path.join(m.prefix, 'Model/ClassificationWhole/Audio', model_paths[fold]))
        p, r, f = evaluate_classifier(model, feats, targs, test_idxs, cfg['batch_size'])
        ps.append(p); rs.append(r); fs.append(f)
    print('precison: {} \n recall: {} \n f1 score: {}'.format(np.mean(ps), np.mean(rs), np.mean(fs)))
    return float(np.mean(ps)), float(np.mean(rs)), float(np.mean(fs))


def check_text_classifier(root, idxs_paths=('train_idxs_0.63_1.npy', 'train_idxs_0.65_2.npy', 'train_idxs_0.60_3.npy'),
                          model_paths=('BiLSTM_128_0.64_1.pt', 'BiLSTM_128_0.66_2.pt', 'BiLSTM_128_0.62_3.pt'), config=None):
    m = text_bilstm_whole
    m.load_features(root)
    cfg = dict(m.config if config is None else config)
    feats, targs = m.text_features, m.text_targets
    ps, rs, fs = [], [], []
    for fold in range(3):
        tr = np.load(os.path.join(m.prefix, 'Features/TextWhole', idxs_paths[fold]), allow_pickle=True)
        feats, targs, test_idxs = _folds_clf(feats, targs, m.text_dep_idxs_tmp, m.text_non_idxs, tr)
        model = _load_into(m.TextBiLSTM(cfg), os.path.join(m.prefix, 'Model/ClassificationWhole/Text', model_paths[fold]))
        p, r, f = evaluate_classifier(model, feats, targs, test_idxs, cfg['batch_size'])
        ps.append(p); rs.append(r); fs.append(f)
    print('precison: {} \n recall: {} \n f1 score: {}'.format(np.mean(ps), np.mean(rs), np.mean(fs)))
    return float(np.mean(ps)), float(np.mean(rs)), float(np.mean(fs))


def check_fusion_classifier(root, idxs_paths=('train_idxs_0.63_1.npy', 'train_idxs_0.65_2.npy', 'train_idxs_0.60_3.npy'),
                            model_paths=('fuse_0.69_1.pt', 'fuse_0.68_2.pt', 'fuse_0.62_3.pt')):
    """FuseModelChecking.py: mini-batched fusion evaluate over the three saved fusion checkpoints."""
    m = fuse_net_whole
    m.load_features(root)
    m.build()
    ps, rs, fs = [], [], []
    for fold in range(3):
        tr = np.load(os.path.join(m.prefix, 'Features/TextWhole', idxs_paths[fold]), allow_pickle=True)
        te_tmp = list(set(list(m.fuse_dep_idxs) + list(m.fuse_non_idxs)) - set(tr))
        _, test_idxs = m.augment_pairs([], te_tmp)
        _load_into(m.model, os.path.join(m.prefix, 'Model/ClassificationWhole/Fuse', model_paths[fold]))
        m.model.eval()
        preds = []
        X = [m.fuse_features[i] for i in test_idxs]; Y = [m.fuse_targets[i] for i in test_idxs]
        for lo, hi in _common.minibatches(len(X), m.config['batch_size']):
            tf, af = m.model.pretrained_feature(X[lo:hi])
            preds.append(m.model(torch.cat((tf, af), dim=1)).data.max(1, keepdim=True)[1].cpu())
        conf = _common.standard_confusion_matrix(np.asarray(Y), torch.cat(preds).numpy())
        _, p, r, f = _common.prf(conf)
        print(conf); print('precision {} recall {} f1 {}'.format(p, r, f))
        ps.append(p); rs.append(r); fs.append(f)
    print('precison: {} \n recall: {} \n f1 score: {}'.format(np.mean(ps), np.mean(rs), np.mean(fs)))
    return float(np.mean(ps)), float(np.mean(rs)), float(np.mean(fs))


def check_audio_regressor(root, model_path, fold=0, config=None):
    """Regression/AudioModelChecking.py:129-208: strict load of one regression checkpoint, full-batch MAE / RMSE on
    the fold's 10 depressed + 44 non-depressed test volunteers."""
    m = audio_bilstm_perm
    m.load_features(root)
    cfg = dict(m.config if config is None else config)
    model = _load_into(m.AudioBiLSTM(cfg), os.path.join(m.prefix, model_path), strict=True)
    idx = list(m.dep_idxs[fold * 10:(fold + 1) * 10]) + list(m.non_idxs[fold * 44:(fold + 1) * 44])
    model.eval()
    pred = model(np.ascontiguousarray(m.audio_features[idx], dtype=np.float32)).data.flatten().cpu().numpy()
    y = np.asarray(m.audio_targets[idx], np.float64)
    mae = float(np.mean(np.abs(y - pred))); rmse = float(np.sqrt(np.mean((y - pred) ** 2)))
    print('MAE: {:.4f}\t RMSE: {:.4f}\n'.format(mae, rmse))
    return mae, rmse
