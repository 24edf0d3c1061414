"""Bits shared by the five script-shaped modules: metrics, checkpoint I/O, mini-batch slicing,
permutation augmentation.  (Reference: the duplicated helpers at Classification/audio_gru_whole.py:123-159,
text_bilstm_whole.py:116-152, fuse_net_whole.py:30-68.)"""
import itertools
import os

import numpy as np
import torch

from . import parallel


def save(model, filename):
    """Reference `save()` pickles the whole nn.Module (audio_gru_whole.py:123-126).  The HIP-backed module is
    not picklable as a torch module, so the checkpoint holds what every consumer of those files actually reads:
    the state_dict (name -> CPU tensor) plus the config needed to rebuild the module."""
    save_filename = '{}.pt'.format(filename)
    os.makedirs(os.path.dirname(save_filename) or '.', exist_ok=True)
    payload = {'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()},
               'class': type(model).__name__, 'variant': getattr(model, 'variant', None)}
    if parallel.rank() == 0:
        torch.save(payload, save_filename)
        print('Saved as %s' % save_filename)


def load_checkpoint_state_dict(path):
    obj = torch.load(path, map_location='cpu', weights_only=False)
    if isinstance(obj, dict) and 'state_dict' in obj:
        return obj['state_dict']
    if hasattr(obj, 'state_dict'):          # a reference-made pickle of a whole torch module
        return obj.state_dict()
    return obj


def standard_confusion_matrix(y_test, y_test_pred):
    """[[TP, FP], [FN, TN]] (audio_gru_whole.py:128-146).  Like sklearn's confusion_matrix followed by the 2x2
    unpack, this raises ValueError when only one class is present in y_true and y_pred together."""
    yt = np.asarray(y_test.cpu().numpy() if torch.is_tensor(y_test) else y_test).reshape(-1).astype(np.int64)
    yp = np.asarray(y_test_pred.cpu().numpy() if torch.is_tensor(y_test_pred) else y_test_pred).reshape(-1).astype(np.int64)
    labels = np.unique(np.concatenate([yt, yp]))
    if labels.size != 2:
        raise ValueError('not enough values to unpack: confusion matrix is %dx%d' % (labels.size, labels.size))
    lo, hi = labels
    tp = int(np.sum((yt == hi) & (yp == hi))); tn = int(np.sum((yt == lo) & (yp == lo)))
    fp = int(np.sum((yt == lo) & (yp == hi))); fn = int(np.sum((yt == hi) & (yp == lo)))
    return np.array([[tp, fp], [fn, tn]])


def prf(conf_matrix):
    """accuracy / precision / recall / F1 exactly as the reference computes them (audio_gru_whole.py:223-226):
    0/0 gives nan (with numpy's RuntimeWarning), not an exception."""
    tp, fp = conf_matrix[0]; fn, tn = conf_matrix[1]
    accuracy = float(tp + tn) / np.sum(conf_matrix)
    with np.errstate(invalid='ignore', divide='ignore'):
        precision = np.float64(tp) / np.float64(tp + fp)
        recall = np.float64(tp) / np.float64(tp + fn)
        f1 = 2 * (precision * recall) / (precision + recall)
    return accuracy, float(precision), float(recall), float(f1)


def minibatches(n, batch_size):
    """(start, stop) of the reference's python-slice mini-batching; the last batch may be ragged."""
    for i in range(0, n, batch_size):
        yield i, min(n, i + batch_size)


def rank_slice(lo, hi):
    """Rows of the global mini-batch [lo, hi) this data-parallel rank works on."""
    a, b = parallel.shard_slice(hi - lo)
    return lo + a, lo + b


def permutation_augment(features, targets, idxs, is_positive, keep, label=None):
    """Append time-axis permutations of the selected samples (audio_gru_whole.py:268-299): for every idx in
    `idxs` with is_positive(idx) the permutations numbered `keep` (lexicographic order of
    itertools.permutations) are appended to features/targets and their new indices returned; other
    indices pass through.  Returns (features, targets, out_idxs)."""
    out = []
    new_f = []; new_t = []
    n0 = len(features)
    for idx in idxs:
        if is_positive(idx):
            feat = features[idx]
            for count, perm in enumerate(itertools.permutations(range(feat.shape[0]))):
                if count in keep:
                    new_f.append(feat[list(perm)])
                    new_t.append(targets[idx] if label is None else label)
                    out.append(n0 + len(new_f) - 1)
        else:
            out.append(idx)
    if new_f:
        features = np.concatenate([features, np.stack(new_f)], 0)
        targets = np.concatenate([targets, np.asarray(new_t, dtype=targets.dtype)])
    return features, targets, out
