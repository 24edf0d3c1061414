"""Drop-in for Classification/audio_gru_whole.py on MI355X: same `config`, `AudioBiLSTM(config)`,
`train(epoch, train_idxs)`, `evaluate(model, test_idxs, fold, train_idxs_tmp, train_idxs)`,
`get_param_group`, `save`, `standard_confusion_matrix`, `model_performance` and the same module-global
protocol (`model`, `optimizer`, `criterion`, `audio_features`, `audio_targets`, `train_acc`, `max_f1` ...),
with the arithmetic in libdep_rnn.so.  Unlike the reference nothing is loaded or trained at import:
call `load_features(prefix)` (the .npz/.npy loader, reference lines 18-22) and `main()` (lines 257-319).
"""
import os

import numpy as np
import torch

from . import _common, models, nn, parallel
from ._common import save, standard_confusion_matrix  # noqa: F401  (part of the module surface)

prefix = os.path.abspath(os.path.join(os.getcwd(), "."))
audio_features = None
audio_targets = None
audio_dep_idxs_tmp = None
audio_non_idxs = None

config = {
    'num_classes': 2,
    'dropout': 0.5,
    'rnn_layers': 2,
    'embedding_size': 256,
    'batch_size': 8,
    'epochs': 170,
    'learning_rate': 6e-6,
    'hidden_dims': 256,
    'bidirectional': False,
    'cuda': False
}

model = None
optimizer = None
criterion = None
train_acc = -1
max_f1 = max_acc = max_rec = max_prec = -1


def load_features(root=None):
    """np.load(...)['arr_0'] of the audio features (squeeze axis 2) and labels (reference lines 18-22)."""
    global prefix, audio_features, audio_targets, audio_dep_idxs_tmp, audio_non_idxs
    if root is not None:
        prefix = os.path.abspath(root)
    audio_features = np.squeeze(np.load(os.path.join(prefix, 'Features/AudioWhole/whole_samples_clf_256.npz'))['arr_0'], axis=2)
    audio_targets = np.load(os.path.join(prefix, 'Features/AudioWhole/whole_labels_clf_256.npz'))['arr_0']
    audio_dep_idxs_tmp = np.where(audio_targets == 1)[0]
    audio_non_idxs = np.where(audio_targets == 0)[0]


class AudioBiLSTM(models.AudioGRU):
    """LayerNorm -> 2-layer GRU -> mean over T -> MLP -> Softmax (reference lines 24-108)."""

    def __init__(self, config, seed=None):
        super().__init__(config, variant='clf', seed=seed)


def model_performance(y_test, y_test_pred_proba):
    y_test_pred = _common.predict(y_test_pred_proba)
    conf_matrix = standard_confusion_matrix(y_test, y_test_pred.cpu().numpy())
    print("Confusion Matrix:")
    print(conf_matrix)
    return y_test_pred, conf_matrix


def train(epoch, train_idxs):
    """One epoch of mini-batch training (reference lines 161-201).  Under torch.distributed every global
    mini-batch is split across ranks; the result equals the single-process run on the same data."""
    global train_acc
    model.train()
    total = nn.LossSum(model.device)                 # device-side sum of the step losses, read once per epoch
    correct_dev = torch.zeros((), dtype=torch.int64, device=model.device)      # counted on the device, read once per epoch
    n_train = len(train_idxs)
    Y_dev = _common.device_labels(audio_targets[train_idxs], model.device, config['num_classes'])
    feed = _common.FeatureFeeder(audio_features, train_idxs, model.device, role='audio_features')       # rows of X_train = audio_features[train_idxs], in HBM
    batches = [(_common.rank_slice(lo, hi), hi - lo) for lo, hi in _common.minibatches(n_train, config['batch_size'])]
    for bi, ((a, b), n_glob) in enumerate(batches):
        parallel.set_global_count(n_glob)
        if b <= a:                                  # this rank owns no row of a small (ragged) mini-batch: zero-contribution step
            total.add(nn.empty_shard_step(model, optimizer))
            continue
        x = feed.rows(a, b, then=batches[bi + 1][0] if bi + 1 < len(batches) else None)
        y = Y_dev[a:b]
        optimizer.zero_grad()
        output = model(x)
        _common.count_correct(output, y, correct_dev)          # arg-max, comparison and running count: one launch
        loss = criterion(output, y)
        loss.backward()
        optimizer.step()
        total.add(loss, model)
    parallel.set_global_count(None)
    total_loss = total.item()                        # the epoch's only host synchronisation on the loss (raises if a sweep gave up)
    correct = int(parallel.all_reduce_sum(correct_dev).item())                  # one collective per epoch, on every rank
    train_acc = correct
    if parallel.rank() == 0:
        print('Train Epoch: {:2d}\t Learning rate: {:.4f}\tLoss: {:.6f}\t Accuracy: {}/{} ({:.0f}%)\n '
              .format(epoch + 1, config['learning_rate'], total_loss, correct, n_train,
                      100. * correct / n_train))


def evaluate(model, test_idxs, fold, train_idxs_tmp, train_idxs):
    """Full-batch evaluation, metrics and the threshold-gated checkpoint (reference lines 204-245)."""
    global max_f1, max_acc, max_prec, max_rec
    model.eval()
    x = _common.FeatureFeeder(audio_features, test_idxs, model.device, role='audio_features').rows(0, len(test_idxs))
    y = torch.from_numpy(np.ascontiguousarray(audio_targets[test_idxs])).type(torch.LongTensor)
    output = model(x)
    loss = criterion(output, y)
    total_loss = loss.item()
    y_test_pred, conf_matrix = model_performance(y, output)
    accuracy, precision, recall, f1_score = _common.prf(conf_matrix)
    print("Accuracy: {}".format(accuracy))
    print("Precision: {}".format(precision))
    print("Recall: {}".format(recall))
    print("F1-Score: {}\n".format(f1_score))
    print('=' * 89)
    if max_f1 <= f1_score and train_acc > len(train_idxs) * 0.90 and f1_score > 0.5:
        max_f1, max_acc, max_rec, max_prec = f1_score, accuracy, recall, precision
        save(model, os.path.join(prefix, 'Model/ClassificationWhole/Audio/BiLSTM_{}_vlad{}_{}_{:.2f}_{}'.format(
            'gru', config['embedding_size'], config['hidden_dims'], max_f1, fold)))
        if parallel.rank() == 0:
            np.save(os.path.join(prefix, 'Features/TextWhole/train_idxs_{:.2f}_{}.npy'.format(f1_score, fold)), train_idxs_tmp)
        print('*' * 64)
        print('model saved: f1: {}\tacc: {}'.format(max_f1, max_acc))
        print('*' * 64)
    return total_loss


def get_param_group(model):
    """Two AdamW groups by the substring test the reference uses (lines 247-255): names containing 'ln' get
    weight_decay 0, everything else 1e-5."""
    nd_list, param_list = [], []
    for name, param in model.named_parameters():
        (nd_list if 'ln' in name else param_list).append(param)
    return [{'params': param_list, 'weight_decay': 1e-5}, {'params': nd_list, 'weight_decay': 0}]


def fold_split(train_idxs_tmp):
    """One fold's index lists (reference lines 268-299): the test volunteers are everyone outside `train_idxs_tmp`; every
    depressed TRAINING volunteer is replaced by all 6 time-axis permutations, every depressed TEST volunteer by permutations
    0, 1, 4, 5, appended to the module's feature / label arrays (which therefore grow from fold to fold).  Returns
    (train_idxs, test_idxs).  Pinned to the reference's own loop statements by tests/golden/fold_bodies.npz."""
    global audio_features, audio_targets
    dep = set(audio_dep_idxs_tmp.tolist())
    test_idxs_tmp = list(set(list(audio_dep_idxs_tmp) + list(audio_non_idxs)) - set(train_idxs_tmp))
    audio_features, audio_targets, train_idxs = _common.permutation_augment(
        audio_features, audio_targets, train_idxs_tmp, lambda i: i in dep, (0, 1, 2, 3, 4, 5), label=1)
    audio_features, audio_targets, test_idxs = _common.permutation_augment(
        audio_features, audio_targets, test_idxs_tmp, lambda i: i in dep, (0, 1, 4, 5), label=1)
    return train_idxs, test_idxs


def main(fold_files=('train_idxs_0.63_1.npy', 'train_idxs_0.60_2.npy', 'train_idxs_0.60_3.npy'), epochs=None):
    """The 3-fold driver (reference lines 257-319): permutation augmentation of the depressed class
    (train: all 6 orders, test: orders 0,1,4,5), fresh model + AdamW + CE per fold, epochs-1 epochs."""
    global model, optimizer, criterion, audio_features, audio_targets
    global max_f1, max_acc, max_rec, max_prec, train_acc
    parallel.init_from_env()
    if audio_features is None:
        load_features()
    folds = [np.load(os.path.join(prefix, 'Features/TextWhole', f), allow_pickle=True) for f in fold_files]
    for fold, train_idxs_tmp in enumerate(folds, start=1):
        train_idxs, test_idxs = fold_split(train_idxs_tmp)
        model = AudioBiLSTM(config)
        parallel.broadcast_params(model)
        optimizer = nn.AdamW(get_param_group(model), lr=config['learning_rate'])
        criterion = nn.CrossEntropyLoss()
        max_f1 = max_acc = max_rec = max_prec = -1
        train_acc = -1
        for ep in range(1, config['epochs'] if epochs is None else epochs):
            train(ep, train_idxs)
            evaluate(model, test_idxs, fold, train_idxs_tmp, train_idxs)


if __name__ == '__main__':
    main()
