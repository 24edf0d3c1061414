"""Drop-in for Classification/text_bilstm_whole.py on MI355X: `config`, `TextBiLSTM(config)`,
`train(epoch, train_idxs)`, `evaluate(model, test_idxs, fold, train_idxs)`, `get_param_group`, `save`,
metrics helpers and the module-global protocol.  The reference trains at import (it has no __main__
guard, lines 260-314); here `load_features()` / `main()` are explicit."""
import os

import numpy as np
import torch

from . import _common, models, nn, parallel
from ._common import save, standard_confusion_matrix  # noqa: F401

prefix = os.path.abspath(os.path.join(os.getcwd(), "."))
text_features = None
text_targets = None
text_dep_idxs_tmp = None
text_non_idxs = None

config = {
    'num_classes': 2,
    'dropout': 0.5,
    'rnn_layers': 2,
    'embedding_size': 1024,
    'batch_size': 4,
    'epochs': 150,
    'learning_rate': 1e-5,
    'hidden_dims': 128,
    'bidirectional': True,
    'cuda': False,
}

model = None
optimizer = None
criterion = None
train_acc = -1
max_f1 = max_acc = max_rec = max_prec = -1


def load_features(root=None):
    """ELMo sentence features (N,T,1024) and labels, `np.load(...)['arr_0']` (reference lines 17-21)."""
    global prefix, text_features, text_targets, text_dep_idxs_tmp, text_non_idxs
    if root is not None:
        prefix = os.path.abspath(root)
    text_features = np.load(os.path.join(prefix, 'Features/TextWhole/whole_samples_clf_avg.npz'))['arr_0']
    text_targets = np.load(os.path.join(prefix, 'Features/TextWhole/whole_labels_clf_avg.npz'))['arr_0']
    text_dep_idxs_tmp = np.where(text_targets == 1)[0]
    text_non_idxs = np.where(text_targets == 0)[0]


class TextBiLSTM(models.TextBiLSTM):
    """2-layer BiLSTM -> attention_net_with_w -> Linear, ReLU, Dropout, Linear, Softmax (reference lines 23-114)."""

    def __init__(self, config, seed=None):
        super().__init__(config, variant='clf', seed=seed)


def model_performance(y_test, y_test_pred_proba):
    y_test_pred = _common.predict(y_test_pred_proba)
    conf_matrix = standard_confusion_matrix(y_test, y_test_pred)
    print("Confusion Matrix:")
    print(conf_matrix)
    return y_test_pred, conf_matrix


def train(epoch, train_idxs):
    """Reference lines 154-193; data-parallel aware like audio_gru_whole.train."""
    global train_acc
    model.train()
    total = nn.LossSum(model.device)                 # device-side sum of the step losses, read once per epoch
    correct_dev = torch.zeros((), dtype=torch.int64, device=model.device)      # counted on the device, read once per epoch
    n_train = len(train_idxs)
    Y_dev = _common.device_labels(text_targets[train_idxs], model.device, config['num_classes'])
    feed = _common.FeatureFeeder(text_features, train_idxs, model.device, role='text_features')       # rows of X_train = text_features[train_idxs], in HBM
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


def evaluate(model, test_idxs, fold, train_idxs):
    """Reference lines 196-235."""
    global max_f1, max_acc, max_prec, max_rec
    model.eval()
    x = _common.FeatureFeeder(text_features, test_idxs, model.device, role='text_features').rows(0, len(test_idxs))
    y = torch.from_numpy(np.ascontiguousarray(text_targets[test_idxs])).type(torch.LongTensor)
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
    if max_f1 <= f1_score and train_acc > len(train_idxs) * 0.9 and f1_score > 0.5:
        max_f1, max_acc, max_rec, max_prec = f1_score, accuracy, recall, precision
        save(model, os.path.join(prefix, 'Model/ClassificationWhole/Text/BiLSTM_{}_{:.2f}_{}'.format(
            config['hidden_dims'], max_f1, fold)))
        print('*' * 64)
        print('model saved: f1: {}\tacc: {}'.format(max_f1, max_acc))
        print('*' * 64)
    return total_loss


def get_param_group(model):
    """Reference lines 237-245: 'ln' in name -> weight_decay 0, else 1e-5."""
    nd_list, param_list = [], []
    for name, param in model.named_parameters():
        (nd_list if 'ln' in name else param_list).append(param)
    return [{'params': param_list, 'weight_decay': 1e-5}, {'params': nd_list, 'weight_decay': 0}]


def fold_split(train_idxs_tmp, train_keep=(0, 1, 2, 3, 4, 5)):
    """One fold's index lists (reference lines 263-292): as audio_gru_whole.fold_split, on the text arrays.  (`train_keep` is
    for the checker, whose copy of this loop leaks the test-side selection into later folds: model_checking.py.)"""
    global text_features, text_targets
    dep = set(text_dep_idxs_tmp.tolist())
    test_idxs_tmp = list(set(list(text_dep_idxs_tmp) + list(text_non_idxs)) - set(train_idxs_tmp))
    text_features, text_targets, train_idxs = _common.permutation_augment(
        text_features, text_targets, train_idxs_tmp, lambda i: i in dep, tuple(train_keep), label=1)
    text_features, text_targets, test_idxs = _common.permutation_augment(
        text_features, text_targets, test_idxs_tmp, lambda i: i in dep, (0, 1, 4, 5), label=1)
    return train_idxs, test_idxs


def main(fold_files=('train_idxs_0.63_1.npy', 'train_idxs_0.60_2.npy', 'train_idxs_0.60_3.npy'), epochs=None):
    """3-fold driver (reference lines 260-314)."""
    global model, optimizer, criterion, text_features, text_targets
    global max_f1, max_acc, max_rec, max_prec, train_acc
    parallel.init_from_env()
    if text_features is None:
        load_features()
    folds = [np.load(os.path.join(prefix, 'Features/TextWhole', f), allow_pickle=True) for f in fold_files]
    for fold, train_idxs_tmp in enumerate(folds, start=1):
        train_idxs, test_idxs = fold_split(train_idxs_tmp)
        model = TextBiLSTM(config)
        parallel.broadcast_params(model)
        optimizer = nn.AdamW(get_param_group(model), lr=config['learning_rate'])
        criterion = nn.CrossEntropyLoss()
        max_f1 = max_acc = max_rec = max_prec = -1
        train_acc = -1
        for ep in range(1, config['epochs'] if epochs is None else epochs):
            train(ep, train_idxs)
            evaluate(model, test_idxs, fold, train_idxs)


if __name__ == '__main__':
    main()
