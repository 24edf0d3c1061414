"""Drop-in for Regression/audio_bilstm_perm.py on MI355X (SDS-score regressor: 2-layer GRU -> sum over T ->
MLP -> ReLU, L1Loss, Adam).  Same `config`, `AudioBiLSTM`, `train(epoch) -> train_mae`,
`evaluate(fold, model, train_mae) -> total_loss` and module globals (`train_dep_idxs`, `train_non_idxs`,
`test_dep_idxs`, `test_non_idxs`, `audio_features`, `audio_targets`, `min_mae`, `min_rmse`)."""
import os

import numpy as np
import torch

from . import _common, models, nn, parallel
from ._common import save  # noqa: F401

prefix = os.path.abspath(os.path.join(os.getcwd(), "./"))
audio_features = None
audio_targets = None
dep_idxs = None
non_idxs = None

config = {
    'num_classes': 1,
    'dropout': 0.5,
    'rnn_layers': 2,
    'embedding_size': 256,
    'batch_size': 2,
    'epochs': 120,
    'learning_rate': 1e-5,
    'hidden_dims': 256,
    'bidirectional': False,
    'cuda': False
}

model = None
optimizer = None
criterion = None
train_dep_idxs = []
train_non_idxs = []
test_dep_idxs = []
test_non_idxs = []
min_mae = 100
min_rmse = 100


def load_features(root=None):
    """Reference lines 17-30: regression features/labels (.npz 'arr_0') and the dep/non index .npy files."""
    global prefix, audio_features, audio_targets, dep_idxs, non_idxs
    if root is not None:
        prefix = os.path.abspath(root)
    audio_features = np.squeeze(np.load(os.path.join(prefix, 'Features/AudioWhole/whole_samples_reg_256.npz'))['arr_0'], axis=2)
    audio_targets = np.load(os.path.join(prefix, 'Features/AudioWhole/whole_labels_reg_256.npz'))['arr_0']
    dep_idxs = np.load(os.path.join(prefix, 'Features/AudioWhole/dep_idxs.npy'), allow_pickle=True)
    non_idxs = np.load(os.path.join(prefix, 'Features/AudioWhole/non_idxs.npy'), allow_pickle=True)


class AudioBiLSTM(models.AudioGRU):
    def __init__(self, config, seed=None):
        super().__init__(config, variant='reg', seed=seed)


def _mae_rmse(y, pred):
    y = np.asarray(y, np.float64); pred = np.asarray(pred, np.float64)
    return float(np.mean(np.abs(y - pred))), float(np.sqrt(np.mean((y - pred) ** 2)))


def train(epoch):
    """Reference lines 134-172."""
    model.train()
    total = nn.LossSum(model.device)                 # device-side sum of the step losses, read once per epoch
    idx = list(train_dep_idxs) + list(train_non_idxs)
    pred_dev = _common.prediction_buffer(len(idx), model.device)       # zero-filled; every rank writes its own rows
    Y_train = audio_targets[idx]
    Y_dev = _common.device_labels(Y_train, model.device)
    feed = _common.FeatureFeeder(audio_features, idx, model.device, role='audio_features')       # rows of X_train = audio_features[idx], in HBM
    batches = [((lo, hi), _common.rank_slice(lo, hi)) for lo, hi in _common.minibatches(len(idx), config['batch_size'])]
    for bi, ((lo, hi), (a, b)) in enumerate(batches):
        parallel.set_global_count(hi - lo)
        if b <= a:                                  # empty shard of a small (ragged) mini-batch: zero-contribution step
            total.add(nn.empty_shard_step(model, optimizer))
            continue
        x = feed.rows(a, b, then=batches[bi + 1][1] if bi + 1 < len(batches) else None)
        y = Y_dev[a:b]
        optimizer.zero_grad()
        output = model(x)
        loss = criterion(output, y.view(-1, 1))
        loss.backward()
        optimizer.step()
        _common.store_predictions(pred_dev, a, output)     # this rank's rows; the others' stay zero until the epoch-end SUM
        total.add(loss, model)
    parallel.set_global_count(None)
    total_loss = total.item()                        # the epoch's only host synchronisation on the loss (raises if a sweep gave up)
    # per step every rank issues: the gradient exchange, then the loss scalar (nn.Loss.item); the predictions of the whole epoch
    # are assembled by ONE all-reduce here -- same sequence on working and empty-shard ranks, no per-step host copy
    pred = parallel.all_reduce_sum(pred_dev).cpu().numpy().astype(np.float64) if len(idx) else np.array([])
    train_mae, train_rmse = _mae_rmse(Y_train, pred)
    if parallel.rank() == 0:
        print('Train Epoch: {:2d}\t Learning rate: {:.4f}\t Loss: {:.4f}\t MAE: {:.4f}\t RMSE: {:.4f}\n '
              .format(epoch + 1, config['learning_rate'], total_loss, train_mae, train_rmse))
    return train_mae


def evaluate(fold, model, train_mae):
    """Reference lines 175-213."""
    global min_mae, min_rmse
    model.eval()
    idx = list(test_dep_idxs) + list(test_non_idxs)
    Y_test = audio_targets[idx]
    x = _common.FeatureFeeder(audio_features, idx, model.device, role='audio_features').rows(0, len(idx))
    y = torch.from_numpy(np.ascontiguousarray(Y_test)).type(torch.FloatTensor)
    output = model(x)
    loss = criterion(output, y.view(-1, 1))
    total_loss = loss.item()
    pred = output.data.flatten().cpu().numpy()
    mae, rmse = _mae_rmse(Y_test, pred)
    print('MAE: {:.4f}\t RMSE: {:.4f}\n'.format(mae, rmse))
    print('=' * 89)
    if mae <= min_mae and mae < 8.5 and train_mae < 13:
        min_mae, min_rmse = mae, rmse
        save(model, os.path.join(prefix, 'Model/Regression/Audio{}/{}_vlad{}_{}_{:.2f}'.format(
            fold + 1, 'gru', config['embedding_size'], config['hidden_dims'], min_mae)))
        print('*' * 64)
        print('model saved: mae: {}\t rmse: {}'.format(min_mae, min_rmse))
        print('*' * 64)
    return total_loss


def fold_split(fold):
    """Fold `fold`'s index lists (reference lines 216-240), left in the module globals train() / evaluate() read: test =
    dep_idxs[10 fold : 10 fold + 10] + non_idxs[44 fold : 44 fold + 44]; the first 14 depressed training volunteers (in
    list(set(...)) order, as the reference iterates them) are replaced by all 6 time-axis permutations with their own score."""
    global audio_features, audio_targets, train_dep_idxs, train_non_idxs, test_dep_idxs, test_non_idxs
    test_dep_idxs_tmp = dep_idxs[fold * 10:(fold + 1) * 10]
    test_non_idxs = non_idxs[fold * 44:(fold + 1) * 44]
    train_dep_idxs_tmp = list(set(dep_idxs) - set(test_dep_idxs_tmp))
    train_non_idxs = list(set(non_idxs) - set(test_non_idxs))
    first14 = set(train_dep_idxs_tmp[:14])
    audio_features, audio_targets, train_dep_idxs = _common.permutation_augment(
        audio_features, audio_targets, train_dep_idxs_tmp, lambda i: i in first14, (0, 1, 2, 3, 4, 5))
    test_dep_idxs = test_dep_idxs_tmp
    return train_dep_idxs, train_non_idxs, test_dep_idxs, test_non_idxs


def main(epochs=None):
    """3-fold driver (reference lines 215-260): 10 dep / 44 non test volunteers per fold; the first 14 depressed
    training volunteers are expanded to all 6 time-axis permutations."""
    global model, optimizer, criterion, audio_features, audio_targets
    global train_dep_idxs, train_non_idxs, test_dep_idxs, test_non_idxs, min_mae, min_rmse
    parallel.init_from_env()
    if audio_features is None:
        load_features()
    for fold in range(3):
        fold_split(fold)
        model = AudioBiLSTM(config)
        parallel.broadcast_params(model)
        optimizer = nn.Adam(model.parameters(), lr=config['learning_rate'])
        criterion = nn.L1Loss()
        min_mae = 100; min_rmse = 100
        train_mae = 100
        for ep in range(1, config['epochs'] if epochs is None else epochs):
            train_mae = train(ep)
            evaluate(fold, model, train_mae)


if __name__ == '__main__':
    main()
