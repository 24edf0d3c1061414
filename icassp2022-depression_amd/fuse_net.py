"""Drop-in for Regression/fuse_net.py on MI355X: regression late fusion -- frozen encoders,
`forward = ReLU((sigmoid(x M^T) * x) W^T)` (reference lines 345-351) and the split-weight SmoothL1 `MyLoss`
(lines 353-366).  As in the reference only `fc_final.0.weight` ever receives a gradient (the loss does not
touch `modal_attn`).  Surface: `config`, `fusion_net`, `MyLoss`, `train(model, epoch) -> train_mae`,
`evaluate(model, fold, train_mae) -> total_loss`, globals `fuse_features`, `fuse_targets`,
`train_dep_idxs`, `train_non_idxs`, `test_dep_idxs`, `test_non_idxs`, `min_mae`, `min_rmse`."""
import itertools
import os

import numpy as np
import torch

from . import _common, models, nn, parallel
from ._common import save  # noqa: F401

prefix = os.path.abspath(os.path.join(os.getcwd(), "./"))
fuse_features = None
fuse_targets = None
dep_idxs = non_idxs = None
text_model_paths = ['Model/Regression/Text1/BiLSTM_128_7.75.pt', 'Model/Regression/Text2/BiLSTM_128_8.46.pt',
                    'Model/Regression/Text3/BiLSTM_128_8.01.pt']
audio_model_paths = ['Model/Regression/Audio1/gru_vlad256_256_7.60.pt', 'Model/Regression/Audio2/gru_vlad256_256_8.38.pt',
                     'Model/Regression/Audio3/gru_vlad256_256_8.25.pt']

config = {
    'num_classes': 1,
    'dropout': 0.5,
    'rnn_layers': 2,
    'audio_embed_size': 256,
    'text_embed_size': 1024,
    'batch_size': 4,
    'epochs': 150,
    'learning_rate': 8e-5,
    'audio_hidden_dims': 256,
    'text_hidden_dims': 128,
    'cuda': False,
    'lambda': 1e-2,
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
    """Reference lines 18-31."""
    global prefix, fuse_features, fuse_targets, dep_idxs, non_idxs
    if root is not None:
        prefix = os.path.abspath(root)
    text_features = np.load(os.path.join(prefix, 'Features/TextWhole/whole_samples_reg_avg.npz'))['arr_0']
    text_targets = np.load(os.path.join(prefix, 'Features/TextWhole/whole_labels_reg_avg.npz'))['arr_0']
    audio_features = np.squeeze(np.load(os.path.join(prefix, 'Features/AudioWhole/whole_samples_reg_256.npz'))['arr_0'], axis=2)
    fuse_features = [[audio_features[i], text_features[i]] for i in range(text_features.shape[0])]
    fuse_targets = text_targets
    dep_idxs = np.load(os.path.join(prefix, 'Features/AudioWhole/dep_idxs.npy'), allow_pickle=True)
    non_idxs = np.load(os.path.join(prefix, 'Features/AudioWhole/non_idxs.npy'), allow_pickle=True)


class fusion_net(models.FusionNet):
    def __init__(self, text_embed_size, text_hidden_dims, rnn_layers, dropout, num_classes,
                 audio_hidden_dims, audio_embed_size, seed=None):
        super().__init__(text_embed_size, text_hidden_dims, rnn_layers, dropout, num_classes, audio_hidden_dims,
                         audio_embed_size, variant='reg', seed=seed)


class MyLoss(models.MyLoss):
    def __init__(self):
        super().__init__('reg')


def build(seed=None):
    global model, optimizer, criterion
    model = fusion_net(config['text_embed_size'], config['text_hidden_dims'], config['rnn_layers'], config['dropout'],
                       config['num_classes'], config['audio_hidden_dims'], config['audio_embed_size'], seed=seed)
    optimizer = nn.Adam(model.parameters(), lr=config['learning_rate'])
    criterion = MyLoss()
    return model


def _mae_rmse(y, pred):
    y = np.asarray(y, np.float64); pred = np.asarray(pred, np.float64)
    return float(np.mean(np.abs(y - pred))), float(np.sqrt(np.mean((y - pred) ** 2)))


def train(model, epoch):
    """Reference lines 373-412."""
    model.train()
    total = nn.LossSum(model.device)                 # device-side sum of the step losses, read once per epoch
    idx = list(train_dep_idxs) + list(train_non_idxs)
    pred_dev = _common.prediction_buffer(len(idx), model.device)       # zero-filled; every rank writes its own rows
    Y_train = [fuse_targets[i] for i in idx]
    feed = _common.PairFeeder(fuse_features, idx, model.device)
    for lo, hi in _common.minibatches(len(idx), config['batch_size']):
        a, b = _common.rank_slice(lo, hi)
        parallel.set_global_count(hi - lo)
        y = Y_train[a:b]
        if b <= a:                                  # empty shard of a small mini-batch: zero-contribution step
            total.add(nn.empty_shard_step(model, optimizer))
            continue
        optimizer.zero_grad()
        text_feature, audio_feature = model.pretrained_feature(feed.rows(a, b))
        output = model(_common.concat_features(text_feature, audio_feature))
        loss = criterion(text_feature, audio_feature, y, model)
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


def evaluate(model, fold, train_mae):
    """Reference lines 414-456."""
    global min_mae, min_rmse
    model.eval()
    total_loss = 0
    pred = np.array([])
    idx = list(test_dep_idxs) + list(test_non_idxs)
    Y_test = [fuse_targets[i] for i in idx]
    feed = _common.PairFeeder(fuse_features, idx, model.device)
    for lo, hi in _common.minibatches(len(idx), config['batch_size']):
        y = Y_test[lo:hi]
        text_feature, audio_feature = model.pretrained_feature(feed.rows(lo, hi))
        output = model(_common.concat_features(text_feature, audio_feature))
        loss = criterion(text_feature, audio_feature, y, model)
        pred = np.hstack((pred, output.data.flatten().cpu().numpy()))
        total_loss += loss.item()
    mae, rmse = _mae_rmse(Y_test, pred)
    print('MAE: {:.4f}\t RMSE: {:.4f}\n'.format(mae, rmse))
    print('=' * 89)
    if mae <= min_mae and mae < 8.2 and train_mae < 13:
        min_mae, min_rmse = mae, rmse
        save(model, os.path.join(prefix, 'Model/Regression/Fuse{}/fuse_{:.2f}'.format(fold + 1, min_mae)))
        print('*' * 64)
        print('model saved: mae: {}\t rmse: {}'.format(min_mae, min_rmse))
        print('*' * 64)
    return total_loss


def _single_modality_eval(model, col, crit):
    idx = list(test_dep_idxs) + list(test_non_idxs)
    X_test = np.array([fuse_features[i][col] for i in idx])
    Y_test = np.array([fuse_targets[i] for i in idx])
    model.eval()
    x = torch.from_numpy(np.ascontiguousarray(X_test)).type(torch.FloatTensor)
    y = torch.from_numpy(np.ascontiguousarray(Y_test)).type(torch.FloatTensor)
    optimizer.zero_grad()
    output = model(x)
    loss = crit(output, y.view(-1, 1))
    loss.item()
    pred = output.data.flatten().cpu().numpy()
    mae, rmse = _mae_rmse(Y_test, pred)
    print('MAE: {:.4f}\t RMSE: {:.4f}\n'.format(mae, rmse))
    print('=' * 89)


def evaluate_audio(model):
    """Reference lines 458-490: the AUDIO regressor alone (`model(x)` on the audio half of every test pair, one full
    batch) scored with the module-global `criterion` (a two-argument loss at that point of the reference's workflow);
    prints MAE / RMSE, returns nothing."""
    _single_modality_eval(model, 0, criterion)


def evaluate_text(model):
    """Reference lines 492-524: the TEXT regressor alone, scored with a local SmoothL1Loss."""
    _single_modality_eval(model, 1, nn.SmoothL1Loss())


def transplant(model, text_state_dict, audio_state_dict):
    """Reference lines 562-583 (no `ln` here; every parameter keeps requires_grad=True, yet only
    fc_final.0.weight is reached by the loss)."""
    audio_keys = ['lstm_net_audio.weight_ih_l0', 'lstm_net_audio.weight_hh_l0', 'lstm_net_audio.bias_ih_l0',
                  'lstm_net_audio.bias_hh_l0', 'lstm_net_audio.weight_ih_l1', 'lstm_net_audio.weight_hh_l1',
                  'lstm_net_audio.bias_ih_l1', 'lstm_net_audio.bias_hh_l1', 'fc_audio.1.weight', 'fc_audio.1.bias',
                  'fc_audio.4.weight', 'fc_audio.4.bias']
    model.load_state_dict(text_state_dict, strict=False)
    model.load_state_dict({k: audio_state_dict[k] for k in audio_keys}, strict=False)
    for param in model.parameters():
        param.requires_grad = True


def main(epochs=None):
    """3-fold driver (reference lines 526-592)."""
    global model, optimizer, criterion, fuse_targets
    global train_dep_idxs, train_non_idxs, test_dep_idxs, test_non_idxs, min_mae, min_rmse
    parallel.init_from_env()
    if fuse_features is None:
        load_features()
    for fold in range(3):
        test_dep_idxs_tmp = dep_idxs[fold * 10:(fold + 1) * 10]
        test_non_idxs = non_idxs[fold * 44:(fold + 1) * 44]
        train_dep_idxs_tmp = list(set(dep_idxs) - set(test_dep_idxs_tmp))
        train_non_idxs = list(set(non_idxs) - set(test_non_idxs))
        train_dep_idxs = []
        for i, idx in enumerate(train_dep_idxs_tmp):
            feat = fuse_features[idx]
            if i < 14:
                for pa, pt in zip(itertools.permutations(feat[0], 3), itertools.permutations(feat[1], 3)):
                    fuse_features.append([np.stack(pa), np.stack(pt)])
                    fuse_targets = np.hstack((fuse_targets, fuse_targets[idx]))
                    train_dep_idxs.append(len(fuse_features) - 1)
            else:
                train_dep_idxs.append(idx)
        test_dep_idxs = test_dep_idxs_tmp
        build()
        parallel.broadcast_params(model)
        transplant(model, _common.load_checkpoint_state_dict(os.path.join(prefix, text_model_paths[fold])),
                   _common.load_checkpoint_state_dict(os.path.join(prefix, audio_model_paths[fold])))
        min_mae = 100; min_rmse = 100
        train_mae = 100
        for ep in range(1, config['epochs'] if epochs is None else epochs):
            train_mae = train(model, ep)
            evaluate(model, fold, train_mae)


if __name__ == '__main__':
    main()
