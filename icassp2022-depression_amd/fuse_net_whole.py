"""Drop-in for Classification/fuse_net_whole.py on MI355X: late fusion of the frozen audio-GRU and
text-BiLSTM encoders through `Linear(384, 2, bias=False)` + Softmax, trained with the split-weight `MyLoss`.
Same `config`, `fusion_net(...)`, `MyLoss()`, `train(epoch, train_idxs)`, `evaluate(model, test_idxs, fold,
train_idxs)`, module-level `model` / `optimizer` / `criterion`, `fuse_features` (python list of
[audio_i (T,256), text_i (T,1024)] pairs) and `fuse_targets`.  The reference's unused imports (librosa,
tensorflow, allennlp ...) are not needed."""
import itertools
import os

import numpy as np
import torch

from . import _common, models, nn, parallel
from ._common import save, standard_confusion_matrix  # noqa: F401

prefix = os.path.abspath(os.path.join(os.getcwd(), "./"))
text_features = text_targets = audio_features = audio_targets = None
fuse_features = None
fuse_targets = None
fuse_dep_idxs = None
fuse_non_idxs = None

config = {
    'num_classes': 2,
    'dropout': 0.3,
    'rnn_layers': 2,
    'audio_embed_size': 256,
    'text_embed_size': 1024,
    'batch_size': 2,
    'epochs': 100,
    'learning_rate': 8e-6,
    'audio_hidden_dims': 256,
    'text_hidden_dims': 128,
    'cuda': False,
    'lambda': 1e-5,
}

model = None
optimizer = None
criterion = None
max_f1 = max_acc = max_train_acc = train_acc = -1


def load_features(root=None):
    """Reference lines 18-28."""
    global prefix, text_features, text_targets, audio_features, audio_targets
    global fuse_features, fuse_targets, fuse_dep_idxs, fuse_non_idxs
    if root is not None:
        prefix = os.path.abspath(root)
    text_features = np.load(os.path.join(prefix, 'Features/TextWhole/whole_samples_clf_avg.npz'))['arr_0']
    text_targets = np.load(os.path.join(prefix, 'Features/TextWhole/whole_labels_clf_avg.npz'))['arr_0']
    audio_features = np.squeeze(np.load(os.path.join(prefix, 'Features/AudioWhole/whole_samples_clf_256.npz'))['arr_0'], axis=2)
    audio_targets = np.load(os.path.join(prefix, 'Features/AudioWhole/whole_labels_clf_256.npz'))['arr_0']
    fuse_features = [[audio_features[i], text_features[i]] for i in range(text_features.shape[0])]
    fuse_targets = text_targets
    fuse_dep_idxs = np.where(text_targets == 1)[0]
    fuse_non_idxs = np.where(text_targets == 0)[0]


def model_performance(y_test, y_test_pred_proba):
    """Here the second argument already holds predicted labels (reference lines 54-66)."""
    y_test_pred = y_test_pred_proba
    conf_matrix = standard_confusion_matrix(y_test, y_test_pred)
    print("Confusion Matrix:")
    print(conf_matrix)
    return y_test_pred, conf_matrix


class fusion_net(models.FusionNet):
    """Reference lines 245-374."""

    def __init__(self, text_embed_size, text_hidden_dims, rnn_layers, dropout, num_classes,
                 audio_hidden_dims, audio_embed_size, seed=None):
        super().__init__(text_embed_size, text_hidden_dims, rnn_layers, dropout, num_classes, audio_hidden_dims,
                         audio_embed_size, variant='clf', seed=seed)


class MyLoss(models.MyLoss):
    """CE(text half) + CE(audio half) on the split fc_final weight (reference lines 376-395)."""

    def __init__(self):
        super().__init__('clf')


def build(seed=None):
    """The module-level construction the reference performs at import (lines 413-419)."""
    global model, optimizer, criterion
    model = fusion_net(config['text_embed_size'], config['text_hidden_dims'], config['rnn_layers'], config['dropout'],
                       config['num_classes'], config['audio_hidden_dims'], config['audio_embed_size'], seed=seed)
    optimizer = nn.Adam(model.parameters(), lr=config['learning_rate'])
    criterion = MyLoss()
    return model


def _batch(items, labels, lo, hi):
    a, b = _common.rank_slice(lo, hi)
    return items[a:b], labels[a:b]


def train(epoch, train_idxs):
    """Reference lines 421-465."""
    global max_train_acc, train_acc
    model.train()
    total = nn.LossSum(model.device)                 # device-side sum of the step losses, read once per epoch
    correct_dev = torch.zeros((), dtype=torch.int64, device=model.device)      # counted on the device, read once per epoch
    n_train = len(train_idxs)
    Y_train = [fuse_targets[idx] for idx in train_idxs]
    Y_dev = _common.device_labels(np.asarray(Y_train), model.device, config['num_classes'])
    feed = _common.PairFeeder(fuse_features, train_idxs, model.device)       # the pairs X_train = [fuse_features[i] ...], in HBM
    for lo, hi in _common.minibatches(n_train, config['batch_size']):
        a, b = _common.rank_slice(lo, hi)
        y = Y_train[a:b]
        parallel.set_global_count(hi - lo)
        if b <= a:                             # empty shard of a small mini-batch (batch_size 2 < world): zero-contribution step
            total.add(nn.empty_shard_step(model, optimizer))
            continue
        optimizer.zero_grad()
        text_feature, audio_feature = model.pretrained_feature(feed.rows(a, b))
        concat_x = _common.concat_features(text_feature, audio_feature)
        output = model(concat_x)
        _common.count_correct(output, Y_dev[a:b], correct_dev)
        loss = criterion(text_feature, audio_feature, y, model)
        loss.backward()
        optimizer.step()
        total.add(loss, model)
    parallel.set_global_count(None)
    total_loss = total.item()                        # the epoch's only host synchronisation on the loss (raises if a sweep gave up)
    correct = int(parallel.all_reduce_sum(correct_dev).item())                  # one collective per epoch, on every rank
    max_train_acc = correct
    train_acc = correct
    if parallel.rank() == 0:
        print('Train Epoch: {:2d}\t Learning rate: {:.4f}\tLoss: {:.6f}\t Accuracy: {}/{} ({:.0f}%)\n '.format(
            epoch, config['learning_rate'], total_loss / n_train, correct, n_train,
            100. * correct / n_train))


def evaluate(model, test_idxs, fold, train_idxs):
    """Mini-batched evaluation (reference lines 468-520)."""
    global max_train_acc, max_acc, max_f1
    model.eval()
    total_loss = 0
    Y_test = [fuse_targets[idx] for idx in test_idxs]
    pred_dev = torch.empty(len(Y_test), 1, dtype=torch.int64, device=model.device)      # every mini-batch's arg-max lands here
    feed = _common.PairFeeder(fuse_features, test_idxs, model.device)
    for lo, hi in _common.minibatches(len(Y_test), config['batch_size']):
        y = Y_test[lo:hi]
        text_feature, audio_feature = model.pretrained_feature(feed.rows(lo, hi))
        output = model(_common.concat_features(text_feature, audio_feature))
        loss = criterion(text_feature, audio_feature, y, model)
        _common.predict(output, out=pred_dev[lo:hi])
        total_loss += loss.item()
    pred = pred_dev.cpu()
    y_test_pred, conf_matrix = model_performance(Y_test, pred)
    print('\nTest set: Average loss: {:.4f}'.format(total_loss / len(Y_test)))
    print('Calculating additional test metrics...')
    accuracy, precision, recall, f1_score = _common.prf(conf_matrix)
    print("Accuracy: {}".format(accuracy))
    print("Precision: {}".format(precision))
    print("Recall: {}".format(recall))
    print("F1-Score: {}\n".format(f1_score))
    print('=' * 89)
    if max_f1 < f1_score and max_train_acc >= len(train_idxs) * 0.9 and f1_score > 0.61:
        max_f1, max_acc = f1_score, accuracy
        save(model, os.path.join(prefix, 'Model/ClassificationWhole/Fuse/fuse_{:.2f}_{}'.format(max_f1, fold)))
        print('*' * 64)
        print('model saved: f1: {}\tacc: {}'.format(max_f1, max_acc))
        print('*' * 64)
    return total_loss


def transplant(model, text_state_dict, audio_state_dict):
    """The name-based weight transplant of reference lines 566-588: every key of the text checkpoint that
    exists in fusion_net is taken (strict=False: e.g. `fc_out.0.*` of a text_bilstm_whole checkpoint has
    no counterpart and is dropped -- SURVEY 2.1 quirk 5), then the listed audio keys (`fc_audio.4.*` is
    listed but has no counterpart and is dropped too)."""
    audio_keys = ['lstm_net_audio.weight_ih_l0', 'lstm_net_audio.weight_hh_l0', 'lstm_net_audio.bias_ih_l0',
                  'lstm_net_audio.bias_hh_l0', 'lstm_net_audio.weight_ih_l1', 'lstm_net_audio.weight_hh_l1',
                  'lstm_net_audio.bias_ih_l1', 'lstm_net_audio.bias_hh_l1', 'fc_audio.1.weight', 'fc_audio.1.bias',
                  'fc_audio.4.weight', 'fc_audio.4.bias', 'ln.weight', 'ln.bias']
    model.load_state_dict(text_state_dict, strict=False)
    model.load_state_dict({k: audio_state_dict[k] for k in audio_keys}, strict=False)
    for param in model.parameters():
        param.requires_grad = False
    model.fc_final[0].weight.requires_grad = True


def augment_pairs(train_idxs_tmp, test_idxs_tmp):
    """Reference lines 531-564: zip(permutations(audio), permutations(text)) for depressed volunteers."""
    global fuse_features, fuse_targets
    dep = set(np.asarray(fuse_dep_idxs).tolist())

    def expand(idxs, keep):
        global fuse_targets
        out = []
        for idx in idxs:
            if idx in dep:
                feat = fuse_features[idx]
                pairs = zip(itertools.permutations(feat[0], 3), itertools.permutations(feat[1], 3))
                for count, fuse_perm in enumerate(pairs):
                    if count in keep:
                        fuse_features.append([np.stack(fuse_perm[0]), np.stack(fuse_perm[1])])
                        fuse_targets = np.hstack((fuse_targets, 1))
                        out.append(len(fuse_features) - 1)
            else:
                out.append(idx)
        return out
    return expand(train_idxs_tmp, range(6)), expand(test_idxs_tmp, (0, 1, 4, 5))


def main(idxs_paths=('train_idxs_0.63_1.npy', 'train_idxs_0.65_2.npy', 'train_idxs_0.60_3.npy'),
         text_model_paths=('BiLSTM_128_0.64_1.pt', 'BiLSTM_128_0.66_2.pt', 'BiLSTM_128_0.62_3.pt'),
         audio_model_paths=('BiLSTM_gru_vlad256_256_0.67_1.pt', 'BiLSTM_gru_vlad256_256_0.67_2.pt',
                            'BiLSTM_gru_vlad256_256_0.63_3.pt'), epochs=None):
    """3-fold driver (reference lines 522-603)."""
    global max_f1, max_acc, max_train_acc
    parallel.init_from_env()
    if fuse_features is None:
        load_features()
    if model is None:
        build()
    for fold in range(1, 4):
        train_idxs_tmp = np.load(os.path.join(prefix, 'Features/TextWhole/{}'.format(idxs_paths[fold - 1])), allow_pickle=True)
        test_idxs_tmp = list(set(list(fuse_dep_idxs) + list(fuse_non_idxs)) - set(train_idxs_tmp))
        train_idxs, test_idxs = augment_pairs(train_idxs_tmp, test_idxs_tmp)
        tsd = _common.load_checkpoint_state_dict(os.path.join(prefix, 'Model/ClassificationWhole/Text', text_model_paths[fold - 1]))
        asd = _common.load_checkpoint_state_dict(os.path.join(prefix, 'Model/ClassificationWhole/Audio', audio_model_paths[fold - 1]))
        transplant(model, tsd, asd)
        max_f1 = max_acc = max_train_acc = -1
        for ep in range(1, config['epochs'] if epochs is None else epochs):
            train(ep, train_idxs)
            evaluate(model, test_idxs, fold, train_idxs)


if __name__ == '__main__':
    main()
