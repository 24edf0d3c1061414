"""CPU yardstick for bench.py's `cpu_baseline` leg  --  TEST/BENCH INFRASTRUCTURE ONLY.

The reference's CPU path *is* stock PyTorch (`'cuda': False` in every script), and its files cannot travel
to the GPU box, so this module restates the two benchmarked layer stacks on stock torch.nn (allowed here:
the "no torch.nn.GRU/LSTM" rule governs the HIP product path, not the CPU yardstick -- SURVEY 8d).
tests/test_host_cpu.py::test_torch_baseline_matches_reference_fixture (and ..._fusion_...) loads the reference's own
state_dict fixture into these classes and checks identical outputs, so timing them is timing the
reference's arithmetic (kind = "port").

  AudioClf  : Classification/audio_gru_whole.py:59-73,103-108
  TextClf   : Classification/text_bilstm_whole.py:47-114
  FusionClf : Classification/fuse_net_whole.py:245-395 (frozen encoders -> pretrained_feature, bias-free head, MyLoss)
"""
import time

import torch
import torch.nn as tnn
import torch.nn.functional as F


class AudioClf(tnn.Module):
    def __init__(self, emb, hid, layers=2, p=0.5, classes=2):
        super().__init__()
        self.attention_layer = tnn.Sequential(tnn.Linear(hid, hid), tnn.ReLU(inplace=True))     # dead, as in the reference
        self.lstm_net_audio = tnn.GRU(emb, hid, num_layers=layers, dropout=p, batch_first=True)
        self.ln = tnn.LayerNorm(emb)
        self.fc_audio = tnn.Sequential(tnn.Dropout(p), tnn.Linear(hid, hid), tnn.ReLU(), tnn.Dropout(p),
                                       tnn.Linear(hid, classes), tnn.Softmax(dim=1))

    def forward(self, x):
        y, _ = self.lstm_net_audio(self.ln(x))
        return self.fc_audio(y.mean(dim=1))


class TextClf(tnn.Module):
    def __init__(self, emb, hid, layers=2, p=0.5, classes=2):
        super().__init__()
        self.attention_layer = tnn.Sequential(tnn.Linear(hid, hid), tnn.ReLU(inplace=True))
        self.lstm_net = tnn.LSTM(emb, hid, num_layers=layers, dropout=p, bidirectional=True)
        self.fc_out = tnn.Sequential(tnn.Linear(hid, hid), tnn.ReLU(), tnn.Dropout(p), tnn.Linear(hid, classes),
                                     tnn.Softmax(dim=1))
        self.ln1 = tnn.LayerNorm(emb); self.ln2 = tnn.LayerNorm(hid)

    def forward(self, x):
        out, (hn, _) = self.lstm_net(x.permute(1, 0, 2))
        out = out.permute(1, 0, 2); hn = hn.permute(1, 0, 2)
        a, b = torch.chunk(out, 2, -1)
        h = a + b
        q = self.attention_layer(hn.sum(dim=1).unsqueeze(1))
        w = F.softmax(torch.bmm(q, torch.tanh(h).transpose(1, 2)), dim=-1)
        return self.fc_out(torch.bmm(w, h).squeeze(1))


class FusionClf(tnn.Module):
    """fusion_net of Classification/fuse_net_whole.py:245-374: text BiLSTM + attention + fc_out (Dropout, Linear, ReLU,
    Dropout), audio LayerNorm + GRU + SUM pool + fc_audio (Dropout, Linear, ReLU, Dropout), concat -> Linear(no bias)."""

    def __init__(self, text_emb, text_hid, audio_emb, audio_hid, layers=2, p=0.3, classes=2):
        super().__init__()
        self.attention_layer = tnn.Sequential(tnn.Linear(text_hid, text_hid), tnn.ReLU(inplace=True))
        self.lstm_net = tnn.LSTM(text_emb, text_hid, num_layers=layers, dropout=p, bidirectional=True)
        self.fc_out = tnn.Sequential(tnn.Dropout(p), tnn.Linear(text_hid, text_hid), tnn.ReLU(), tnn.Dropout(p))
        self.lstm_net_audio = tnn.GRU(audio_emb, audio_hid, num_layers=layers, dropout=p, batch_first=True)
        self.fc_audio = tnn.Sequential(tnn.Dropout(p), tnn.Linear(audio_hid, audio_hid), tnn.ReLU(), tnn.Dropout(p))
        self.ln = tnn.LayerNorm(audio_emb)
        self.modal_attn = tnn.Linear(text_hid + audio_hid, text_hid + audio_hid, bias=False)     # dead in the classifier
        self.fc_final = tnn.Sequential(tnn.Linear(text_hid + audio_hid, classes, bias=False), tnn.Softmax(dim=1))

    def pretrained_feature(self, xa, xt):
        with torch.no_grad():
            out, (hn, _) = self.lstm_net(xt.permute(1, 0, 2))
            out = out.permute(1, 0, 2); hn = hn.permute(1, 0, 2)
            a, b = torch.chunk(out, 2, -1)
            h = a + b
            q = self.attention_layer(hn.sum(dim=1).unsqueeze(1))
            w = F.softmax(torch.bmm(q, torch.tanh(h).transpose(1, 2)), dim=-1)
            tf = self.fc_out(torch.bmm(w, h).squeeze(1))
            y, _ = self.lstm_net_audio(self.ln(xa))
            af = self.fc_audio(y.sum(dim=1))
        return tf, af

    def forward(self, feat):
        return self.fc_final(feat)


def my_loss(model, tf, af, y):
    """MyLoss (fuse_net_whole.py:376-395): CE(text W[:, :Ht]^T) + CE(audio W[:, Ht:]^T)."""
    W = model.fc_final[0].weight
    Ht = tf.shape[1]
    return F.cross_entropy(tf @ W[:, :Ht].t(), y) + F.cross_entropy(af @ W[:, Ht:].t(), y)


def param_groups(model):
    nd = [p for n, p in model.named_parameters() if 'ln' in n]
    rest = [p for n, p in model.named_parameters() if 'ln' not in n]
    return [{'params': rest, 'weight_decay': 1e-5}, {'params': nd, 'weight_decay': 0}]


def time_train_step(kind, B, T, Fdim, H, steps=3, warmup=1, threads=None, lr=6e-6):
    """Full train step (fwd + CE-on-softmax + bwd + AdamW, dropout on) on CPU; returns (utt/s, seconds/step, threads)."""
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(0)
    if kind == 'fusion':
        return _time_fusion_step(B, T, Fdim, H, steps, warmup, lr)
    model = (AudioClf if kind == 'audio' else TextClf)(Fdim, H)
    opt = torch.optim.AdamW(param_groups(model), lr=lr)
    crit = tnn.CrossEntropyLoss()
    x = torch.randn(B, T, Fdim); y = torch.randint(0, 2, (B,))
    model.train()
    ts = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = crit(model(x.clone().requires_grad_(True)), y)
        loss.backward()
        opt.step()
        loss.item()
        if i >= warmup:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return B / med, med, torch.get_num_threads()


def _time_fusion_step(B, T, Fa, Ha, steps, warmup, lr, Ft=1024, Ht=128):
    """BASELINE configs[3] step on CPU: frozen encoders forward (train-mode dropout, as the reference), head forward,
    MyLoss, backward to fc_final only, Adam."""
    model = FusionClf(Ft, Ht, Fa, Ha)
    for n, p in model.named_parameters():
        p.requires_grad_(n == 'fc_final.0.weight')
    opt = torch.optim.Adam([model.fc_final[0].weight], lr=lr)
    xa = torch.randn(B, T, Fa); xt = torch.randn(B, T, Ft); y = torch.randint(0, 2, (B,))
    model.train()
    ts = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        tf, af = model.pretrained_feature(xa, xt)
        model(torch.cat((tf, af), dim=1))
        loss = my_loss(model, tf, af, y)
        loss.backward()
        opt.step()
        loss.item()
        if i >= warmup:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return B / med, med, torch.get_num_threads()
