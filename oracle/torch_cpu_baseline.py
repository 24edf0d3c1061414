"""CPU yardstick for bench.py's `cpu_baseline` leg  --  TEST/BENCH INFRASTRUCTURE ONLY.

The reference's CPU path *is* stock PyTorch (`'cuda': False` in every script), and its files cannot travel
to the GPU box, so this module restates the two benchmarked layer stacks on stock torch.nn (allowed here:
the "no torch.nn.GRU/LSTM" rule governs the HIP product path, not the CPU yardstick -- SURVEY 8d).
tests/test_oracle_golden.py::test_torch_baseline_matches_reference_fixture loads the reference's own
state_dict fixture into these classes and checks identical outputs, so timing them is timing the
reference's arithmetic (kind = "port").

  AudioClf : Classification/audio_gru_whole.py:59-73,103-108
  TextClf  : Classification/text_bilstm_whole.py:47-114
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


def param_groups(model):
    nd = [p for n, p in model.named_parameters() if 'ln' in n]
    rest = [p for n, p in model.named_parameters() if 'ln' not in n]
    return [{'params': rest, 'weight_decay': 1e-5}, {'params': nd, 'weight_decay': 0}]


def time_train_step(kind, B, T, Fdim, H, steps=3, warmup=1, threads=None, lr=6e-6):
    """Full train step (fwd + CE-on-softmax + bwd + AdamW, dropout on) on CPU; returns (utt/s, seconds/step, threads)."""
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(0)
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
