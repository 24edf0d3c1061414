#!/usr/bin/env python3
"""DEP_TRACE=1 python tools/trace_lstm.py [bwd] : phase timings (shader clocks) of workgroup 0, wave 0 of the BiLSTM-128 forward (or, with
`bwd', backward) cluster sweep of layer 0 (rnn_cluster_lstm.hip) at cfg3's T and B; DEP_LSTM_DF=0/1 and DEP_LSTM_SE=0 trace the forms the
defaults replaced."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L  # noqa: E402

B, T, F, H = 512, 300, 256, 128
dev = torch.device('cuda:0')
torch.manual_seed(0)
x = torch.randn(B, T, F, device=dev)
k = H ** -0.5
W = []
for l in range(2):
    for d in range(2):
        for shp in ((4 * H, F if l == 0 else 2 * H), (4 * H, H), (4 * H,), (4 * H,)):
            W.append((torch.rand(*shp, device=dev) * 2 - 1) * k)
rnn = L.Rnn(L.CELL_LSTM, B, T, F, H, 2, 2, True, 0.5, L.POOL_NONE, dev)     # layer 0 (header slot 0: the one traced) writes dropout(y) too
h_n = torch.empty(4, B, H, device=dev)
bwd = 'bwd' in sys.argv[1:]
G = [torch.empty_like(w) for w in W]
dy = torch.randn(B, T, 2 * H, device=dev)
dh_n = torch.randn(4, B, H, device=dev)
for _ in range(3):
    rnn.forward(x, W, seed=5, h_n=h_n)
    if bwd:
        rnn.backward(x, W, G, dy=dy, dh_n=dh_n, dx=None)
torch.cuda.synchronize()
rnn.check()
off = L.load().dep_rnn_workspace_xbuf_offset(C.byref(rnn.desc))
tr = rnn.workspace[(off + 6400) // 4:(off + 6400) // 4 + 64].view(torch.int64).cpu().numpy().reshape(4, 8)
if bwd:
    names = ['ring read + gate gradients + planes + write-out ring', 'barrier', 'LDS fragment reads + 24 MFMAs + partial-dh stores issued',
             'stores acknowledged (+ drain barrier) + flag', 'mask draw + poll', 'gather 4 partials + sum']
elif os.environ.get('DEP_LSTM_DF', '2') == '3':
    names = ['ring read + gates + c, h + publish + re-arm issue + write-out ring', '-', '-', 'fragment loads until no word is the sentinel',
             '24 MFMAs + partial write', 'barrier + K-half sum']
elif os.environ.get('DEP_LSTM_DF', '2') != '0':
    names = ['ring read + gates + c, h + publish issue + write-out ring', 'publish acknowledged', 'flag', 'poll (2 source members, 8 wave flags)',
             '4 fragment loads + mask draw + 24 MFMAs + partial write', 'barrier + K-half sum']
else:
    names = ['LDS fragment reads + 24 MFMAs + partial write', 'barrier + ring read + K-half sum + gates + c, h + publish issue',
             'publish acknowledged + barrier + flag', 'write-out ring (+ mask draw)', 'poll (4 member flags)', 'gather 8 KB + unpack to LDS planes + barrier']
for s in range(4):
    d = [int(tr[s, i + 1] - tr[s, i]) for i in range(6)]
    print(f'step {196 + s}: total {int(tr[s, 6] - tr[s, 0])} ticks ; ' + ' | '.join(f'{n}: {v}' for n, v in zip(names, d)))
print(f'whole loop of workgroup 0: {int(tr[1, 7] - tr[0, 7])} ticks for T={T} steps = {int(tr[1, 7] - tr[0, 7]) / T:.0f} per step')
