#!/usr/bin/env python3
"""DEP_TRACE=1 python tools/trace_fwd.py : phase timings (shader cycles) of workgroup 0 of the forward cluster sweep."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L  # noqa: E402

B, T, F, H = 512, 300, 256, 256
dev = torch.device('cuda:0')
torch.manual_seed(0)
x = torch.randn(B, T, F, device=dev)
k = H ** -0.5
W = []
for l in range(1):
    W += [(torch.rand(3 * H, F, device=dev) * 2 - 1) * k, (torch.rand(3 * H, H, device=dev) * 2 - 1) * k,
          (torch.rand(3 * H, device=dev) * 2 - 1) * k, (torch.rand(3 * H, device=dev) * 2 - 1) * k]
rnn = L.Rnn(L.CELL_GRU, B, T, F, H, 1, 1, True, 0.0, L.POOL_MEAN, dev)
pooled = torch.empty(B, H, device=dev)
for _ in range(2):
    rnn.forward(x, W, pooled=pooled)
torch.cuda.synchronize()
off = L.load().dep_rnn_workspace_xbuf_offset(C.byref(rnn.desc))
TOFF = int(os.environ.get('TRACE_OFF', 6400))
tr = rnn.workspace[(off + TOFF) // 4:(off + TOFF) // 4 + 64].view(torch.int64).cpu().numpy().reshape(4, 8)
names = ['top->matvec done', 'pair-exch barrier', 'elementwise', 'payload store+drain', 'barrier+flag+y/sv stores+poll', 'poll barrier', 'load h + LDS + barrier']
for s in range(4):
    d = [int(tr[s, i + 1] - tr[s, i]) for i in range(7)]
    nxt = int(tr[s + 1, 0] - tr[s, 7]) if s < 3 else 0
    print(f'step {100 + s}: total {int(tr[s, 7] - tr[s, 0])} cyc ; ' + ' | '.join(f'{n}: {v}' for n, v in zip(names, d)) + f' ; to next top: {nxt}')
