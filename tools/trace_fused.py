#!/usr/bin/env python3
"""DEP_TRACE=1 python tools/trace_fused.py : phase timings (shader cycles) of workgroup 0 of the fused 2-layer GRU forward
(rnn_fused2.hip): thread 0 (group 0), thread 256 (group 1) and thread 512 (group 2, input projection + HBM streams)."""
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
for l in range(2):
    W += [(torch.rand(3 * H, F if l == 0 else H, device=dev) * 2 - 1) * k, (torch.rand(3 * H, H, device=dev) * 2 - 1) * k,
          (torch.rand(3 * H, device=dev) * 2 - 1) * k, (torch.rand(3 * H, device=dev) * 2 - 1) * k]
p = float(os.environ.get('DROP', '0.5'))
rnn = L.Rnn(L.CELL_GRU, B, T, F, H, 2, 1, True, p, L.POOL_MEAN, dev)
pooled = torch.empty(B, H, device=dev)
for _ in range(3):
    rnn.forward(x, W, pooled=pooled, seed=3)
torch.cuda.synchronize()
rnn.check()
off = L.load().dep_rnn_workspace_xbuf_offset(C.byref(rnn.desc))
tr = rnn.workspace[(off + 6400) // 4:(off + 6400) // 4 + 192].view(torch.int64).cpu().numpy().reshape(3, 4, 8)
n0 = ['frags+MFMA+red', 'barrier#1', 'gates+publish+deposit+mask draw', 'drain vmcnt', 'barrier#2+flag', 'poll', 'gather->LDS']
for g, row in ((0, 0), (1, 2)):                     # thread 0 (group 0), thread 256 (group 1)
    for s in range(4):
        a = tr[row, s]
        d = [int(a[i + 1] - a[i]) for i in range(7)]
        nxt = int(tr[row, s + 1, 0] - a[7]) if s < 3 else 0
        print(f'g{g} step {100 + s}: start {int(a[0] - tr[0, 0, 0])} total {int(a[7] - a[0])} + barrier#3 {nxt} | ' + ' | '.join(f'{n}: {v}' for n, v in zip(n0, d)))
for s in range(4):
    a = tr[1, s]
    print(f'g2 step {100 + s}: start {int(a[0] - tr[0, 0, 0])} MFMA+red {int(a[1] - a[0])} | barrier#1 {int(a[2] - a[1])} | slot Y (red, gbuf, flush)->#2 {int(a[4] - a[2])} | slot Z (gather, prefetch issue) {int(a[7] - a[4])}'
          + (f' | barrier#3 {int(tr[1, s + 1, 0] - a[7])}' if s < 3 else ''))
