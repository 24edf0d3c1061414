#!/usr/bin/env python3
"""DEP_TRACE=1 python tools/trace_fbwd.py : phase timings (shader cycles) of workgroup 0 of the fused two-layer GRU backward
(rnn_fused2_bwd.hip, all-gather form): thread 0 (group 0: layer-1 gate gradients + W_hh(l1) product), thread 256 (group 1: W_ih(l1)
product one step behind + every HBM stream), thread 512 (group 2: layer-0 gate gradients + W_hh(l0) product); fused steps 100..103."""
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
G = [torch.empty_like(w) for w in W]
p = float(os.environ.get('DROP', '0.5'))
rnn = L.Rnn(L.CELL_GRU, B, T, F, H, 2, 1, True, p, L.POOL_MEAN, dev)
pooled = torch.empty(B, H, device=dev); dpool = torch.randn(B, H, device=dev)
for _ in range(3):
    rnn.forward(x, W, pooled=pooled, seed=3)
    rnn.backward(x, W, G, dpooled=dpool, dx=None)
torch.cuda.synchronize()
rnn.check()
off = L.load().dep_rnn_workspace_xbuf_offset(C.byref(rnn.desc))
tr = rnn.workspace[(off + 6400) // 4:(off + 6400) // 4 + 192].view(torch.int64).cpu().numpy().reshape(3, 4, 8)
crit = ['gate gradients + publish issue', 'publish acknowledged', 'flag + poll (2 source members)', '12 fragment loads (2 rounds) + 36 MFMAs + red',
        '-', '(to the barrier)', 'barrier']
g1 = ['(top: last step\'s streams landed)', '-', 'poll (data one step old)', '12 fragment loads + 36 MFMAs + red', 'wait for the issue signal',
      'streams issued (DMA, write-out, mask draw)', 'barrier']
for g, nm in ((0, 'group 0 (layer 1)'), (2, 'group 2 (layer 0)'), (1, 'group 1 (W_ih + streams)')):
    names = g1 if g == 1 else crit
    for s in range(4):
        a = [int(v) for v in tr[g, s]]
        if a[0] == 0:
            continue
        # stamps a wave did not pass in this step keep the value of an earlier one: show differences of the stamps that advanced
        parts, last = [], a[0]
        for i in (range(1, 8) if g == 1 else (1, 2, 3, 4, 6, 7)):        # (slot 5 belongs to group 1; slots 1, 2 of group 1 carry no phase)
            if g == 1 and i == 2:
                continue
            if a[i] > last:
                parts.append(f'{names[i - 1]}: {a[i] - last}'); last = a[i]
        nxt = int(tr[g, s + 1, 0]) if s < 3 else 0
        print(f'{nm} step {100 + s}: ' + (f'total {nxt - a[0]} | ' if nxt else '') + ' | '.join(parts))
