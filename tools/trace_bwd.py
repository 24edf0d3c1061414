#!/usr/bin/env python3
"""DEP_TRACE=1 python tools/trace_bwd.py : phase timings (shader cycles) of workgroup 0 of the backward cluster sweep."""
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
W = [(torch.rand(3 * H, F, device=dev) * 2 - 1) * k, (torch.rand(3 * H, H, device=dev) * 2 - 1) * k,
     (torch.rand(3 * H, device=dev) * 2 - 1) * k, (torch.rand(3 * H, device=dev) * 2 - 1) * k]
G = [torch.empty_like(w) for w in W]
rnn = L.Rnn(L.CELL_GRU, B, T, F, H, 1, 1, True, 0.0, L.POOL_MEAN, dev)
pooled = torch.empty(B, H, device=dev); dpool = torch.randn(B, H, device=dev)
for _ in range(2):
    rnn.forward(x, W, pooled=pooled)
    rnn.backward(x, W, G, dpooled=dpool, dx=None)
torch.cuda.synchronize()
off = L.load().dep_rnn_workspace_xbuf_offset(C.byref(rnn.desc))
tr = rnn.workspace[(off + 6400) // 4:(off + 6400) // 4 + 64].view(torch.int64).cpu().numpy().reshape(4, 8)
names = ['gate grads + LDS + barrier', 'prefetch issue + MFMA', 'payload stores + drain', 'barrier + flag', 'poll', 'gather + sum']
if os.environ.get('DEP_BWD_AG', '1') != '0':      # the all-gather step (default since round 5) (rnn_cluster_bwd.hip, AG): one barrier, at the end
    names = ['ring read + gate grads + publish issue', 'publish acknowledged', 'flag + mask draw', 'poll (2 source members)',
             '12 fragment loads + 36 MFMAs + red write', 'barrier + K-quarter sum']
for s in range(4):
    d = [int(tr[s, i + 1] - tr[s, i]) for i in range(6)]
    print(f'step {199 - s}: total {int(tr[s, 6] - tr[s, 0])} cyc ; ' + ' | '.join(f'{n}: {v}' for n, v in zip(names, d)))

print(f'whole loop of workgroup 0: {int(tr[1, 7] - tr[0, 7])} ticks for T={T} steps = {int(tr[1, 7] - tr[0, 7]) / T:.0f} per step')
sv = rnn.workspace[(off + 6400) // 4 + 64:(off + 6400) // 4 + 64 + 32].view(torch.int64).cpu().numpy().reshape(4, 4)
for s_ in range(4):
    a = sv[s_]
    print(f'service wave, step k={100 + s_}: issue {int(a[1] - a[0]) if a[1] else 0} | flush {int(a[2] - a[1]) if a[1] else int(a[2] - a[0])} | wait at barrier#1 {int(a[3] - a[2])}')

import numpy as np
al = rnn.workspace[(off + 6400) // 4 + 128:(off + 6400) // 4 + 128 + T].cpu().numpy().view(np.uint32).astype(np.int64)
d = np.diff(al) & 0xffffffff
print('per-step ticks: mean %.0f  median %.0f  p10 %.0f  p90 %.0f  max %.0f' % (d.mean(), np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max()))
print('steps 40..79:', ' '.join(str(int(x)) for x in d[40:80]))
