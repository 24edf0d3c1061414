"""Experiment: does a weight-gradient GEMM on a second stream hide behind the (latency-bound) backward sweep, or does it
inflate the sweep's hand-off latency by more than it saves?  Prints the sweep's mean launch duration (HIP events inside
the library) and the wall time of [backward + side GEMM] with the GEMM (a) serial after the backward, (b) concurrent."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L  # noqa: E402

dev = torch.device('cuda:0')
B, T, F, H, Lyr = 512, 300, 256, 256, 2
g = torch.Generator().manual_seed(1)
k = 1.0 / np.sqrt(H)
W = []
for l in range(Lyr):
    for shp in ((3 * H, F if l == 0 else H), (3 * H, H), (3 * H,), (3 * H,)):
        W.append(((torch.rand(*shp, generator=g) * 2 - 1) * k).to(dev))
Gd = [torch.empty_like(w) for w in W]
x = torch.randn(B, T, F, generator=g).to(dev)
dpool = torch.randn(B, H, generator=g).to(dev)
rnn = L.Rnn(L.CELL_GRU, B, T, F, H, Lyr, 1, True, 0.5, L.POOL_MEAN, dev)
pooled = torch.empty(B, H, device=dev)
# side GEMM: one dW-shaped contraction (768 x 256, K = B*T), the TN form
K = B * T
ga = torch.randn(K, 3 * H, device=dev); gb = torch.randn(K, H, device=dev); gc = torch.empty(3 * H, H, device=dev)
ws = L.gemm_ws(1, 0, 3 * H, H, K, dev)
side = torch.cuda.Stream()


def run(mode, n_side, iters=10):
    L.profile_enable(True); L.profile_read()
    walls = []
    for _ in range(iters):
        rnn.forward(x, W, seed=5, pooled=pooled)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == 'concurrent':
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                for _ in range(n_side):
                    L.gemm_split(1, 0, 3 * H, H, K, ga, 3 * H, gb, H, gc, H, ws=ws)
            rnn.backward(x, W, Gd, dpooled=dpool)
        else:
            rnn.backward(x, W, Gd, dpooled=dpool)
            for _ in range(n_side):
                L.gemm_split(1, 0, 3 * H, H, K, ga, 3 * H, gb, H, gc, H, ws=ws)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    rnn.check()
    prof = L.profile_read(); L.profile_enable(False)
    sw = prof['gru_bwd_sweep']
    walls.sort()
    print(f'{mode:10s} side_gemms={n_side}  wall median {walls[len(walls) // 2] * 1e3:.3f} ms   '
          f'bwd sweep {sw[0] / max(sw[1], 1):.3f} ms/launch   tn {prof["gemm_tn"][0] / max(prof["gemm_tn"][1], 1):.3f} ms/launch')


for n in (0, 2, 4):
    run('serial', n)
    if n:
        run('concurrent', n)
