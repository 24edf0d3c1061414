"""Experiment: the latency-bound GRU backward sweep on XCDs 0-3 (two workgroups per CU, DEP_BWD_XHALF=1)
with the power-bound dW-shaped GEMMs confined to XCDs 4-7 (dep_gemm_set_xcds) on a second stream -- do they hide behind it?
CU-masked streams are not honoured here (tools/micro/cumask.hip), so the partition is done by block index: blockIdx % 8 is the
XCD, and the blocks of the other half leave at entry.

    DEP_BWD_XHALF=0|1 python tools/exp_overlap2.py
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L  # noqa: E402

lib = L.load()
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
K = B * T
ga = torch.randn(K, 3 * H, device=dev); gb = torch.randn(K, H, device=dev); gc = torch.empty(3 * H, H, device=dev)
ws = L.gemm_ws(1, 0, 3 * H, H, K, dev)
side = torch.cuda.Stream()
xhalf = os.environ.get('DEP_BWD_XHALF', '0') == '1'


def side_gemms(n, confined):
    if confined:
        lib.dep_gemm_set_xcds(4, 4)
    for _ in range(n):
        L.gemm_split(1, 0, 3 * H, H, K, ga, 3 * H, gb, H, gc, H, ws=ws)
    lib.dep_gemm_set_xcds(0, 8)


def run(mode, n_side, confined, iters=10):
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
                side_gemms(n_side, confined)
            rnn.backward(x, W, Gd, dpooled=dpool)
        else:
            rnn.backward(x, W, Gd, dpooled=dpool)
            side_gemms(n_side, confined)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    rnn.check()
    prof = L.profile_read(); L.profile_enable(False)
    sw = prof['gru_bwd_sweep']; tn = prof['gemm_tn']
    walls.sort()
    print(f'xhalf={int(xhalf)} {mode:10s} side_gemms={n_side} confined={int(confined)}  wall median {walls[len(walls) // 2] * 1e3:.3f} ms   '
          f'bwd sweep {sw[0] / max(sw[1], 1):.3f} ms/launch   tn {tn[0] / max(tn[1], 1):.3f} ms/launch ({tn[1] // iters} per iter)')


# reference gradients of the default configuration cannot be compared across processes here; the parity suite covers XHALF separately
run('serial', 0, False)
for n in (2, 3):
    run('serial', n, False)
    run('serial', n, True)
    run('concurrent', n, True)
    run('concurrent', n, False)
