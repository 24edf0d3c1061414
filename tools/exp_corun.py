"""Round 4 experiment: what do a latency-bound GRU backward sweep and a power-bound dW GEMM cost each other on DISJOINT XCD halves?

Round 3's tools/exp_overlap2.py could not answer that: a confined persistent GEMM still owns blocks on the other half (they leave
at entry, but must be PLACED first), and next to a sweep that fills those CUs' LDS they are placed only when the sweep ends -- the
side GEMM's completion event (and every later launch of its stream) waited for the sweep.  Here

  * the sweep is the default burst-stream kernel, ONE workgroup per CU, on XCDs 0-3 (DEP_BWD_XHALF=2: half a batch, 16 tiles x 8 members);
  * the main stream's own GEMMs are confined to XCDs 0-3 as well (they run between the sweeps, not beside them);
  * the side stream carries ONE long TN contraction (K = 4 x 153,600 rows) confined to XCDs 4-7, launched FIRST, so that its blocks
    of XCDs 0-3 have come and gone before the first sweep is dispatched.

    DEP_BWD_XHALF=2 python tools/exp_corun.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
B, T, F, H, Lyr = 256, 300, 256, 256, 2
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
K1 = 512 * T
ga = torch.randn(4 * K1, 3 * H, device=dev); gb = torch.randn(4 * K1, H, device=dev); gc = torch.empty(3 * H, H, device=dev)
side = torch.cuda.Stream()
xh = os.environ.get('DEP_BWD_XHALF', '0')


def tn(Krows, lo, n, ws):
    lib.dep_gemm_set_xcds(lo, n)
    L.gemm_split(1, 0, 3 * H, H, Krows, ga, 3 * H, gb, H, gc, H, ws=ws)
    lib.dep_gemm_set_xcds(0, 8)


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


ws1 = L.gemm_ws(1, 0, 3 * H, H, K1, dev); ws4 = L.gemm_ws(1, 0, 3 * H, H, 4 * K1, dev)
print(f'# DEP_BWD_XHALF={xh}  B={B} (16 tiles), GRU-256 x2 backward; side = TN 768x256 over K rows')
print(f'TN K=153600 all XCDs        {timed(lambda: tn(K1, 0, 8, ws1)):.3f} ms')
print(f'TN K=153600 XCDs 4-7 only   {timed(lambda: tn(K1, 4, 4, ws1)):.3f} ms')
print(f'TN K=614400 all XCDs        {timed(lambda: tn(4 * K1, 0, 8, ws4)):.3f} ms')
t_side_alone = timed(lambda: tn(4 * K1, 4, 4, ws4))
print(f'TN K=614400 XCDs 4-7 only   {t_side_alone:.3f} ms')


def main_loop(n, confined):
    if confined:
        lib.dep_gemm_set_xcds(0, 4)
    for _ in range(n):
        rnn.backward(x, W, Gd, dpooled=dpool)
    lib.dep_gemm_set_xcds(0, 8)


def run(tag, n_main, with_side, confined, iters=5):
    rnn.forward(x, W, seed=5, pooled=pooled)
    torch.cuda.synchronize()
    walls, sides = [], []
    L.profile_enable(True); L.profile_read()
    for _ in range(iters):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if with_side:
            with torch.cuda.stream(side):
                e0.record()
                L.profile_enable(False)
                tn(4 * K1, 4, 4, ws4)
                L.profile_enable(True)
                e1.record()
            time.sleep(0.0002)                 # the side kernel's blocks of XCDs 0-3 have left before the first sweep arrives
        main_loop(n_main, confined)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
        if with_side:
            sides.append(e0.elapsed_time(e1))
    rnn.check()
    prof = L.profile_read(); L.profile_enable(False)
    sw = prof['gru_bwd_sweep']; tnp = prof['gemm_tn']; nn_ = prof['gemm_nn']
    walls.sort(); sides.sort()
    s = f'{tag:34s} wall {walls[len(walls) // 2]:.3f} ms for {n_main} backward(s): sweep {sw[0] / max(sw[1], 1):.3f} ms/launch, ' \
        f'own TN {tnp[0] / max(tnp[1], 1):.3f}, own NN {nn_[0] / max(nn_[1], 1):.3f} ms/launch'
    if with_side:
        s += f' | side GEMM {sides[len(sides) // 2]:.3f} ms (alone {t_side_alone:.3f})'
    print(s)


n = 1
run('main alone, own GEMMs all XCDs', n, False, False)
run('main alone, own GEMMs XCDs 0-3', n, False, True)
run('main + side (4-7), own on 0-3', n, True, True)
n = 2
run('2x main alone, own GEMMs XCDs 0-3', n, False, True)
run('2x main + side (4-7), own on 0-3', n, True, True)
