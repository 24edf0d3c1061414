#!/usr/bin/env python3
"""Micro-benchmark of the dep_rnn operator alone (no head / optimizer): forward and backward of the
2-layer GRU (cfg2) or BiLSTM (cfg3) stack on synthetic data.   python tools/bench_rnn.py gru|lstm [B T F H]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L  # noqa: E402


def main():
    cell = sys.argv[1] if len(sys.argv) > 1 else 'gru'
    if len(sys.argv) > 5:
        B, T, F, H = map(int, sys.argv[2:6])
    else:
        B, T, F, H = (512, 300, 256, 256) if cell == 'gru' else (512, 300, 1024, 128)
    steps = int(os.environ.get('STEPS', 5))
    dev = torch.device('cuda:0')
    dirs = 1 if cell == 'gru' else 2
    G = 3 if cell == 'gru' else 4
    torch.manual_seed(0)
    x = torch.randn(B, T, F, device=dev)
    W = []
    for l in range(2):
        for d in range(dirs):
            inp = F if l == 0 else H * dirs
            k = H ** -0.5
            W += [(torch.rand(G * H, inp, device=dev) * 2 - 1) * k, (torch.rand(G * H, H, device=dev) * 2 - 1) * k,
                  (torch.rand(G * H, device=dev) * 2 - 1) * k, (torch.rand(G * H, device=dev) * 2 - 1) * k]
    Gd = [torch.empty_like(w) for w in W]
    rnn = L.Rnn(L.CELL_GRU if cell == 'gru' else L.CELL_LSTM, B, T, F, H, 2, dirs, True, 0.5,
                L.POOL_MEAN if cell == 'gru' else L.POOL_NONE, dev, impl=int(os.environ.get('DEP_IMPL', 0)))
    pooled = torch.empty(B, H, device=dev) if cell == 'gru' else None
    hn = torch.empty(2 * dirs, B, H, device=dev)
    dpool = torch.randn(B, H, device=dev) if cell == 'gru' else None
    dy = None if cell == 'gru' else torch.randn(B, T, 2 * H, device=dev)
    dhn = None if cell == 'gru' else torch.randn(2 * dirs, B, H, device=dev)
    dx = torch.empty_like(x)

    def fwd(i):
        rnn.forward(x, W, seed=i, pooled=pooled, h_n=hn)

    def bwd():
        rnn.backward(x, W, Gd, dy=dy, dpooled=dpool, dh_n=dhn, dx=dx)

    fwd(0); bwd(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for i in range(steps):
        e[0].record(); fwd(i + 1); e[1].record(); bwd(); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    rnn.check()          # raises if a cluster sweep gave up waiting for a member
    print(f'{cell} B={B} T={T} F={F} H={H}: fwd {tf / steps:.3f} ms  bwd {tb / steps:.3f} ms  '
          f'total {(tf + tb) / steps:.3f} ms  -> {B / ((tf + tb) / steps) * 1e3:.0f} utt/s (rnn only)')


if __name__ == '__main__':
    main()
