#!/usr/bin/env python3
"""Run one GEMM form (or the GRU sweeps) in a loop for N seconds: python tools/loop_gemm.py [nt|nn|tn|rnn] [seconds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L
what = sys.argv[1] if len(sys.argv) > 1 else 'nt'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device('cuda:0')
BT, H3, H = 512 * 300, 768, 256
X = torch.randn(BT, H, device=dev); W = torch.randn(H3, H, device=dev); G = torch.randn(BT, H3, device=dev)
C1 = torch.empty(BT, H3, device=dev); C2 = torch.empty(BT, H, device=dev); C3 = torch.empty(H3, H, device=dev)
ws = L.gemm_ws(1, 0, H3, H, BT, dev)
if what == 'rnn':
    rnn = L.Rnn('gru', 512, 300, 256, 256, 2, 1, True, 0.5, dev, pool=True)
    wts = [torch.randn(n, device=dev) * 0.05 for n in (768 * 256, 768 * 256, 768, 768)] * 2
fn = {'nt': lambda: L.gemm_split(0, 1, BT, H3, H, X, H, W, H, C1, H3),
      'nn': lambda: L.gemm_split(0, 0, BT, H, H3, G, H3, W, H, C2, H),
      'tn': lambda: L.gemm_split(1, 0, H3, H, BT, G, H3, X, H, C3, H, ws=ws)}.get(what)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(50): fn()
    torch.cuda.synchronize(); n += 50
print(what, n, 'calls', (time.time() - t0) / n * 1e3, 'ms/call')
