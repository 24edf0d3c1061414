#!/usr/bin/env python3
"""Time the three big GEMM forms of cfg2 in isolation: python tools/bench_gemm.py [split|f32]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L
mode = sys.argv[1] if len(sys.argv) > 1 else 'split'
fn = L.gemm_split if mode == 'split' else L.gemm
dev = torch.device('cuda:0')
BT, H3, H = 512 * 300, 768, 256
X = torch.randn(BT, H, device=dev); W = torch.randn(H3, H, device=dev); G = torch.randn(BT, H3, device=dev)
C1 = torch.empty(BT, H3, device=dev); C2 = torch.empty(BT, H, device=dev); C3 = torch.empty(H3, H, device=dev)
ws = L.gemm_ws(1, 0, H3, H, BT, dev)
cases = {'NT proj (BT,768)=X(BT,256) W^T': lambda: fn(0, 1, BT, H3, H, X, H, W, H, C1, H3),
         'NN dX (BT,256)=G(BT,768) W': lambda: fn(0, 0, BT, H, H3, G, H3, W, H, C2, H),
         'TN dW (768,256)=G^T X': lambda: fn(1, 0, H3, H, BT, G, H3, X, H, C3, H, ws=ws)}
for name, f in cases.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'{mode:5s} {name:36s} {ms:7.3f} ms  {2*BT*H3*H/ms/1e9:7.1f} TF')
