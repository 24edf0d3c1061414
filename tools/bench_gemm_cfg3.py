#!/usr/bin/env python3
"""Time the four big split-precision GEMM forms of cfg3 (BiLSTM-128 x 2, F = 1024, direction-stacked weights) in isolation and print their
bf16-MFMA rate (3 products per multiply-add):  python tools/bench_gemm_cfg3.py      (env: DEP_GEMM_NT256, DEP_GEMM_BM, DEP_GEMM_PERSIST, ...)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L
dev = torch.device('cuda:0')
T, B, F, H8, H2 = 300, 512, 1024, 1024, 256
BT = B * T
torch.manual_seed(0)
X = torch.randn(BT, F, device=dev); W0 = torch.randn(H8, F, device=dev); G = torch.randn(BT, H8, device=dev)
Y = torch.randn(BT, H2, device=dev); W1 = torch.randn(H8, H2, device=dev)
C0 = torch.empty(BT, H8, device=dev); C1 = torch.empty(H8, F, device=dev); C2 = torch.empty(BT, H2, device=dev); C3 = torch.empty(H8, H2, device=dev)
ws = L.gemm_ws(1, 0, H8, F, BT, dev)
fn = L.gemm_split
cases = {'NT proj l0 (BT,1024)=X(BT,1024) W^T': (lambda: fn(0, 1, BT, H8, F, X, F, W0, F, C0, H8), BT * H8 * F),
         'NT proj l1 (BT,1024)=Y(BT,256) W^T': (lambda: fn(0, 1, BT, H8, H2, Y, H2, W1, H2, C0, H8), BT * H8 * H2),
         'TN dW_ih l0 (1024,1024)=G^T X': (lambda: fn(1, 0, H8, F, BT, G, H8, X, F, C1, F, ws=ws), BT * H8 * F),
         'TN dW_ih l1 (1024,256)=G^T Y': (lambda: fn(1, 0, H8, H2, BT, G, H8, Y, H2, C3, H2, ws=ws), BT * H8 * H2),
         'NN dX l1 (BT,256)=G(BT,1024) W1': (lambda: fn(0, 0, BT, H2, H8, G, H8, W1, H2, C2, H2), BT * H8 * H2)}
for name, (f, mac) in cases.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'{name:42s} {ms:7.3f} ms  {2 * 3 * mac / ms / 1e9:7.1f} TF/s of bf16 MFMA ({2 * mac / ms / 1e9:6.1f} TF/s fp32-equivalent)')
