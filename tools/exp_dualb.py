#!/usr/bin/env python3
"""Would one dW GEMM per layer (C (768, 512) = dgi^T [in | y]) beat the two it replaces (768x256 and 512x256, K = B*T)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L
dev = torch.device('cuda:0')
BT, H3, H = 512 * 300, 768, 256
G = torch.randn(BT, H3, device=dev); X2 = torch.randn(BT, 2 * H, device=dev); X = torch.randn(BT, H, device=dev); Gh = torch.randn(BT, H, device=dev)
C = torch.empty(H3, 2 * H, device=dev); C1 = torch.empty(H3, H, device=dev); C2 = torch.empty(2 * H, H, device=dev); C3 = torch.empty(H, H, device=dev)
ws = L.gemm_ws(1, 0, H3, 2 * H, BT, dev)
cases = {'fused  (768,512) = G^T [in|y]': lambda: L.gemm_split(1, 0, H3, 2 * H, BT, G, H3, X2, 2 * H, C, 2 * H, ws=ws),
         'dW_ih  (768,256) = G^T in': lambda: L.gemm_split(1, 0, H3, H, BT, G, H3, X, H, C1, H, ws=ws),
         'dW_hh  (512,256) = G[:, :512]^T y': lambda: L.gemm_split(1, 0, 2 * H, H, BT, G, H3, X, H, C2, H, seq_T=300, shiftB=-1, ws=ws),
         'dW_hn  (256,256) = Ghn^T y': lambda: L.gemm_split(1, 0, H, H, BT, Gh, H, X, H, C3, H, seq_T=300, shiftB=-1, ws=ws)}
for name, f in cases.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:40s} {e0.elapsed_time(e1) / 10:7.3f} ms')
