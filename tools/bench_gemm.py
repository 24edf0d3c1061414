#!/usr/bin/env python3
"""Time (and check) the big GEMM forms of cfg2 in isolation: python tools/bench_gemm.py [split|f32] [--check]
DEP_GEMM_WS=0|2|3|4 selects the persistent kernel / the wave-specialised kernel with that many register sets of prefetch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L
mode = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'split'
check = '--check' in sys.argv
fn = L.gemm_split if mode == 'split' else L.gemm
dev = torch.device('cuda:0')
T = 300
BT, H3, H = 512 * T, 768, 256
torch.manual_seed(0)
X = torch.randn(BT, H, device=dev); W = torch.randn(H3, H, device=dev); G = torch.randn(BT, H3, device=dev)
C1 = torch.empty(BT, H3, device=dev); C2 = torch.empty(BT, H, device=dev); C3 = torch.empty(H3, H, device=dev); C4 = torch.empty(2 * H, H, device=dev)
ws = L.gemm_ws(1, 0, H3, H, BT, dev)
cases = {'NT proj (BT,768)=X(BT,256) W^T': (lambda: fn(0, 1, BT, H3, H, X, H, W, H, C1, H3), lambda: X.double() @ W.double().t(), C1),
         'NN dX (BT,256)=G(BT,768) W': (lambda: fn(0, 0, BT, H, H3, G, H3, W, H, C2, H), lambda: G.double() @ W.double(), C2),
         'TN dW (768,256)=G^T X': (lambda: fn(1, 0, H3, H, BT, G, H3, X, H, C3, H, ws=ws), lambda: G.double().t() @ X.double(), C3),
         'TN dWhh (512,256)=G[:, :512]^T shift(X)': (lambda: fn(1, 0, 2 * H, H, BT, G, H3, X, H, C4, H, seq_T=T, shiftB=-1, ws=ws), None, C4)}
for name, (f, ref, out) in cases.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    err = ''
    if check:
        if ref is None:        # row-shifted operand: h_{t-1} of the same sequence, zero at t = 0
            Xs = torch.zeros_like(X); Xv = X.view(512, T, H); Xs.view(512, T, H)[:, 1:] = Xv[:, :-1]
            r = G[:, :2 * H].double().t() @ Xs.double()
        else:
            r = ref()
        err = f'  max rel err {float((out.double() - r).abs().max() / r.abs().max()):.2e}'
    print(f'{mode:5s} WS={os.environ.get("DEP_GEMM_WS", "0")} {name:42s} {ms:7.3f} ms{err}')
