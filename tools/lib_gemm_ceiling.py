#!/usr/bin/env python3
"""Calibration only (never a product path): what the vendor bf16 GEMM (torch.matmul -> hipBLASLt / rocBLAS) sustains on this box at the shapes of the
split-precision contractions, i.e. the ceiling a perfectly tuned single-product kernel reaches here.  Three such products = one bf16x3 contraction."""
import sys
import torch

dev = torch.device('cuda:0')


def bench(name, a, b, flops, n=20):
    for _ in range(3):
        a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'{name}: {ms:.4f} ms per product = {flops / ms / 1e9:.0f} TFLOP/s bf16 ; x3 products = {3 * ms:.4f} ms')


K = 153600
for (M, N, tag) in ((768, 256, 'cfg2 TN dW (768 x 256, K = 153600)'), (1024, 1024, 'cfg3 TN dW_ih l0 (1024 x 1024)'), (1024, 256, 'cfg3 TN (1024 x 256)')):
    a = torch.randn(K, M, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    bench(tag, a.t(), b, 2.0 * M * N * K)
for (N, Kk, tag) in ((768, 256, 'cfg2 NT projection (153600 x 768, K = 256)'), (1024, 1024, 'cfg3 NT projection l0 (153600 x 1024, K = 1024)'), (1024, 256, 'cfg3 NT projection l1 (K = 256)')):
    a = torch.randn(K, Kk, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, Kk, device=dev, dtype=torch.bfloat16)
    bench(tag, a, w.t(), 2.0 * K * N * Kk)
# square reference
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
bench('8192^3 reference', a, b, 2.0 * 8192 ** 3)
