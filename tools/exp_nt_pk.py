#!/usr/bin/env python3
"""Experiment (round 6, session 13): the NT projections with pre-split (PK) operands -- rows of x and / or the weight -- against the fp32 operands
the kernel splits while staging.  Checks that every combination is bit-identical to the fp32-operand result, then times them."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L  # noqa: E402

dev = torch.device('cuda:0')
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'icassp2022-depression_amd', 'libdep_rnn.so'))


def pk_image(x):
    """(rows, cols) fp32 -> the PK image: physical rows 2j / 2j+1 = (hi, lo) bf16 pairs of logical rows 2j, 2j+1 (gemm_bf16x3.hip)."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    h = hi.view(torch.int16).to(torch.int32) & 0xffff
    l = lo.view(torch.int16).to(torch.int32) & 0xffff
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    out[0::2] = h[0::2] | (h[1::2] << 16)
    out[1::2] = l[0::2] | (l[1::2] << 16)
    return out.view(torch.float32)


def run(M, N, K, tag):
    torch.manual_seed(0)
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    Xp, Wp = pk_image(X), pk_image(W)
    outs = {}
    for fa, fb in ((0, 0), (1, 0), (0, 1), (1, 1)):
        Cm = torch.empty(M, N, device=dev)
        A, Bm = (Xp if fa else X), (Wp if fb else W)

        def f():
            lib.dep_gemm_debug_formats(fa, fb)
            L.gemm_split(0, 1, M, N, K, A, K, Bm, K, Cm, N, bias=b)
            lib.dep_gemm_debug_formats(0, 0)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        outs[(fa, fb)] = Cm.clone()
        same = bool((Cm.view(torch.int32) == outs[(0, 0)].view(torch.int32)).all())
        print(f'{tag}: A {"PK " if fa else "f32"} B {"PK " if fb else "f32"}: {e0.elapsed_time(e1) / 20:.4f} ms  bit-identical to f32/f32: {same}')


run(153600, 768, 256, 'cfg2 NT (153600 x 768, K 256)')
run(153600, 1024, 1024, 'cfg3 NT l0 (153600 x 1024, K 1024)')
run(153600, 1024, 256, 'cfg3 NT l1 (153600 x 1024, K 256)')
