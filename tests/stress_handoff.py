"""Stress driver of the cross-CU hand-off of the cluster sweeps (run by tests/test_stress_gpu.py in a subprocess, because
the library reads DEP_CLUSTER_NOFAST / DEP_NUM_CUS once per process).

    python tests/stress_handoff.py --cell gru --iters 20 [--load]

Full grid (B = 512, T = 300: 512 / 256 co-resident workgroups, 32 clusters polling flags concurrently), forward + backward
`iters` times on the same inputs.  Before every iteration the reserve and the whole workspace (exchange payload buffers
included) are poisoned with NaN bit patterns, so a member that reads a stale or not-yet-written payload word produces a
NaN or a different bit pattern: every iteration must reproduce the first one bit for bit, and dep_rnn_status must stay
clean.  --load runs a GEMM loop on a second stream while the sweeps run (uneven load: the GEMM's workgroups take CU
slots, delay cluster members and keep the L2 / fabric busy).  Prints one JSON line.

Finding this driver produced (round 2): the GRU forward kernels that fill a CU's register file -- the fused two-layer launch
(rnn_fused2.hip, twelve 168-VGPR waves per CU) and the 16-unit-member kernel (rnn_cluster16.hip, two 5-wave workgroups per
CU) -- leave no room for anybody else, so a foreign workgroup that arrives while the launch is still being
dispatched cannot be placed, blocks the dispatcher, and the not-yet-resident cluster members never arrive: the sweep
gives up LOUDLY (status 5 -> DepError), never silently.  That kernel therefore needs the GPU to itself (the product runs
the forward on one stream with nothing beside it; DEP_FUSED2=0 DEP_CLUSTER16=0 selects the 4-wave one-workgroup-per-CU
forward that tolerates co-scheduled kernels).  Every other sweep kernel (one workgroup per CU, LDS / VGPR headroom left) is exercised under
load here, which is what an overlapped gradient all-reduce needs: collectives only ever overlap the BACKWARD sweeps.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cell', default='gru')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--B', type=int, default=512)
    ap.add_argument('--T', type=int, default=300)
    ap.add_argument('--F', type=int, default=64)
    ap.add_argument('--H', type=int, default=0)
    ap.add_argument('--load', action='store_true')
    ap.add_argument('--torch-load', action='store_true', help='the side-stream load is torch.matmul instead of dep_gemm_bf16x3')
    ap.add_argument('--fwd-only', action='store_true')
    ap.add_argument('--model-form', action='store_true', help='GRU: the training step\'s gradient inputs (dpooled only, no dy, no dX): '
                    'gru2_bwd_fused<.., HASDY = false, ..>, three input slots / prefetch distance 2')
    ap.add_argument('--load-phase', default='both', help='both | bwd : which half of the step the side-stream load overlaps')
    ap.add_argument('--load-kind', default='split', help='split | f32 | torch')
    ap.add_argument('--load-m', type=int, default=4096)
    ap.add_argument('--load-stream', default='side', help='side | main | side-sync | side-once')
    ap.add_argument('--two-refs', action='store_true', help='a loaded iteration may equal the exclusive-forward reference OR the '
                    'reference of the co-schedule-tolerant forward (the device-side fallback of dep_rnn_forward), bit for bit')
    a = ap.parse_args()
    from icassp2022_depression_amd import _lib as L
    dev = torch.device('cuda:0')
    cell = L.CELL_GRU if a.cell == 'gru' else L.CELL_LSTM
    dirs = 1 if a.cell == 'gru' else 2
    G = 3 if a.cell == 'gru' else 4
    H = a.H or (256 if a.cell == 'gru' else 128)
    B, T, F, Lyr = a.B, a.T, a.F, 2
    g = torch.Generator().manual_seed(7)
    k = 1.0 / np.sqrt(H)
    W = []
    for l in range(Lyr):
        for _ in range(dirs):
            inp = F if l == 0 else H * dirs
            for shp in ((G * H, inp), (G * H, H), (G * H,), (G * H,)):
                W.append(((torch.rand(*shp, generator=g) * 2 - 1) * k).to(dev))
    Gd = [torch.empty_like(w) for w in W]
    x = torch.randn(B, T, F, generator=g).to(dev)
    dy = (torch.randn(B, T, H * dirs, generator=g) * 0.3).to(dev)
    dpool = torch.randn(B, H, generator=g).to(dev) if a.cell == 'gru' else None
    dhn = (torch.randn(Lyr * dirs, B, H, generator=g) * 0.3).to(dev) if a.cell != 'gru' else None
    rnn = L.Rnn(cell, B, T, F, H, Lyr, dirs, True, 0.5, L.POOL_MEAN if a.cell == 'gru' else L.POOL_NONE, dev, impl=3)
    pooled = torch.empty(B, H, device=dev) if a.cell == 'gru' else None
    h_n = torch.empty(Lyr * dirs, B, H, device=dev)
    dx = torch.empty(B, T, F, device=dev)
    model_form = a.model_form and a.cell == 'gru'

    side = torch.cuda.Stream()
    stop = [False]
    if a.load:
        M = a.load_m
        ga = torch.randn(M, M, device=dev); gb = torch.randn(M, M, device=dev); gc = torch.empty(M, M, device=dev)

    once = [False]

    def load_burst(n):
        if a.load_stream == 'side-once':
            if once[0]:
                return
            once[0] = True
        import contextlib
        with (contextlib.nullcontext() if a.load_stream == 'main' else torch.cuda.stream(side)):
            for _ in range(n):
                if a.torch_load or a.load_kind == 'torch':
                    torch.matmul(ga, gb, out=gc)
                elif a.load_kind == 'f32':
                    L.gemm(0, 1, M, M, M, ga, M, gb, M, gc, M)
                else:
                    L.gemm_split(0, 1, M, M, M, ga, M, gb, M, gc, M)
        if a.load_stream in ('side-sync', 'side-once'):
            torch.cuda.synchronize()

    ref = None
    ref2 = None
    fallbacks = 0
    mismatches = 0
    status_bad = 0
    nan_seen = 0
    detail = []
    errs = []
    import time
    t_iter = []
    first = -2 if a.two_refs else -1
    for it in range(first, a.iters):                        # iteration -1: the unloaded reference run (-2: the same on the tolerant kernels)
        loaded = a.load and it >= 0
        if a.two_refs:
            L.load().dep_rnn_set_exclusive(0 if it == -2 else 1)
        t0 = time.perf_counter()
        rnn.reserve.view(torch.int32).fill_(-1)             # 0xffffffff: a NaN pattern in every word
        rnn.workspace.view(torch.int32).fill_(-1)
        for t in Gd:
            t.fill_(float('nan'))
        if loaded and a.load_phase == 'both':
            load_burst(3 + it % 4)                         # the sweeps start while the side stream's GEMMs hold CUs
        rnn.forward(x, W, seed=99, pooled=pooled, h_n=h_n)
        fb_word = rnn.fallback_word()
        fb_copy = fb_word.clone() if fb_word is not None else None          # stream-ordered: read after the forward, before the next one clears it
        if loaded:
            if a.load_phase == 'bwd':
                torch.cuda.synchronize()                    # the forward ran alone; the load overlaps the backward only
            load_burst(2 + it % 5)
        if not a.fwd_only:
            rnn.backward(x, W, Gd, dy=None if model_form else dy, dpooled=dpool, dh_n=dhn, dx=None if model_form else dx)
            if model_form:
                dx.zero_()
        else:
            dx.zero_(); [t.zero_() for t in Gd]
        try:
            rnn.check()
        except L.DepError as e:
            status_bad += 1
            if len(errs) < 3:
                errs.append(str(e)[-60:])
        t_iter.append(round(time.perf_counter() - t0, 3))
        outs = [rnn.layer_output().clone(), rnn.layer_output(0).clone(), h_n.clone(), dx.clone()] + [t.clone() for t in Gd]
        names = ['y_top', 'y_l0', 'h_n', 'dx'] + ['dW%d' % i for i in range(len(Gd))]
        if L.get_gemm_mode() == 3 and a.cell == 'gru':      # bf16-storage mode: the reserve's sequences are bf16 arrays (half of each fp32 slot stays poisoned)
            outs, names = outs[2:], names[2:]
        if pooled is not None:
            outs.append(pooled.clone()); names.append('pooled')
        if any(bool(torch.isnan(o).any()) for o in outs):
            nan_seen += 1
        if fb_copy is not None and int(fb_copy.item()) != 0:
            fallbacks += 1
        if a.two_refs and it == -2:
            ref2 = outs
        elif ref is None:
            ref = outs
        elif ref2 is not None and all(torch.equal(u, v) for u, v in zip(ref2, outs)):
            pass                                            # the device fell back to the tolerant kernels: their bits
        elif not all(torch.equal(u, v) for u, v in zip(ref, outs)):
            mismatches += 1
            if len(detail) < 6:
                for nm, u, v in zip(names, ref, outs):
                    if not torch.equal(u, v):
                        d = (u != v) | (torch.isnan(u) != torch.isnan(v))
                        idx = d.nonzero()
                        detail.append({'iter': it, 'tensor': nm, 'n_diff': int(d.sum()), 'max_abs': float((u - v).abs().nan_to_num(1e30).max()),
                                       'first': idx[0].tolist(), 'last': idx[-1].tolist(),
                                       'rows': sorted(set(idx[:, 0].tolist()))[:24] if idx.dim() == 2 and idx.shape[1] >= 2 else None})
    torch.cuda.synchronize()
    print(json.dumps({'cell': a.cell, 'iters': a.iters, 'mismatches': mismatches, 'status_bad': status_bad,
                      'nan_iters': nan_seen, 'load': bool(a.load), 'fallbacks': fallbacks,
                      'nofast': os.environ.get('DEP_CLUSTER_NOFAST', ''), 'num_cus': os.environ.get('DEP_NUM_CUS', ''), 'errs': errs, 't_iter': t_iter, 'detail': detail[:2]}))
    return 0 if (mismatches == 0 and status_bad == 0 and nan_seen == 0) else 1


if __name__ == '__main__':
    sys.exit(main())
