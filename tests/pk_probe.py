"""Helper of test_presplit_gpu.py: one GRU-256 x2 forward + backward on seeded data; saves every gradient (and dX) to argv[1].
Run twice by the test, with DEP_DGI_PK=0 and =1 in the environment (the switch is read once per process).  `--batch cases.json` runs several
cases in one process (tests/golden/make_device_bits.py: sixteen cases, two processes)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icassp2022_depression_amd import _lib as L  # noqa: E402

def run_case(out, B, T, F, flags):
    if os.environ.get('PROBE_INSTANCES'):
        L.instance_log_enable(True)
    want_dx = 'dx' in flags
    nody = 'nody' in flags                 # GRU: the training step's call form (dpooled only; AudioBiLSTM.backward) -> the HASDY = false instances
    lstm = 'lstm' in flags                 # the BiLSTM-128 x2 stack of the text model instead of the GRU-256 x2 one
    H, Lyr, dirs, G = (128, 2, 2, 4) if lstm else (256, 2, 1, 3)
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(B * 1000 + T)
    k = 1.0 / np.sqrt(H)
    W = []
    for l in range(Lyr):
        for d in range(dirs):
            for shp in ((G * H, F if l == 0 else H * dirs), (G * H, H), (G * H,), (G * H,)):
                W.append(((torch.rand(*shp, generator=g) * 2 - 1) * k).to(dev))
    Gd = [torch.full_like(w, float('nan')) for w in W]
    x = torch.randn(B, T, F, generator=g).to(dev)
    dy = torch.randn(B, T, H * dirs, generator=g).to(dev)
    dx = torch.full((B, T, F), float('nan'), device=dev) if want_dx else None
    if lstm:
        rnn = L.Rnn(L.CELL_LSTM, B, T, F, H, Lyr, 2, True, 0.5, L.POOL_NONE, dev)
        h_n = torch.empty(2 * Lyr, B, H, device=dev)
        dh_n = torch.randn(2 * Lyr, B, H, generator=g).to(dev)
        rnn.forward(x, W, seed=11, h_n=h_n)
        rnn.backward(x, W, Gd, dy=dy, dh_n=dh_n, dx=dx)
    else:
        dpool = torch.randn(B, H, generator=g).to(dev)
        rnn = L.Rnn(L.CELL_GRU, B, T, F, H, Lyr, 1, True, 0.5, L.POOL_MEAN, dev)
        pooled = torch.empty(B, H, device=dev)
        rnn.forward(x, W, seed=11, pooled=pooled)
        rnn.backward(x, W, Gd, dy=None if nody else dy, dpooled=dpool, dx=dx)
    rnn.check()
    torch.cuda.synchronize()
    res = {'g%d' % i: t.cpu().numpy() for i, t in enumerate(Gd)}
    if os.environ.get('PROBE_INSTANCES'):              # which kernel template instances the case launched (dep_instance_log_*)
        res['instances'] = np.array(sorted(L.instance_log_read(reset=True)))
    if lstm:
        res['h_n'] = h_n.cpu().numpy()
    else:
        res['pooled'] = pooled.cpu().numpy()
    if dx is not None:
        res['dx'] = dx.cpu().numpy()
    np.savez(out, **res)


if __name__ == '__main__':
    if sys.argv[1] == '--batch':                  # a JSON file [[out, B, T, F, [flags]], ...]: several cases in ONE process (the switches are per process anyway)
        import json
        for out_, B_, T_, F_, fl_ in json.load(open(sys.argv[2])):
            run_case(out_, int(B_), int(T_), int(F_), list(fl_))
            torch.cuda.empty_cache()
    else:
        run_case(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5:])
