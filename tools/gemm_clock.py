#!/usr/bin/env python3
"""Shader clock DURING the split-precision GEMM forms: python tools/gemm_clock.py [cfg2|cfg3]
A resident probe wave per XCD (tools/micro/libclkprobe.so) samples s_memtime against the 100 MHz counter in 0.25 ms windows while a loop of
one GEMM form runs on another stream; prints ms per call, TFLOP/s of bf16 products (3 per multiply), the mean / min clock of the windows that
lie inside the loop, and the implied matrix-pipe occupancy  = products / (1024 SIMDs x clock x 1024 flop/cycle)  (32 cycles per
v_mfma_f32_32x32x16_bf16; the pipe accepts one every 16 when two waves feed it, tools/micro/mfma_rate: "busy" can exceed 100 % of this)."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icassp2022_depression_amd import _lib as L
P = ctypes.CDLL(os.path.join(ROOT, 'tools', 'micro', 'libclkprobe.so'))
P.clk_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
P.clk_now_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
T = 300; B = 512; BT = B * T
torch.manual_seed(0)
if which == 'cfg2':
    F, H, G = 256, 256, 3
else:
    F, H, G = 1024, 128, 8            # direction-stacked BiLSTM layer 0: N = 2 dirs x 4 gates x 128
X = torch.randn(BT, F, device=dev); W = torch.randn(G * H, F, device=dev); Gd = torch.randn(BT, G * H, device=dev)
C1 = torch.empty(BT, G * H, device=dev); C2 = torch.empty(BT, F, device=dev); C3 = torch.empty(G * H, F, device=dev)
ws = L.gemm_ws(1, 0, G * H, F, BT, dev)
fn = L.gemm_split
cases = [('NT proj  (BT,%d)=X(BT,%d) W^T' % (G * H, F), lambda: fn(0, 1, BT, G * H, F, X, F, W, F, C1, G * H), 2.0 * BT * G * H * F),
         ('NN dX    (BT,%d)=G(BT,%d) W' % (F, G * H), lambda: fn(0, 0, BT, F, G * H, Gd, G * H, W, F, C2, F), 2.0 * BT * G * H * F),
         ('TN dW    (%d,%d)=G^T X' % (G * H, F), lambda: fn(1, 0, G * H, F, BT, Gd, G * H, X, F, C3, F, ws=ws), 2.0 * BT * G * H * F)]
NB, NWIN, WIN = 64, 120, 25000          # 64 probe waves, 120 windows of 0.25 ms (100 MHz ticks)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name, f, flops in cases:
    for _ in range(3): f()
    torch.cuda.synchronize()
    out = torch.zeros(NB, 1 + 3 * NWIN, dtype=torch.int64, device=dev); marks = torch.zeros(2, dtype=torch.int64, device=dev)
    P.clk_probe_launch(ctypes.c_void_p(sb.cuda_stream), ctypes.c_void_p(out.data_ptr()), NB, NWIN, WIN)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        torch.cuda._sleep(200000)
        P.clk_now_launch(ctypes.c_void_p(sa.cuda_stream), ctypes.c_void_p(marks.data_ptr()))
        e0.record(sa)
        n = 0
        for _ in range(40): f(); n += 1
        e1.record(sa)
        P.clk_now_launch(ctypes.c_void_p(sa.cuda_stream), ctypes.c_void_p(marks[1:].data_ptr()))
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    o = out.cpu().numpy(); m = marks.cpu().numpy()
    mhz = []
    for b in range(NB):
        for w in range(NWIN):
            r0, cyc, tk = o[b, 1 + 3 * w: 4 + 3 * w]
            if r0 >= m[0] + 50 and r0 + tk <= m[1] - 50 and tk > 0: mhz.append(cyc / tk * 100.0)
    mhz = np.array(mhz) if mhz else np.array([float('nan')])
    clk = float(mhz.mean())
    tf3 = 3 * flops / (ms * 1e-3) / 1e12
    occ = 3 * flops / (ms * 1e-3) / (1024 * clk * 1e6 * 1024)
    print(f'{which} {name:40s} {ms:7.3f} ms  {tf3:7.1f} TF bf16-products  clock mean {clk:6.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f}, {len(mhz)} windows)  pipe occupancy at that clock {100 * occ:5.1f} %')
# the probe alone: nothing else running
out = torch.zeros(NB, 1 + 3 * NWIN, dtype=torch.int64, device=dev)
P.clk_probe_launch(ctypes.c_void_p(sb.cuda_stream), ctypes.c_void_p(out.data_ptr()), NB, NWIN, WIN)
torch.cuda.synchronize()
o = out.cpu().numpy()
idle = [o[b, 2 + 3 * w] / o[b, 3 + 3 * w] * 100.0 for b in range(NB) for w in range(NWIN) if o[b, 3 + 3 * w] > 0]
print(f'{which} probe alone: clock mean {np.mean(idle):.0f} MHz')
