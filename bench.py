#!/usr/bin/env python3
"""Headline benchmark: utterances/s of one full train step (forward + CE-on-softmax loss + backward +
AdamW, dropout on) of the audio GRU-256 x2 classifier on synthetic (B,T,F) = (512,300,256) per GPU
(BASELINE.json configs[1]), weak-scaled over N GPUs with one RCCL gradient all-reduce per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload audio_gru|text_bilstm|fusion]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under torch.distributed.run (one rank per
GPU, 127.0.0.1 rendezvous); the process group's size must equal --gpus or the run aborts.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the persistent recurrent sweep with the largest
total time): SURVEY 8(d)'s algorithmic flops per launch over its mean launch duration, measured with HIP events on the
launch stream during the timed region, against the peak of the matrix pipe the sweep runs on (fp32 MFMA in exact mode; the
bf16 pipe / 3 for the 3-term split) -- 8(d) shows the path is matrix-pipe bound, not HBM bound.  Beside it: `achieved_hbm`
with 8(d)'s compulsory bytes AND this design's bytes (both labelled), the serial floor (dependent steps at 1 us), and
`traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json), refused when the
kernel sources changed since they were measured.  `eval_forward` is the forward-only rate.  `cpu_baseline` (N=1 only)
times the same step on the host cores with stock torch.nn (oracle/torch_cpu_baseline.py, validated against the reference's
fixtures): median of 5 steps after 2 warm-ups.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2516.6    # dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0             # HBM3E peak
WORKLOADS = {
    # name: (module, class, B per GPU, T, F, H)
    'audio_gru': ('audio_gru_whole', 'AudioBiLSTM', 512, 300, 256, 256),
    'text_bilstm': ('text_bilstm_whole', 'TextBiLSTM', 512, 300, 1024, 128),
    # BASELINE.json configs[3]: late fusion -- frozen audio-GRU + text-BiLSTM encoders (forward only, dropout active as in
    # the reference's train() mode), concat, bias-free linear head trained with the split-weight MyLoss and Adam
    'fusion': ('fuse_net_whole', 'fusion_net', 512, 300, 256, 256),
}


def usable_cores():
    """Host cores this process may really use: min(affinity mask, cgroup CPU quota).  os.cpu_count() alone
    over-subscribes a quota-limited container (256 logical CPUs visible, 16 granted on the GPU box)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(n, argv):
    """`bench.py --gpus N` started as a plain process: become N ranks (one per GPU) of one torch.distributed.run job."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ); env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.execvpe(cmd[0], cmd, env)


def launch_check(args):
    """CPU-testable part of the multi-rank launch (tests/test_host_cpu.py): join the group with the requested backend,
    check its size against --gpus, all-reduce one number, print one JSON line from rank 0."""
    from icassp2022_depression_amd import parallel
    parallel.init_from_env(args.backend)
    world = parallel.world_size()
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the process group has {world} ranks')
    t = torch.ones(1)
    if args.backend == 'nccl':
        t = t.cuda()
    parallel.all_reduce_sum(t)
    parallel.barrier()
    if parallel.rank() == 0:
        print(json.dumps({'launch_check': True, 'world': world, 'sum': float(t.item()), 'backend': args.backend}))


def build_workload(name, dev, rank, world):
    """Model, optimizer, synthetic HBM-resident batch and the train-step closure of one WORKLOADS entry."""
    import importlib
    from icassp2022_depression_amd import _common, nn, parallel
    modname, cls, B, T, F, H = WORKLOADS[name]
    mod = importlib.import_module('icassp2022_depression_amd.' + modname)
    torch.manual_seed(0)
    g = torch.Generator(device='cpu'); g.manual_seed(1234 + rank)
    y = torch.randint(0, 2, (B,), generator=g).to(dev)
    out = {'mod': mod, 'B': B, 'T': T, 'F': F, 'H': H, 'y': y}
    if name == 'fusion':
        cfg = dict(mod.config); cfg.update(audio_embed_size=F, audio_hidden_dims=H, text_embed_size=1024, text_hidden_dims=128)
        model = mod.fusion_net(cfg['text_embed_size'], cfg['text_hidden_dims'], cfg['rnn_layers'], cfg['dropout'],
                               cfg['num_classes'], cfg['audio_hidden_dims'], cfg['audio_embed_size'], seed=0)
        parallel.broadcast_params(model)
        optimizer = nn.Adam(model.parameters(), lr=cfg['learning_rate'])
        criterion = mod.MyLoss()
        xa = torch.randn(B, T, F, generator=g).to(dev)       # synthetic paired features, resident in HBM
        xt = torch.randn(B, T, 1024, generator=g).to(dev)
        model.train()

        def step():
            parallel.set_global_count(B * world)
            optimizer.zero_grad()
            tf, af = model.pretrained_feature((xa, xt))
            model(_common.concat_features(tf, af))          # dep_copy2d x 2, as the product loop (fuse_net_whole.train)
            loss = criterion(tf, af, y, model)
            loss.backward()
            optimizer.step()
            return loss
        out.update(xa=xa, xt=xt, eval_fn=lambda: model.pretrained_feature((xa, xt)))
    else:
        cfg = dict(mod.config); cfg.update(embedding_size=F, hidden_dims=H)
        model = getattr(mod, cls)(cfg, seed=0)
        parallel.broadcast_params(model)
        optimizer = nn.AdamW(mod.get_param_group(model), lr=cfg['learning_rate'])
        criterion = nn.CrossEntropyLoss()
        x = torch.randn(B, T, F, generator=g).to(dev)        # synthetic features, resident in HBM
        model.train()

        def step():
            parallel.set_global_count(B * world)
            optimizer.zero_grad()
            o = model(x)
            loss = criterion(o, y)
            loss.backward()                                   # includes the RCCL all-reduce of the grad bucket
            optimizer.step()
            return loss
        out.update(x=x, eval_fn=lambda: model(x))
    out.update(model=model, optimizer=optimizer, criterion=criterion, cfg=cfg, step=step)
    return out


def dominant_sweep(prof, steps, name, B, T, H):
    """The recurrent sweep category with the largest total time and SURVEY 8(d)'s algorithmic flops of one of its launches."""
    cats = {k: v for k, v in prof.items() if v[1] > 0}
    sweeps = {k: v for k, v in cats.items() if 'sweep' in k}
    dom = max(sweeps, key=lambda k: sweeps[k][0])
    dom_ms = sweeps[dom][0] / sweeps[dom][1]
    if dom.startswith('lstm'):
        G, dirs, Hs = 4, 2, 128                               # the text encoder's hidden size in every workload
    else:
        G, dirs, Hs = 3, 1, H
    launches_per_step = sweeps[dom][1] / steps               # 2 for per-layer sweeps, 1 for a launch that carries both layers
    layers_per_launch = 2.0 / launches_per_step
    sweep_flops = 2.0 * B * T * (G * Hs) * Hs * dirs * layers_per_launch
    return {'cats': cats, 'dom': dom, 'dom_ms': dom_ms, 'G': G, 'dirs': dirs, 'H': Hs, 'launches_per_step': launches_per_step,
            'layers_per_launch': layers_per_launch, 'sweep_flops': sweep_flops}


# the time-parallel contractions of one train step, algorithmic flops (2 M N K, one product per multiply) per GEMM form.
# GRU stack: layer 0's input projection (layer 1's runs inside the fused forward) and the four weight gradients (layer 1's dX inside the
# fused backward); BiLSTM stack: direction-stacked projections of both layers, dX of layer 1, dW_ih (direction-stacked) and dW_hh (per
# direction) of both layers; fusion: the two frozen encoders' forward projections only.
def gemm_flops_per_step(name, B, T):
    BT = float(B) * T
    gru = {'gemm_nt': 2 * BT * 768 * 256, 'gemm_tn': 4 * 2 * BT * 768 * 256}
    lstm_f = 2 * BT * 1024 * 1024 + 2 * BT * 1024 * 256
    lstm = {'gemm_nt': lstm_f, 'gemm_nn': 2 * BT * 1024 * 256,
            'gemm_tn': 2 * BT * 1024 * 1024 + 2 * BT * 1024 * 256 + 2 * (2 * 2 * BT * 512 * 128)}
    return {'audio_gru': gru, 'text_bilstm': lstm, 'fusion': {'gemm_nt': gru['gemm_nt'] + lstm_f}}[name]


SURVEY_8D_BYTES_PER_STEP = {'audio_gru': 0.944e9, 'text_bilstm': 2.517e9, 'fusion': 0.786e9}     # SURVEY 8(d), fp32 storage column


def pmc_step_bytes(name):
    """HBM bytes of one train step from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json) -- only when the record was taken at the
    kernel digest this library was built from."""
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        stamp = open(os.path.join(ROOT, 'icassp2022-depression_amd', 'libdep_rnn.so.stamp')).read().strip()
        return rec.get('step_bytes', {}).get(name) if rec.get('kernel_digest') == stamp else None
    except Exception:
        return None


def kernel_families(cats, steps, name, B, T, H, split, step_ms):
    """The two kernel families of a step side by side: recurrent sweeps and time-parallel contractions, each with its
    algorithmic flops, ms per step, fraction of the dense matrix-pipe peak of the mode and share of the step; `dominant` = the larger one."""
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
    gf = gemm_flops_per_step(name, B, T)
    fam = {}
    g_ms = sum(v[0] for k, v in cats.items() if k.startswith('gemm')) / steps
    g_fl = sum(gf.get(k, 0.0) for k in cats if k.startswith('gemm'))
    s_ms = sum(v[0] for k, v in cats.items() if 'sweep' in k) / steps
    s_fl = 0.0
    for k in cats:
        if 'sweep' not in k:
            continue
        Gk, dk, Hk = (4, 2, 128) if k.startswith('lstm') else (3, 1, H)
        # both layers' recurrent products; the fused GRU launches also carry layer 1's input projection (forward) / dX (backward)
        extra = 1.5 if (k.startswith('gru') and cats[k][1] / steps < 1.5) else 1.0
        s_fl += 2.0 * B * T * (Gk * Hk) * Hk * dk * 2 * extra
    for key, ms, fl in (('gemm', g_ms, g_fl), ('sweeps', s_ms, s_fl)):
        if ms > 0:
            fam[key] = {'ms_per_step': round(ms, 4), 'gflop_per_step': round(fl / 1e9, 1), 'tflops': round(fl / (ms * 1e-3) / 1e12, 2),
                        'frac_of_mfma_peak': round(fl / (ms * 1e-3) / 1e12 / peak, 4), 'share_of_step': round(ms / step_ms, 4),
                        'kernels': {k: round(v[0] / steps, 4) for k, v in cats.items() if (k.startswith('gemm') if key == 'gemm' else 'sweep' in k)}}
    fam['dominant'] = max((k for k in fam), key=lambda k: fam[k]['ms_per_step']) if fam else None
    fam['peak_tflops'] = round(peak, 1)
    return fam


def time_other_workload(name, dev, L, steps=8, warmup=3):
    """BASELINE configs[2] / [3] in the same invocation: a few train steps after the headline region, so that
    the driver's own `bench.py --gpus 1` run times all three per-GPU workloads."""
    wl = build_workload(name, dev, 0, 1)
    step, model = wl['step'], wl['model']
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    L.profile_enable(True); L.profile_read()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = L.profile_read(); L.profile_enable(False)
    B, T, H = wl['B'], wl['T'], wl['H']
    d = dominant_sweep(prof, steps, name, B, T, H)
    split = L.get_gemm_mode() == 1
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
    tfl = d['sweep_flops'] / (d['dom_ms'] * 1e-3) / 1e12
    fb = [int(w.item()) for w in (model.fallback_words() if hasattr(model, 'fallback_words') else [])]
    ms = dt / steps * 1e3
    res = {'ms_per_step': round(ms, 3), 'value': round(B / (ms * 1e-3), 1), 'unit': 'utterances/s', 'steps': steps, 'warmup': warmup,
           'final_loss': round(loss.item(), 6),
           'dominant_kernel': {'kernel': d['dom'], 'avg_launch_ms': round(d['dom_ms'], 4), 'tflops': round(tfl, 2),
                               'frac_of_mfma_peak': round(tfl / peak, 4), 'peak_tflops': round(peak, 1)},
           'gru_forward_path': (None if (not fb or 'gru_fwd_sweep' not in d['cats']) else ('fallback (co-schedule-tolerant sweeps redid the forward)' if any(fb) else
                                                     'exclusive fused two-layer launch' if L.load().dep_rnn_get_exclusive() else 'tolerant sweeps (dep_rnn_set_exclusive(0))')),
           'kernels_ms_per_step': {k: round(v[0] / steps, 4) for k, v in d['cats'].items()},
           'families': kernel_families(d['cats'], steps, name, B, T, H, split, ms),
           'step_traffic': {'survey_8d_bytes_per_step': SURVEY_8D_BYTES_PER_STEP[name], 'pmc_bytes_per_step': pmc_step_bytes(name)},
           'workload': '%s.%s train step, B=%d T=%d F=%d H=%d' % (WORKLOADS[name][0], WORKLOADS[name][1], B, T, wl['F'], H)}
    # labelled throughput mode (dep_set_gemm_mode(2): single bf16 products in the time-parallel contractions, the gate gradients read through the hi
    # rows of the sweep's PK image; fp32 storage of everything else, sweeps unchanged) -- own tolerance (tests/test_presplit_gpu.py), never the parity line
    if split:
        L.set_gemm_mode(2)
        try:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            L.profile_enable(True); L.profile_read()
            t2 = time.perf_counter()
            for _ in range(max(3, steps // 2)):
                lf = step()
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t2) / max(3, steps // 2) * 1e3
            pr2 = L.profile_read(); L.profile_enable(False)
            res['bf16_products'] = {'ms_per_step': round(ms2, 3), 'value': round(B / (ms2 * 1e-3), 1), 'unit': 'utterances/s', 'final_loss': round(lf.item(), 6),
                                    'kernels_ms_per_step': {k: round(v[0] / max(3, steps // 2), 4) for k, v in pr2.items() if v[1] > 0},
                                    'note': 'dep_set_gemm_mode(2): a_hi*b_hi only in the time-parallel GEMMs; NOT within the 1e-4 parity bar'}
        finally:
            L.set_gemm_mode(1)
    if res['step_traffic']['pmc_bytes_per_step']:
        res['step_traffic']['wasted_traffic_ratio'] = round(res['step_traffic']['pmc_bytes_per_step'] / SURVEY_8D_BYTES_PER_STEP[name], 2)
    del wl, step, model
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='audio_gru', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-workloads', action='store_true', help='skip extra.other_workloads (the other two BASELINE per-GPU workloads, a few steps each)')
    ap.add_argument('--profile-run', action='store_true', help='warm-up + timed steps only (rocprofv3 passes: every launch belongs to a train step)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process-group backend (gloo: launch check on CPU, or a dry run of the N-rank bench with the ranks sharing one GPU)')
    ap.add_argument('--launch-check', action='store_true', help='only exercise the N-rank launch + one all-reduce')
    args = ap.parse_args()

    world_env = int(os.environ.get('WORLD_SIZE', '0'))
    if args.gpus > 1 and world_env == 0:
        relaunch_under_torchrun(args.gpus, sys.argv[1:])          # does not return
    if args.launch_check:
        return launch_check(args)

    from icassp2022_depression_amd import _lib as L, nn, parallel
    import importlib
    comm_kind = 'none'
    if world_env > 1:
        # init_from_env builds the C-ABI's own RCCL communicator too (per-layer ranges overlapped with the backward pass); if it
        # cannot be built on EVERY rank all ranks agree to fall back -- visibly, in `config.backend` -- to torch.distributed
        parallel.init_from_env(args.backend)        # 'gloo': dry run of the multi-rank logic with the ranks sharing the visible GPUs
        comm_kind = {'rccl-native': 'rccl via dep_comm_* (layer ranges overlapped with backward)',
                     'torch.distributed': 'rccl via torch.distributed (single bucket after backward): ' + str(parallel._native.get('why'))
                     }.get(parallel.transport(), parallel.transport())
    rank, world = parallel.rank(), parallel.world_size()
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the process group has {world} ranks '
                         f'(WORLD_SIZE={os.environ.get("WORLD_SIZE")}): refusing to report an N={args.gpus} number')
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)

    modname, cls, B, T, F, H = WORKLOADS[args.workload]
    wl = build_workload(args.workload, dev, rank, world)
    mod, model, optimizer, criterion, cfg, step = wl['mod'], wl['model'], wl['optimizer'], wl['criterion'], wl['cfg'], wl['step']
    x, xa, xt = wl.get('x'), wl.get('xa'), wl.get('xt')

    for _ in range(args.warmup):
        step()
    parallel.barrier(); torch.cuda.synchronize()
    L.profile_enable(True); L.profile_read()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize(); parallel.barrier()
    dt = time.perf_counter() - t0
    prof = L.profile_read(); L.profile_enable(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    final_loss = loss.item()
    # which GRU forward produced the headline: the exclusive fused launch, or -- on a shared GPU -- its on-device fallback
    fbw = [int(w.item()) for w in (model.fallback_words() if hasattr(model, 'fallback_words') else [])]
    stw = [int(w.item()) for w in (model.status_words() if hasattr(model, 'status_words') else [])]
    PATHS = ['not a GRU stack', 'exclusive fused two-layer launch', 'tolerant sweeps (dep_rnn_set_exclusive(0))',
             'fallback (co-schedule-tolerant sweeps redid the forward)']
    my_path = 0 if not fbw else (3 if any(fbw) else (1 if L.load().dep_rnn_get_exclusive() else 2))
    fwd_path = None if not fbw else PATHS[my_path]
    # the line is rank 0's, the forward path / sweep status / transport are EVERY rank's -- a fallback or a raised
    # status word on rank k must show in the N-GPU line.  One small all-gather outside the timed region.
    per_rank = None
    if world > 1:
        import torch.distributed as dist
        mine = torch.tensor([my_path, max(stw) if stw else 0, 1 if parallel.transport() == 'rccl-native' else 0], dtype=torch.int64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if dist.get_backend() != 'nccl':
            mine = mine.cpu(); allr = [t.cpu() for t in allr]
        dist.all_gather(allr, mine)
        rows = [[int(v) for v in t.tolist()] for t in allr]
        per_rank = {'gru_forward_path_by_rank': [PATHS[r[0]] for r in rows], 'sweep_status_by_rank': [r[1] for r in rows],
                    'native_rccl_by_rank': [bool(r[2]) for r in rows]}
        if len({r[0] for r in rows}) > 1:
            fwd_path = 'MIXED over ranks: ' + '; '.join(f'rank {i}: {PATHS[r[0]]}' for i, r in enumerate(rows))
        elif fbw:
            fwd_path = PATHS[rows[0][0]] + f' (all {world} ranks)'
        if any(r[1] for r in rows):
            raise SystemExit(f'bench.py: a recurrent sweep raised its status word on some rank {[r[1] for r in rows]}: refusing to report the step')

    extras = not args.profile_run
    # exposed communication per rank: the same step with the gradient exchange switched off, timed like the
    # headline region; exposed = step - step_without_comm.  The replicas drift apart without the exchange: parameters and optimizer
    # moments are snapshotted and restored around the leg.
    exposed = None
    if world > 1 and extras:
        import torch.distributed as dist
        snap = model._flat.clone()
        st_snap = {k: (m.clone(), v.clone()) for k, (m, v) in optimizer._state.items()}
        step_snap = optimizer._step
        parallel.set_comm_enabled(False)
        try:
            for _ in range(2):
                step()
            torch.cuda.synchronize(); parallel.barrier()
            n_x = max(3, min(args.steps, 10))
            t4 = time.perf_counter()
            for _ in range(n_x):
                step()
            torch.cuda.synchronize()
            ms_nc = (time.perf_counter() - t4) / n_x * 1e3
        finally:
            parallel.set_comm_enabled(True)
            model._flat.copy_(snap)
            for k, (m, v) in st_snap.items():
                optimizer._state[k][0].copy_(m); optimizer._state[k][1].copy_(v)
            optimizer._step = step_snap
        parallel.barrier()
        mine = torch.tensor([ms_nc], device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if dist.get_backend() != 'nccl':
            mine = mine.cpu(); allr = [t.cpu() for t in allr]
        dist.all_gather(allr, mine)
        step_ms = dt / args.steps * 1e3
        exposed = {'step_without_comm_ms_by_rank': [round(float(t.item()), 4) for t in allr],
                   'exposed_comm_ms_by_rank': [round(step_ms - float(t.item()), 4) for t in allr],
                   'note': 'step_without_comm: the same train step with the gradient exchange switched off (local gradients), timed on every rank; '
                           'exposed = ms_per_step (max over ranks, exchange on) - that'}
    # gradient exchange alone (every rank; gathered on rank 0): the same ranges the step reduces, nothing beside them
    comm_alone = None
    if world > 1 and extras:
        import torch.distributed as dist
        in_call, post = model.sync_plan()
        model._grad_ready = True
        torch.cuda.synchronize(); parallel.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            parallel.reduce_zero_contribution(model, in_call, post) if parallel.native_comm() is not None else parallel.all_reduce_grads(model)
        e1.record(); torch.cuda.synchronize()
        mine = torch.tensor([e0.elapsed_time(e1) / 10], device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if dist.get_backend() != 'nccl':
            mine = mine.cpu(); allr = [t.cpu() for t in allr]
        dist.all_gather(allr, mine)
        comm_alone = [round(float(t.item()), 4) for t in allr]

    # forward-only (evaluate) rate, outside the headline region: eval-mode kernels, nothing saved for a backward
    eval_ms = None
    if extras:
        model.eval()
        eval_fn = (lambda: model.pretrained_feature((xa, xt))) if args.workload == 'fusion' else (lambda: model(x))
        eval_fn(); torch.cuda.synchronize()
        n_eval = max(3, min(args.steps, 10))
        t1 = time.perf_counter()
        for _ in range(n_eval):
            eval_fn()
        torch.cuda.synchronize()
        eval_ms = (time.perf_counter() - t1) / n_eval * 1e3
        model.train()

    # the same step with EXACT fp32 products everywhere (DEP_GEMM_MODE=f32): the headline runs the 3-term bf16 split
    split_mode = L.get_gemm_mode() == 1
    f32_exact = None
    if extras and split_mode and rank == 0 and world == 1:
        L.set_gemm_mode(0)
        try:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            n_f = max(3, min(args.steps, 8))
            t2 = time.perf_counter()
            for _ in range(n_f):
                lf = step()
            torch.cuda.synchronize()
            ms_f = (time.perf_counter() - t2) / n_f * 1e3
            f32_exact = {'ms_per_step': round(ms_f, 3), 'value': round(B / (ms_f * 1e-3), 1), 'unit': 'utterances/s',
                         'final_loss': round(lf.item(), 6), 'note': 'dep_set_gemm_mode(0): fp32 MFMA for every contraction and sweep'}
        finally:
            L.set_gemm_mode(1)
    # ... and with single bf16 products in the large contractions (dep_set_gemm_mode(2), BASELINE configs[1]'s "bf16"): a labelled
    # throughput mode with its own tolerance (tests: 5e-3 on outputs), never the headline
    bf16_products = None
    if extras and split_mode and rank == 0 and world == 1:
        L.set_gemm_mode(2)
        try:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            n_f = max(3, min(args.steps, 10))
            t2 = time.perf_counter()
            for _ in range(n_f):
                lf = step()
            torch.cuda.synchronize()
            ms_f = (time.perf_counter() - t2) / n_f * 1e3
            bf16_products = {'ms_per_step': round(ms_f, 3), 'value': round(B / (ms_f * 1e-3), 1), 'unit': 'utterances/s',
                             'final_loss': round(lf.item(), 6),
                             'note': 'dep_set_gemm_mode(2): a_hi*b_hi only in the time-parallel GEMMs (fp32 storage and accumulation, '
                                     'recurrent sweeps unchanged); NOT within the 1e-4 parity bar'}
        finally:
            L.set_gemm_mode(1)

    # ... and with bf16 STORAGE on top (dep_set_gemm_mode(3)): hidden sequences, hn and gate gradients as bf16 in HBM, saved gates 16-bit
    # fixed point, state / accumulation fp32 -- BASELINE configs[1]'s "bf16" taken literally; own tolerance (tests: 1e-2 on gradients)
    bf16_storage = None
    if extras and split_mode and rank == 0 and world == 1 and args.workload == 'audio_gru':
        L.set_gemm_mode(3)
        try:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            n_f = max(3, min(args.steps, 10))
            L.profile_enable(True); L.profile_read()
            t2 = time.perf_counter()
            for _ in range(n_f):
                lf = step()
            torch.cuda.synchronize()
            ms_f = (time.perf_counter() - t2) / n_f * 1e3
            pr3 = L.profile_read(); L.profile_enable(False)
            bf16_storage = {'ms_per_step': round(ms_f, 3), 'value': round(B / (ms_f * 1e-3), 1), 'unit': 'utterances/s',
                            'final_loss': round(lf.item(), 6),
                            'kernels_ms_per_step': {k: round(v[0] / n_f, 4) for k, v in pr3.items() if v[1] > 0},
                            'note': 'dep_set_gemm_mode(3): single bf16 products AND bf16 storage of y / hn / gate gradients (16-bit fixed-point '
                                    'saved gates), fp32 state, accumulation and recurrence; NOT within the 1e-4 parity bar'}
        finally:
            L.set_gemm_mode(1)

    # train() as a user calls it: the script's own epoch loop over a host-resident corpus of 4 mini-batches (features uploaded
    # once and gathered on the device, labels / loss.item() / accuracy count per step as in the reference)
    train_e2e = None
    if extras and rank == 0 and world == 1 and args.workload != 'fusion':
        import contextlib, io
        import numpy as np
        nb = 6
        rng = np.random.default_rng(7)
        feats = rng.standard_normal((nb * B, T, F), dtype=np.float32)
        targs = rng.integers(0, 2, nb * B)
        which = 'audio' if args.workload == 'audio_gru' else 'text'
        saved_cfg = dict(mod.config)
        mod.config.update(cfg); mod.config['batch_size'] = B
        setattr(mod, which + '_features', feats); setattr(mod, which + '_targets', targs)
        mod.model, mod.optimizer, mod.criterion = model, optimizer, criterion
        idx = list(range(nb * B))
        with contextlib.redirect_stdout(io.StringIO()):
            mod.train(1, idx)                                   # includes the one-time upload of the corpus
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            mod.train(2, idx)
            torch.cuda.synchronize()
        ms_e = (time.perf_counter() - t3) / nb * 1e3
        train_e2e = {'ms_per_minibatch': round(ms_e, 3), 'value': round(B / (ms_e * 1e-3), 1), 'unit': 'utterances/s',
                     'vs_bench_step': round(ms_e / (dt / args.steps * 1e3), 3),
                     'note': f'{modname}.train() over {nb} host mini-batches of {B}: features resident in HBM after one upload, '
                             'labels uploaded once per epoch, loss sum and accuracy count read once per epoch'}
        mod.config.clear(); mod.config.update(saved_cfg)
        from icassp2022_depression_amd import _common
        _common.invalidate_device_features()

    # BASELINE configs[2] (text BiLSTM) and [3] (late fusion) at their per-GPU shapes, driver-timed in this same invocation
    other = None
    if extras and rank == 0 and world == 1 and not args.no_other_workloads:
        other = {}
        for name in sorted(WORKLOADS):
            if name != args.workload:
                other[name] = time_other_workload(name, dev, L)

    if world > 1:
        parallel.barrier()
        parallel.destroy_native_comm()                          # every rank tears its communicator down
    if rank != 0:
        return
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    # ---- roofline of the dominant kernel ------------------------------------------------
    cats = {k: v for k, v in prof.items() if v[1] > 0}
    sweeps = {k: v for k, v in cats.items() if 'sweep' in k}
    dom = max(sweeps, key=lambda k: sweeps[k][0])
    dom_ms = sweeps[dom][0] / sweeps[dom][1]
    H_model = H
    if dom.startswith('lstm'):
        G, dirs, H = 4, 2, 128                                # the text encoder's hidden size in every workload
    else:
        G, dirs = 3, 1
    launches_per_step = sweeps[dom][1] / args.steps          # 2 for per-layer sweeps, 1 for a launch that carries both layers
    layers_per_launch = 2.0 / launches_per_step
    # SURVEY 8(d) algorithmic work of ONE launch of the dominant sweep: recurrent flops 2 * T * (G H) * H per utterance, layer
    # and direction; COMPULSORY HBM bytes per (utterance, step, layer, direction): the hidden sequence once -- written by the
    # forward sweep, read by the backward sweep (8d: "hidden sequences written once in fwd and read once in bwd, nothing else,
    # gates recomputed") = 4H bytes in fp32.  The DESIGN bytes count what this implementation chose to move instead (saved
    # gate tensors, the materialised input projection and its gradient), DESIGN.md section 4.
    sweep_flops = 2.0 * B * T * (G * H) * H * dirs * layers_per_launch
    fused_gru = dom.startswith('gru') and layers_per_launch > 1.5
    if fused_gru:
        # a fused two-layer GRU launch carries a THIRD product of the same size beside the two recurrences: layer 1's input projection
        # W_ih(l1) dropout(h0) in the forward, the gradient entering layer 0, dgi(l1) W_ih(l1), in the backward (the former projection / dX GEMMs)
        sweep_flops *= 1.5
    fusion = args.workload == 'fusion'
    design_per_ut = ({'gru_fwd_sweep': 18 * H, 'lstm_fwd_sweep': 44 * H} if fusion else
                     {'gru_fwd_sweep': 34 * H, 'gru_bwd_sweep': 32 * H, 'lstm_fwd_sweep': 84 * H, 'lstm_bwd_sweep': 88 * H})[dom]      # (round 4: GRU saved gates r, z, n are 2 bytes each: backward 38H -> 32H)
    compulsory_per_ut = 4 * H * dirs
    design_bytes = float(design_per_ut) * B * T * layers_per_launch
    if dom == 'gru_fwd_sweep' and layers_per_launch > 1.5:
        # the fused two-layer forward: layer 0's input projection in (12H), both layers' h (8H) and saved gates (2 x (6H + 4H)) out, the
        # dropped copy of layer 0's h (4H); layer 1's projection never exists in HBM
        design_bytes = float(44 * H) * B * T              # (round 4: 16-bit r, z, n: 56H -> 44H)
    if dom == 'gru_bwd_sweep' and layers_per_launch > 1.5:
        # the fused two-layer backward (round 5): per layer r, z, n (2 bytes each) + hn + h_{t-1} in (14H), the 4H-wide PK gate gradients out (16H);
        # layer 1's dX / layer 0's dy never exist in HBM
        design_bytes = float(60 * H) * B * T
    compulsory_bytes = float(compulsory_per_ut) * B * T * layers_per_launch
    split = L.get_gemm_mode() == 1                           # the cluster sweeps follow the GEMM precision mode
    mfma_peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
    sec = dom_ms * 1e-3
    tflops = sweep_flops / sec / 1e12
    frac_mfma = tflops / mfma_peak
    # serial floor: the dependent recurrent steps of this launch at 1 us per hand-off
    serial_steps = T + (layers_per_launch - 1)
    serial_floor_ms = serial_steps * 1e-3
    traffic, traffic_note = None, 'no PMC record for this workload / kernel'
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        rec = json.load(open(tpath))
        stamp = open(os.path.join(ROOT, 'icassp2022-depression_amd', 'libdep_rnn.so.stamp')).read().strip()
        ent = rec.get('workloads', {}).get(args.workload, {}).get(dom)
        if ent is not None and rec.get('kernel_digest') == stamp:
            traffic, traffic_note = ent, f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at commit {rec.get('commit')}, same kernel digest"
        elif ent is not None:
            traffic_note = 'PMC record is from other kernel sources (digest mismatch): refused'
    except Exception:
        pass
    step_traffic = None
    try:
        step_traffic = rec.get('step_bytes', {}).get(args.workload) if rec.get('kernel_digest') == stamp else None
    except Exception:
        pass
    # which roof is nearer, from SURVEY 8(d)'s algorithmic figures (flops; compulsory bytes): derived, not assumed.  Neither
    # binds: the sweep is limited by the serial hand-off latency of its T dependent steps (serial_floor below).
    frac_hbm_alg = compulsory_bytes / sec / 1e9 / PEAK_HBM_GBS
    bound = 'mfma' if frac_mfma >= frac_hbm_alg else 'hbm'
    roofline = {'bound': bound, 'kernel': dom,
                'achieved': round(tflops, 3) if bound == 'mfma' else round(compulsory_bytes / sec / 1e9, 1),
                'peak': round(mfma_peak, 1) if bound == 'mfma' else PEAK_HBM_GBS, 'unit': 'TFLOP/s' if bound == 'mfma' else 'GB/s',
                'frac': round(frac_mfma if bound == 'mfma' else frac_hbm_alg, 4), 'traffic': traffic, 'traffic_note': traffic_note,
                'limiter': 'serial hand-off latency of the recurrent steps (see serial_floor); sustained bf16 MFMA rate under the '
                           'package power cap is 1.6-1.7 PFLOP/s (profiles/r03_micro_mfma_rate.txt), the guide peak is kept as `peak`',
                'step_traffic': {'pmc_bytes_per_step': step_traffic, 'survey_8d_bytes_per_step': SURVEY_8D_BYTES_PER_STEP[args.workload],
                                 'note': 'sum over all kernels of (2*FETCH_SIZE + WRITE_SIZE) per train step, rocprofv3 --pmc passes'},
                'mfma_pipe': 'bf16 x3 split (peak = bf16 dense / 3)' if split else 'fp32',
                'flops_per_launch': sweep_flops, 'avg_launch_ms': round(dom_ms, 4), 'launches_per_step': launches_per_step,
                'achieved_mfma': {'tflops': round(tflops, 3), 'frac': round(frac_mfma, 4)},
                'achieved_hbm': {'compulsory_bytes_per_launch': compulsory_bytes,
                                 'compulsory_gbs': round(compulsory_bytes / sec / 1e9, 1),
                                 'compulsory_frac': round(compulsory_bytes / sec / 1e9 / PEAK_HBM_GBS, 4),
                                 'design_bytes_per_launch': design_bytes,
                                 'design_gbs': round(design_bytes / sec / 1e9, 1),
                                 'design_frac': round(design_bytes / sec / 1e9 / PEAK_HBM_GBS, 4),
                                 'peak_gbs': PEAK_HBM_GBS},
                'serial_floor': {'dependent_steps_per_launch': serial_steps, 'us_per_step': round(dom_ms * 1e3 / serial_steps, 3),
                                 'floor_ms_at_1us': round(serial_floor_ms, 4), 'serial_floor_frac': round(serial_floor_ms / dom_ms, 4)},
                'kernels_ms_per_step': {k: round(v[0] / args.steps, 4) for k, v in cats.items()}}
    train_flops_per_utt = {'audio_gru': 1.4156e9, 'text_bilstm': 2.831e9,           # SURVEY 8(d); fusion = the two forwards
                           'fusion': (1.4156e9 + 2.831e9) / 3.0}[args.workload]
    step_tflops = train_flops_per_utt * value / 1e12 / world
    roofline['step_tflops_fp32_equiv'] = round(step_tflops, 2)
    roofline['families'] = kernel_families(cats, args.steps, args.workload, B, T, H, split, ms_per_step)
    if step_traffic:
        roofline['step_traffic']['wasted_traffic_ratio'] = round(step_traffic / SURVEY_8D_BYTES_PER_STEP[args.workload], 2)

    out = {'metric': {'audio_gru': 'utterances/sec (train step) for GRU-256 on (B,T,F)=(512,300,256)',
                      'text_bilstm': 'utterances/sec (train step) for BiLSTM-128x2 on (B,T,F)=(512,300,1024)',
                      'fusion': 'utterance pairs/sec (late-fusion train step: frozen GRU-256 + BiLSTM-128x2 encoders, '
                                'linear head) on (B,T,Fa,Ft)=(512,300,256,1024)'}[args.workload],
           'value': round(value, 1), 'unit': 'utterances/s', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f32', 'precision_mode': 'bf16x3-split' if split_mode else 'f32', 'data': 'synthetic',
           'config': {'workload': (f'{modname}.{cls} train step, B={B}/GPU T={T} F={F} H={H_model} L=2 dropout={cfg["dropout"]} '
                                   + ('Adam, MyLoss (split-weight CE)' if fusion else 'AdamW, CE-on-softmax')),
                      'global_batch': B * world, 'parallelism': f'dp{world}', 'ranks': world,
                      'backend': comm_kind},
           'final_loss': round(final_loss, 6),
           'gru_forward_path': fwd_path,
           'roofline': roofline}
    if world > 1:
        # top level, not only config.backend: which transport carried the gradients, and why if it is not the C-ABI's own communicator
        tr = parallel.transport()
        out['comm_transport'] = tr
        out['comm_transport_is_native_rccl'] = bool(per_rank and all(per_rank['native_rccl_by_rank']))
        if tr != 'rccl-native':
            out['comm_transport_reason'] = ('gloo dry run (ranks share the visible GPUs)' if args.backend == 'gloo'
                                            else str(parallel._native.get('why') or 'native communicator not built on every rank'))
        out['ranks'] = per_rank
    if eval_ms is not None:
        out['eval_forward'] = {'value': round(B / (eval_ms * 1e-3), 1), 'unit': 'utterances/s', 'ms_per_batch': round(eval_ms, 3),
                               'note': 'forward only (evaluate), rank 0, outside the headline region'}
    out['extra'] = {'f32_exact': f32_exact, 'bf16_products': bf16_products, 'bf16_storage': bf16_storage, 'train_e2e': train_e2e, 'comm_alone_ms_per_step_by_rank': comm_alone, 'exposed_comm': exposed,
                    'other_workloads': other,
                    'precision_note': 'storage, state, accumulation and elementwise math fp32; products of the large contractions '
                                      'and of the recurrent sweeps: 3-term bf16 split on the bf16 matrix cores (DEP_GEMM_MODE=f32 = exact)'}

    if world == 1 and not args.no_cpu_baseline and extras:
        import platform
        from oracle import torch_cpu_baseline as tb
        threads = usable_cores()
        kind = {'audio_gru': 'audio', 'text_bilstm': 'text', 'fusion': 'fusion'}[args.workload]
        # bounded sample: median of 5 full-size steps after 2 warm-ups (audio ~2.6 s/step, text ~3.3 s, fusion forward ~1.5 s)
        ups, sec_step, nthr = tb.time_train_step(kind, B, T, F, H_model, steps=5, warmup=2, threads=threads, lr=cfg['learning_rate'])
        cpu_model = ''
        try:
            cpu_model = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
        except Exception:
            cpu_model = platform.processor()
        out['cpu_baseline'] = {'value': round(ups, 1), 'unit': 'utterances/s', 'cores': nthr, 'kind': 'port',
                               'cpu_model': cpu_model, 'torch': torch.__version__,
                               'sample': f'same train step (stock torch.nn CPU, fp32), B={B} T={T}, median of 5 steps after 2 warm-ups, '
                                         f'{sec_step:.2f} s/step'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
