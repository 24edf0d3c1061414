#!/usr/bin/env python3
"""Headline benchmark: utterances/s of one full train step (forward + CE-on-softmax loss + backward +
AdamW, dropout on) of the audio GRU-256 x2 classifier on synthetic (B,T,F) = (512,300,256) per GPU
(BASELINE.json configs[1]), weak-scaled over N GPUs with one RCCL gradient all-reduce per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload audio_gru|text_bilstm|fusion]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the persistent recurrent sweep with the largest
total time): its algorithmic bytes and flops per launch over its mean launch duration, measured
with HIP events on the launch stream during the timed region, against the HBM peak and against the peak of the matrix
pipe the sweep runs on (fp32 MFMA in exact mode; the bf16 pipe / 3 for the 3-term split).  The roof the kernel sits
closer to is reported as `bound`; both fractions are kept in the object.  `cpu_baseline` (N=1 only) times the same train step on the host
cores with stock torch.nn (oracle/torch_cpu_baseline.py, validated against the reference's fixtures).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='audio_gru', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from icassp2022_depression_amd import _lib as L, nn, parallel
    import importlib
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if world_env > 1:
        parallel.init_from_env('nccl')
    rank, world = parallel.rank(), parallel.world_size()
    if world != args.gpus and rank == 0:
        print(f'warning: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(dev)

    modname, cls, B, T, F, H = WORKLOADS[args.workload]
    mod = importlib.import_module('icassp2022_depression_amd.' + modname)
    torch.manual_seed(0)
    g = torch.Generator(device='cpu'); g.manual_seed(1234 + rank)
    y = torch.randint(0, 2, (B,), generator=g).to(dev)
    if args.workload == 'fusion':
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
            model(torch.cat((tf, af), dim=1))
            loss = criterion(tf, af, y, model)
            loss.backward()
            optimizer.step()
            return loss
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
            out = model(x)
            loss = criterion(out, y)
            loss.backward()                                   # includes the RCCL all-reduce of the grad bucket
            optimizer.step()
            return loss

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
    sweep_flops = 2.0 * B * T * (G * H) * H * dirs           # one layer sweep launch (fwd or bwd): gates x H MACs
    # algorithmic HBM bytes per (utterance, time step) of one sweep launch, averaged over the two layers:
    #   GRU fwd : gi 12H + y 4H + saved r,z,n,hn 16H + dropped y 4H (layer 0 only)          = 34H
    #   GRU bwd : saved 16H + h_{t-1} 4H + dy 4H (layer 0 only) + dgi 12H + dghn 4H         = 38H
    #   LSTM fwd: (gi 16H + y 4H + gates 16H + c 4H) x 2 dirs + dropped y 8H (layer 0 only) = 84H
    #   LSTM bwd: (gates 16H + c_t 4H + c_{t-1} 4H + dy 4H + dgi 16H) x 2 dirs              = 88H
    #   fusion (encoders forward only, nothing saved): GRU gi 12H + y 4H + dropped y 4H/2 = 18H ; LSTM (16H + 4H) x 2 + 4H = 44H
    per_ut = ({'gru_fwd_sweep': 18 * H, 'lstm_fwd_sweep': 44 * H} if args.workload == 'fusion' else
              {'gru_fwd_sweep': 34 * H, 'gru_bwd_sweep': 38 * H, 'lstm_fwd_sweep': 84 * H, 'lstm_bwd_sweep': 88 * H})[dom]
    sweep_bytes = float(per_ut) * B * T
    split = L.get_gemm_mode() == 1                           # the cluster sweeps follow the GEMM precision mode
    mfma_peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
    tflops = sweep_flops / (dom_ms * 1e-3) / 1e12
    gbs = sweep_bytes / (dom_ms * 1e-3) / 1e9
    frac_mfma, frac_hbm = tflops / mfma_peak, gbs / PEAK_HBM_GBS
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload, {}).get(dom)
        except Exception:
            traffic = None
    if frac_hbm >= frac_mfma:
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': round(frac_hbm, 4), 'traffic': traffic}
    else:
        roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': round(tflops, 3), 'peak': round(mfma_peak, 1),
                    'unit': 'TFLOP/s', 'frac': round(frac_mfma, 4), 'traffic': traffic}
    roofline.update({'bytes_per_launch': sweep_bytes, 'flops_per_launch': sweep_flops, 'avg_launch_ms': round(dom_ms, 4),
                     'frac_hbm': round(frac_hbm, 4), 'frac_mfma': round(frac_mfma, 4),
                     'mfma_pipe': 'bf16 x3 split' if split else 'fp32',
                     'kernels_ms_per_step': {k: round(v[0] / args.steps, 4) for k, v in cats.items()}})
    train_flops_per_utt = {'audio_gru': 1.4156e9, 'text_bilstm': 2.831e9,           # SURVEY 8(d); fusion = the two forwards
                           'fusion': (1.4156e9 + 2.831e9) / 3.0}[args.workload]
    step_tflops = train_flops_per_utt * value / 1e12 / world
    roofline['step_tflops_fp32_equiv'] = round(step_tflops, 2)

    out = {'metric': {'audio_gru': 'utterances/sec (train step) for GRU-256 on (B,T,F)=(512,300,256)',
                      'text_bilstm': 'utterances/sec (train step) for BiLSTM-128x2 on (B,T,F)=(512,300,1024)',
                      'fusion': 'utterance pairs/sec (late-fusion train step: frozen GRU-256 + BiLSTM-128x2 encoders, '
                                'linear head) on (B,T,Fa,Ft)=(512,300,256,1024)'}[args.workload],
           'value': round(value, 1), 'unit': 'utterances/s', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': (f'{modname}.{cls} train step, B={B}/GPU T={T} F={F} H={H_model} L=2 dropout={cfg["dropout"]} '
                                   + ('Adam, MyLoss (split-weight CE)' if args.workload == 'fusion' else 'AdamW, CE-on-softmax')),
                      'global_batch': B * world, 'parallelism': f'dp{world}'},
           'final_loss': round(final_loss, 6), 'roofline': roofline}

    if world == 1 and not args.no_cpu_baseline and args.workload != 'fusion':      # (no CPU port of the fusion step is kept)
        from oracle import torch_cpu_baseline as tb
        threads = usable_cores()
        kind = 'audio' if args.workload == 'audio_gru' else 'text'
        ups, sec, nthr = tb.time_train_step(kind, B, T, F, H, steps=3, warmup=1, threads=threads, lr=cfg['learning_rate'])
        out['cpu_baseline'] = {'value': round(ups, 1), 'unit': 'utterances/s', 'cores': nthr, 'kind': 'port',
                               'sample': f'same train step (stock torch.nn CPU, fp32), B={B} T={T}, median of 3 steps after 1 warm-up, '
                                         f'{sec:.2f} s/step'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
