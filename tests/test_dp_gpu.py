"""Data parallelism on real device buffers: two ranks (both on cuda:0, gloo rendezvous on 127.0.0.1) each train on
their slice of every mini-batch through audio_gru_whole.train(); the result must equal the single-process run."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def _run(rank, world, port, q, backend='gloo'):
    # backend 'nccl': one rank per GPU over the C-ABI's RCCL communicator (dep_comm_*); 'gloo': the ranks share cuda:0
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank) if backend == 'nccl' else '0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    sys.path.insert(0, ROOT)
    from icassp2022_depression_amd import audio_gru_whole as m, nn, parallel
    if backend == 'nccl':
        torch.cuda.set_device(rank)
    if world > 1:
        parallel.init_from_env(backend)
        if backend == 'nccl':
            assert parallel.transport() == 'rccl-native', parallel._native.get('why')
    g = load_golden('audio_clf_train_eval')
    N, T, F, H = [int(v) for v in g['shape']]
    m.config.update(embedding_size=F, hidden_dims=H, dropout=0.0, batch_size=5, learning_rate=float(g['lr']))
    m.audio_features = g['feats']; m.audio_targets = g['targs']
    m.model = m.AudioBiLSTM(m.config, seed=0)
    m.model.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()})
    m.optimizer = nn.AdamW(m.get_param_group(m.model), lr=m.config['learning_rate'])
    m.criterion = nn.CrossEntropyLoss()
    idx = list(range(17))                     # batches of 5,5,5,2 -> shards 3+2, 3+2, 3+2, 1+1
    with contextlib.redirect_stdout(io.StringIO()):
        m.train(1, idx); m.train(2, idx)
    if rank == 0:
        q.put(({k: v.cpu().numpy() for k, v in m.model.state_dict().items()}, int(m.train_acc)))
    if world > 1:
        parallel.barrier()
        parallel.destroy_native_comm()
        import torch.distributed as dist
        dist.destroy_process_group()


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
def test_two_rank_training_equals_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    res = {}
    for world in (1, 2, 3):                   # world 3: the last mini-batch (2 rows) leaves rank 2 an EMPTY shard
        q = ctx.Queue()
        port = 29700 + os.getpid() % 1000 + world
        procs = [ctx.Process(target=_run, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = q.get(timeout=240)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    sd1, acc1 = res[1]
    for world in (2, 3):
        sdw, accw = res[world]
        assert acc1 == accw
        for k in sd1:
            assert np.abs(sd1[k] - sdw[k]).max() < 2e-6, (world, k)


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason='needs two GPUs (a multi-GPU driver box)')
def test_native_rccl_multi_gpu_training_equals_single_process():
    """RCCL with N > 1 ranks: one process per GPU, backend nccl, the gradient ranges reduced through the
    C-ABI's own communicator (dep_comm_* over xGMI, the top layer's range overlapped with layer 0's backward) -- two epochs of
    audio_gru_whole.train() must leave the parameters of the single-process run.  Runs by itself wherever >= 2 GPUs are visible."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    res = {}
    worlds = (1, 2) if torch.cuda.device_count() < 4 else (1, 2, 4)
    for world in worlds:
        q = ctx.Queue()
        port = 28300 + os.getpid() % 1000 + world
        procs = [ctx.Process(target=_run, args=(r, world, port, q, 'nccl' if world > 1 else 'gloo')) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = q.get(timeout=300)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    sd1, acc1 = res[1]
    for world in worlds[1:]:
        sdw, accw = res[world]
        assert acc1 == accw
        for k in sd1:
            assert np.abs(sd1[k] - sdw[k]).max() < 2e-6, (world, k)


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
def test_native_rccl_comm_overlapped_backward_single_rank():
    """The C-ABI's RCCL entry points on the one GPU this box has: a one-rank communicator (dep_comm_unique_id / init /
    allreduce / allreduce_ranges / destroy through librccl) and dep_rnn_backward_overlapped must leave exactly the gradients
    of the plain backward (a one-rank SUM is the identity), for the audio (LayerNorm fold: layer 0 reduced after the
    backward) and the text (every layer reduced in-call) models.  Multi-rank numerics are covered by the gloo tests; this
    checks the native plumbing: symbol resolution, stream / event ordering, range arithmetic."""
    from icassp2022_depression_amd import audio_gru_whole as ma, text_bilstm_whole as mt, nn, parallel
    rng = np.random.default_rng(0)
    grads = {}
    for native in (False, True):
        if native:
            assert parallel.init_native_comm(force_single=True) is not None
        for name, mod, cls, F, H in (('audio', ma, 'AudioBiLSTM', 24, 128), ('text', mt, 'TextBiLSTM', 40, 128)):
            cfg = dict(mod.config); cfg.update(embedding_size=F, hidden_dims=H, dropout=0.0)
            model = getattr(mod, cls)(cfg, seed=3)
            x = np.random.default_rng(5).standard_normal((9, 11, F)).astype(np.float32)
            y = np.random.default_rng(6).integers(0, 2, 9)
            model.train()
            loss = nn.CrossEntropyLoss()(model(x), y)
            loss.backward()
            torch.cuda.synchronize()
            in_call, post = model.sync_plan()
            parallel.layer_buckets(list(in_call.values()) + post, model._n_live)          # the plan tiles the live bucket
            grads[(name, native)] = model.live_grad_bucket().clone()
    parallel.destroy_native_comm()
    for name in ('audio', 'text'):
        assert torch.equal(grads[(name, False)], grads[(name, True)]), name


def test_no_collective_is_enqueued_in_front_of_a_launch_that_needs_every_cu():
    """VERDICT r5 item 8.  The fused GRU forward AND (since round 5) the fused GRU backward fill every CU (12 x 168 VGPRs, 162 KB LDS): an RCCL
    kernel resident on a CU when they dispatch would break the co-residency their hand-off relies on (the forward then leaves through the
    on-device fallback, status 5 would be the backward's).  DESIGN section 6 argues that by ENQUEUE ORDER no collective can be running then;
    this holds that argument against the library's enqueue-order log (dep_order_log_*) on the native RCCL path with a one-rank communicator,
    three cfg2-shaped train steps:
      * every collective of a step is enqueued AFTER that step's fused backward launch (the in-call ranges behind their layer's weight-gradient
        GEMMs, the rest in finish_grad_sync) -- none between `backward begin` and the fused launch;
      * after the step's last collective the compute stream JOINS the communication stream before the optimizer step, so everything the next
        step's fused forward could meet has completed: between a step's join and the next fused forward no collective is enqueued."""
    from icassp2022_depression_amd import audio_gru_whole as ma, nn, parallel, _lib as L
    assert parallel.init_native_comm(force_single=True) is not None
    try:
        cfg = dict(ma.config); cfg.update(embedding_size=256, hidden_dims=256, dropout=0.5)
        model = ma.AudioBiLSTM(cfg, seed=3); model.train()
        opt = nn.AdamW(ma.get_param_group(model), lr=1e-4)
        crit = nn.CrossEntropyLoss()
        x = torch.randn(64, 40, 256, device='cuda'); y = np.random.default_rng(6).integers(0, 2, 64)
        parallel.set_global_count(64)
        def step():
            opt.zero_grad(); loss = crit(model(x), y); loss.backward(); opt.step()
        step(); torch.cuda.synchronize()
        L.order_log_enable(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        log = L.order_log_read(reset=True); L.order_log_enable(False)
    finally:
        parallel.destroy_native_comm()
    kinds = []
    for e in log:
        if e.startswith('K ') and 'gru2_fwd_fused' in e: kinds.append('FWD')
        elif e.startswith('K ') and 'gru2_bwd_fused' in e: kinds.append('BWD')
        elif e.startswith('C '): kinds.append('C')
        elif e.startswith('N join'): kinds.append('JOIN')
        elif e.startswith('N backward begin'): kinds.append('BB')
        elif e.startswith('N optimizer step'): kinds.append('OPT')
    assert kinds.count('FWD') == 3 and kinds.count('BWD') == 3 and kinds.count('C') >= 3, (kinds, log[:40])     # the exclusive launches ran; collectives were enqueued
    state = 'clean'                 # 'clean': no collective outstanding that the compute stream has not joined
    for k in kinds:
        if k == 'C':
            state = 'outstanding'
        elif k == 'JOIN':
            state = 'clean'
        elif k in ('FWD', 'BWD'):
            assert state == 'clean', ('a collective is enqueued and not joined in front of', k, kinds)
        elif k == 'OPT':
            assert state == 'clean', ('the optimizer step is enqueued before the gradient collectives were joined', kinds)
    # ... and structurally: BB -> BWD -> C+ -> JOIN -> OPT in every step
    seq = [k for k in kinds if k != 'FWD']
    per_step = ''.join({'BB': 'b', 'BWD': 'W', 'C': 'c', 'JOIN': 'j', 'OPT': 'o'}[k] for k in seq)
    import re
    assert re.fullmatch(r'(bWc+jo){3}', per_step), per_step


# ------------------------------------------------------------------------------------------- the other train() loops
def _run_generic(case, rank, world, port, q, dropout):
    """One rank of `case`'s train() on the reference-made train/eval fixture of that script."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    sys.path.insert(0, ROOT)
    import importlib
    from icassp2022_depression_amd import nn, parallel
    torch.manual_seed(2024)                   # the dropout keys derive from torch.initial_seed(), the step counter and the rank
    if world > 1:
        parallel.init_from_env('gloo')
    out = {}
    if case == 'text_clf':
        m = importlib.import_module('icassp2022_depression_amd.text_bilstm_whole')
        g = load_golden('text_clf_train_eval')
        N, T, F, H = [int(v) for v in g['shape']]
        m.config.update(embedding_size=F, hidden_dims=H, dropout=dropout, batch_size=5, learning_rate=float(g['lr']))
        m.text_features = g['feats']; m.text_targets = g['targs']
        m.model = m.TextBiLSTM(m.config, seed=0)
        m.model.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()})
        m.optimizer = nn.AdamW(m.get_param_group(m.model), lr=m.config['learning_rate']); m.criterion = nn.CrossEntropyLoss()
        idx = list(range(12))                 # 5, 5, 2: world 3 leaves rank 2 an empty shard of the last mini-batch
        with contextlib.redirect_stdout(io.StringIO()):
            m.train(1, idx); m.train(2, idx)
        out = dict(acc=int(m.train_acc))
        model = m.model
    elif case == 'audio_reg':
        m = importlib.import_module('icassp2022_depression_amd.audio_bilstm_perm')
        g = load_golden('audio_reg_train_eval')
        N, T, F, H = [int(v) for v in g['shape']]
        m.config.update(embedding_size=F, hidden_dims=H, dropout=dropout, batch_size=5, learning_rate=float(g['lr']))
        m.audio_features = g['feats']; m.audio_targets = g['targs']
        m.model = m.AudioBiLSTM(m.config, seed=0)
        m.model.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()})
        m.optimizer = nn.Adam(m.model.parameters(), lr=m.config['learning_rate']); m.criterion = nn.L1Loss()
        m.train_dep_idxs = [0, 1, 2, 3, 4]; m.train_non_idxs = [5, 6, 7, 8, 9, 10, 11]        # 12 rows: 5, 5, 2
        with contextlib.redirect_stdout(io.StringIO()):
            mae1 = m.train(1); mae2 = m.train(2)
        out = dict(mae=(float(mae1), float(mae2)))                # predictions assembled over the ranks (hi - lo floats per step)
        model = m.model
    else:
        m = importlib.import_module('icassp2022_depression_amd.fuse_net_whole')
        g = load_golden('fuse_clf')
        N, T, Fa, Ft, Ha, Ht = [int(v) for v in g['dims']]
        m.config.update(audio_embed_size=Fa, text_embed_size=Ft, audio_hidden_dims=Ha, text_hidden_dims=Ht, dropout=dropout,
                        batch_size=2, learning_rate=float(g['lr']))
        m.build(seed=0)
        m.model.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()})
        m.optimizer = nn.Adam(m.model.parameters(), lr=m.config['learning_rate'])
        m.fuse_features = [[g['xa'][i], g['xt'][i]] for i in range(N)]; m.fuse_targets = g['y']
        idx = list(range(min(N, 7)))          # batch_size 2 (the reference's): 2, 2, 2, 1 -> empty shards at world 3 in EVERY step
        with contextlib.redirect_stdout(io.StringIO()):
            m.train(1, idx); m.train(2, idx)
        out = dict(acc=int(m.train_acc))
        model = m.model
    if world > 1:                                 # the replicas must still be bit-identical: compare every rank's flat buffer with rank 0's
        import torch.distributed as dist
        ref = model._flat.clone(); dist.broadcast(ref, src=0)
        d = (model._flat - ref).abs().max().reshape(1); dist.all_reduce(d, op=dist.ReduceOp.MAX)
        out['replica_diff'] = float(d.item())
    if rank == 0:
        q.put(({k: v.cpu().numpy() for k, v in model.state_dict().items()}, out))
    if world > 1:
        parallel.barrier()
        dist.destroy_process_group()


def _spawn(case, world, dropout, port_base):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = port_base + os.getpid() % 500 + world
    procs = [ctx.Process(target=_run_generic, args=(case, r, world, port, q, dropout)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('case,port', [('text_clf', 26000), ('audio_reg', 26600), ('fuse_clf', 27200)])
def test_data_parallel_training_equals_single_process_for_every_loop(case, port):
    """text_bilstm_whole.train (attention + BiLSTM gradient ranges), audio_bilstm_perm.train (regression: the prediction
    all-reduce between the gradient exchange and the loss scalar -- ADVICE r2's ordering bug lived here) and fuse_net_whole.train
    (768 trainable floats, batch_size 2 < world 3): 2 and 3 ranks must leave the parameters and the aggregates of the
    single-process run."""
    sd1, out1 = _spawn(case, 1, 0.0, port)
    for world in (2, 3):
        sdw, outw = _spawn(case, world, 0.0, port)
        assert outw.pop('replica_diff') == 0.0
        for k, v in out1.items():
            assert np.allclose(v, outw[k], rtol=0, atol=1e-4), (case, world, k, v, outw[k])
        for k in sd1:
            assert np.abs(sd1[k] - sdw[k]).max() < 2e-6 + 1e-5 * np.abs(sd1[k]).max(), (case, world, k)


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
def test_data_parallel_with_dropout_is_deterministic_per_rank_and_keeps_replicas_identical():
    """Dropout on: a 2-rank run repeated from the same seeds gives bit-identical parameters (each rank's Philox streams depend
    only on seed, site and element), and rank 0's parameters equal rank 1's (checked through the all-reduced update: both ranks
    apply the same summed gradient; the queue carries rank 0's copy of two independent runs)."""
    a, oa = _spawn('text_clf', 2, 0.5, 27800)
    b, ob = _spawn('text_clf', 2, 0.5, 27800)
    assert oa['replica_diff'] == 0.0 and ob['replica_diff'] == 0.0
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    c, _ = _spawn('text_clf', 1, 0.5, 27800)                  # other mask partition (rank-keyed streams): close, not equal
    assert any(not np.array_equal(a[k], c[k]) for k in a) and max(np.abs(a[k] - c[k]).max() for k in a) < 0.1


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
def test_bench_two_rank_dry_run_on_one_gpu():
    """The N-rank code path of bench.py end to end -- self re-exec under torch.distributed.run, process group, sharded step with the
    gradient exchange, max-over-ranks timing, per-rank communication probe, teardown on every rank -- with the two ranks sharing
    this box's GPU through gloo (`--backend gloo`; RCCL refuses two ranks on one device).  What the driver's 8-GPU run executes differs
    only in the transport."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['config']['ranks'] == 2 and d['config']['global_batch'] == 1024 and d['scaling'] == 'weak'
    assert d['config']['backend'].startswith('rccl via torch.distributed') or d['config']['backend'] == 'torch.distributed'
    assert len(d['extra']['comm_alone_ms_per_step_by_rank']) == 2 and d['value'] > 0
    ex = d['extra']['exposed_comm']                           # per-rank step time without the exchange, exposed = step - that
    assert len(ex['step_without_comm_ms_by_rank']) == 2 and len(ex['exposed_comm_ms_by_rank']) == 2
    assert all(t > 0 for t in ex['step_without_comm_ms_by_rank']) and d['extra']['other_workloads'] is None
    assert d['extra']['f32_exact'] is None and d['extra']['train_e2e'] is None and 'cpu_baseline' not in d       # rank-0-at-N=1 legs only
    # round 5: the line explains itself -- every rank's forward path / sweep status, and the transport at top level
    rk = d['ranks']
    assert len(rk['gru_forward_path_by_rank']) == 2 and rk['sweep_status_by_rank'] == [0, 0] and rk['native_rccl_by_rank'] == [False, False]
    assert d['comm_transport'] == 'torch.distributed' and d['comm_transport_is_native_rccl'] is False and 'gloo' in d['comm_transport_reason']
    assert 'rank' in d['gru_forward_path']              # "... (all 2 ranks)" or "MIXED over ranks: ..."
