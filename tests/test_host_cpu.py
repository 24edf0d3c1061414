"""Host logic on CPU: loaders' helpers, metrics, augmentation, data-parallel sharding (gloo, world_size 2),
and the stock-torch CPU baseline against the reference fixtures."""
import itertools
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from icassp2022_depression_amd import _common, parallel
from oracle import ref_numpy as R
from oracle import torch_cpu_baseline as TB


def test_confusion_matrix_and_prf():
    y = np.array([1, 1, 0, 0, 1, 0]); p = np.array([1, 0, 0, 1, 1, 0])
    cm = _common.standard_confusion_matrix(torch.from_numpy(y), p)
    assert cm.tolist() == [[2, 1], [1, 2]]                   # [[TP, FP], [FN, TN]]
    acc, prec, rec, f1 = _common.prf(cm)
    assert (acc, prec, rec) == (4 / 6, 2 / 3, 2 / 3) and abs(f1 - 2 / 3) < 1e-12
    with pytest.raises(ValueError):                          # one class only: the reference's 2x2 unpack fails
        _common.standard_confusion_matrix(np.array([1, 1]), np.array([1, 1]))
    acc, prec, rec, f1 = _common.prf(np.array([[0, 0], [3, 3]]))   # 0/0 precision -> nan, not an exception
    assert np.isnan(prec) and rec == 0.0 and np.isnan(f1)


def test_minibatches_ragged_tail():
    assert list(_common.minibatches(10, 4)) == [(0, 4), (4, 8), (8, 10)]
    assert list(_common.minibatches(8, 8)) == [(0, 8)]


def test_permutation_augment_matches_itertools_order():
    rng = np.random.default_rng(0)
    feats = rng.standard_normal((5, 3, 4)); targs = np.array([0, 1, 0, 1, 0])
    f2, t2, idxs = _common.permutation_augment(feats, targs, [0, 1, 3, 4], lambda i: targs[i] == 1, (0, 1, 4, 5), label=1)
    assert idxs == [0, 5, 6, 7, 8, 9, 10, 11, 12, 4] and f2.shape == (13, 3, 4) and t2[5:].tolist() == [1] * 8
    perms = list(itertools.permutations(feats[1], 3))
    for k, c in enumerate((0, 1, 4, 5)):
        assert np.array_equal(f2[5 + k], np.stack(perms[c]))
    f3, t3, idxs3 = _common.permutation_augment(feats, targs.astype(float), [1], lambda i: True, range(6))
    assert len(idxs3) == 6 and np.array_equal(f3[5], feats[1]) and t3[5:].tolist() == [1.0] * 6


def test_shard_slice_partitions_every_batch():
    for n in (1, 2, 5, 8, 511, 512):
        for w in (1, 2, 4, 8):
            parts = [parallel.shard_slice(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_torch_baseline_matches_reference_fixture():
    """oracle/torch_cpu_baseline.py is the CPU yardstick of bench.py: same state_dict -> same outputs as the
    reference's own classes (fixtures), so timing it times the reference's arithmetic."""
    for name, cls in (('audio_clf_mid', TB.AudioClf), ('text_clf_mid', TB.TextClf)):
        g = load_golden(name)
        B, T, F, H = [int(v) for v in g['shape']]
        m = cls(F, H, p=0.0)
        missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()}, strict=True)
        m.eval()
        with torch.no_grad():
            out = m(torch.from_numpy(g['x'])).numpy()
        assert np.abs(out - g['out_eval']).max() < 1e-6


def test_torch_fusion_baseline_matches_reference_fixture():
    """The CPU yardstick of the fusion workload (bench.py --workload fusion): the reference's fusion_net state_dict loaded
    into oracle.torch_cpu_baseline.FusionClf reproduces the reference's features, output, MyLoss and weight gradient."""
    g = load_golden('fuse_clf')
    Ft, Ht, Fa, Ha = g['xt'].shape[2], g['sd']['fc_out.1.weight'].shape[0], g['xa'].shape[2], g['sd']['fc_audio.1.weight'].shape[0]
    m = TB.FusionClf(Ft, Ht, Fa, Ha, p=0.0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g['sd'].items()}, strict=True)
    m.eval()
    tf, af = m.pretrained_feature(torch.from_numpy(g['xa']), torch.from_numpy(g['xt']))
    assert np.abs(tf.numpy() - g['text_feature']).max() < 1e-5
    assert np.abs(af.numpy() - g['audio_feature']).max() < 1e-5
    with torch.no_grad():
        assert np.abs(m(torch.cat((tf, af), 1)).numpy() - g['out']).max() < 1e-6
    loss = TB.my_loss(m, tf, af, torch.from_numpy(g['y']).long())
    loss.backward()
    assert abs(loss.item() - g['losses'][0]) < 1e-5
    assert np.abs(m.fc_final[0].weight.grad.numpy() - g['gW']).max() < 1e-5 * max(1.0, np.abs(g['gW']).max())


def test_bench_launches_n_ranks_itself():
    """`bench.py --gpus N` started as ONE plain process must become N ranks: the launch path is
    exercised on CPU with gloo; the process group's size is checked against --gpus."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-check', '--backend', 'gloo'],
                       env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and lines, r.stdout[-1500:] + r.stderr[-1500:]
    res = json.loads(lines[-1])
    assert res == {'launch_check': True, 'world': 2, 'sum': 2.0, 'backend': 'gloo'}
    # a group whose size differs from --gpus is refused, not silently benchmarked as N=1
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-check', '--backend', 'gloo'],
                       env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'process group has 1 ranks' in (r.stdout + r.stderr)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from icassp2022_depression_amd import parallel as par
    par.init_from_env('gloo')
    g = load_golden('audio_clf_tiny')
    P = R.to_f64(g['sd']); x = g['x'].astype(np.float64); y = g['y']
    B = x.shape[0] + 1                                         # 5 rows -> ragged shards 3 + 2
    x = np.concatenate([x, x[:1] * 0.5]); y = np.concatenate([y, y[:1]])
    lo, hi = par.shard_slice(B)
    par.set_global_count(B)
    out, cache = R.audio_forward(P, x[lo:hi], {'rnn_layers': 2}, 'clf')
    # every rank normalises by the GLOBAL batch: local mean-loss gradient * (n_local / n_global)
    n_loc = hi - lo
    _, dout = R.ce_on_probs(out, y[lo:hi])
    dout = dout * n_loc / par.global_count(n_loc)
    _, G = R.audio_backward(P, dout, cache)
    names = sorted(G)

    class FakeModel:
        _grad_ready = True
        bucket = torch.from_numpy(np.concatenate([G[k].reshape(-1) for k in names]))

        def live_grad_bucket(self):
            return self.bucket
    fm = FakeModel()
    par.all_reduce_grads(fm)                                   # ONE collective on ONE contiguous bucket
    if rank == 0:
        of, cf = R.audio_forward(P, x, {'rnn_layers': 2}, 'clf')
        _, df = R.ce_on_probs(of, y)
        _, Gf = R.audio_backward(P, df, cf)
        ref = np.concatenate([Gf[k].reshape(-1) for k in names])
        q.put(float(np.abs(fm.bucket.numpy() - ref).max() / np.abs(ref).max()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_sum_of_shards_equals_full_batch_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-12


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from icassp2022_depression_amd import parallel as par
    par.init_from_env('gloo')
    n_live = 1000
    g = torch.from_numpy(np.random.default_rng(100 + rank).standard_normal(n_live))       # fp64 "gradient bucket" of this rank
    whole = g.clone(); dist.all_reduce(whole)
    # the audio classifier's plan shape: top layer + head in-call, layer 0 and the LayerNorm pair afterwards
    spans = par.layer_buckets([(400, 500), (0, 400), (900, 100)], n_live)
    pieces = g.clone()
    for s, c in spans:
        dist.all_reduce(pieces[s:s + c])
    if rank == 0:
        q.put(float((whole - pieces).abs().max()))
    dist.barrier(); dist.destroy_process_group()


def test_per_layer_buckets_equal_single_bucket_gloo():
    """Reducing the flat gradient buffer range by range (the overlapped schedule) gives what one all-reduce of the whole
    bucket gives; and a plan that leaves a gap or overlaps is rejected before anything is sent."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29300 + os.getpid() % 500
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    diff = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert diff < 1e-12
    with pytest.raises(ValueError):
        parallel.layer_buckets([(0, 400), (500, 500)], 1000)
    with pytest.raises(ValueError):
        parallel.layer_buckets([(0, 400), (300, 700)], 1000)


def test_checkpoint_loader_is_safe_first_and_the_exporter_writes_a_module(tmp_path, monkeypatch):
    """ADVICE r1 (checkpoint format): dict checkpoints load with weights_only=True; `export_reference_checkpoint` writes what
    the reference's consumers read (`torch.load(p).state_dict()`); a module pickle is refused when pickles are disallowed."""
    import collections
    import torch
    from icassp2022_depression_amd import _common

    sd = collections.OrderedDict([('ln.weight', torch.arange(4.)), ('gru.weight_ih_l0', torch.ones(6, 4)),
                                  ('fc_audio.1.weight', torch.zeros(3, 2)), ('fc_audio.1.bias', torch.full((3,), 2.))])

    class Fake:
        variant = 'clf'

        def state_dict(self):
            return sd

    stem = str(tmp_path / 'ckpt')
    _common.save(Fake(), stem)
    got = _common.load_checkpoint_state_dict(stem + '.pt', allow_pickle=False)       # tensor-only: no unpickling needed
    assert list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)

    path = _common.export_reference_checkpoint(Fake(), str(tmp_path / 'ref_style'))
    mod = torch.load(path, weights_only=False)                                       # what the reference's consumers do
    msd = mod.state_dict()
    assert list(msd) == list(sd) and all(torch.equal(msd[k], sd[k]) for k in sd)
    with pytest.raises(RuntimeError, match='allow_pickle'):
        _common.load_checkpoint_state_dict(path, allow_pickle=False)
    monkeypatch.setenv('DEP_ALLOW_PICKLE', '0')
    with pytest.raises(RuntimeError):
        _common.load_checkpoint_state_dict(path)
    monkeypatch.delenv('DEP_ALLOW_PICKLE')
    again = _common.load_checkpoint_state_dict(path)
    assert list(again) == list(sd)


# ----------------------------------------------------------------------------- collective order of the regression train() loops
def _order_worker(rank, world, port, q, modname):
    """a rank whose shard of the last mini-batch is empty must issue the step's collectives in the order
    the working ranks do -- gradients, predictions (hi - lo floats), loss scalar.  The script's real train() runs on every
    rank with a CPU stand-in for the model (the HIP model needs a GPU; the loop, nn.empty_shard_step, nn.Loss.item and
    parallel.* are the product's).  7 rows, batch 5, world 3: the second mini-batch (2 rows) leaves rank 2 empty; a size
    mismatch makes gloo raise (or pairs the loss with a prediction)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import importlib
    import torch
    import torch.distributed as dist
    from icassp2022_depression_amd import nn, parallel as par
    m = importlib.import_module('icassp2022_depression_amd.' + modname)
    par.init_from_env('gloo')

    class Out:
        def __init__(self, v):
            self.data = v

    class Fake:
        device = torch.device('cpu')
        _grad_ready = False
        _n_live = 4

        def __init__(self):
            self.bucket = torch.zeros(4)

        def train(self):
            pass

        def live_grad_bucket(self):
            return self.bucket

        def sync_plan(self):
            return {}, []

        def _rows(self, x):
            if isinstance(x, list):                         # fusion: list of (audio, text) pairs
                return torch.tensor([float(np.asarray(p[0]).sum()) for p in x])
            if isinstance(x, tuple):                        # fusion: (audio, text) tensors from _common.PairFeeder
                return x[0].reshape(x[0].shape[0], -1).sum(1)
            return x.reshape(x.shape[0], -1).sum(1)

        def __call__(self, x):
            return Out(self._rows(x).view(-1, 1) if not isinstance(x, torch.Tensor) or x.dim() != 2 else x.sum(1, keepdim=True))

        def pretrained_feature(self, x):
            r = self._rows(x).view(-1, 1)
            return r, 2 * r

    class Opt:
        def zero_grad(self):
            fake._grad_ready = False

        def step(self):
            pass

    def crit(*a):
        out = a[0].data if hasattr(a[0], 'data') and not isinstance(a[0], torch.Tensor) else a[0]
        n_glob = par.global_count(out.shape[0])
        val = out.double().sum().float().view(1) / n_glob

        def bw():
            fake.bucket = torch.full((4,), float(out.shape[0]))
            fake._grad_ready = True
            par.all_reduce_grads(fake)
        return nn.Loss(val, bw, reduce=par.world_size() > 1)

    fake = Fake()
    N = 7
    feats = np.arange(N * 6, dtype=np.float32).reshape(N, 3, 2) / 10
    targs = np.arange(N, dtype=np.float32)
    m.config.update(batch_size=5)
    m.train_dep_idxs, m.train_non_idxs = [0, 1, 2], [3, 4, 5, 6]
    if modname == 'fuse_net':
        m.fuse_features = [[feats[i], feats[i] * 2] for i in range(N)]; m.fuse_targets = targs
        m.optimizer, m.criterion = Opt(), crit
        args = (fake, 1)
    else:
        which = 'audio' if modname.startswith('audio') else 'text'
        setattr(m, which + '_features', feats); setattr(m, which + '_targets', targs)
        m.model, m.optimizer, m.criterion = fake, Opt(), crit
        args = (1,)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        mae = m.train(*args)
    q.put((rank, float(mae)))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize('modname', ['audio_bilstm_perm', 'text_bilstm_perm', 'fuse_net'])
def test_regression_train_empty_shard_collective_order_gloo(modname):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 28100 + os.getpid() % 800 + {'audio_bilstm_perm': 0, 'text_bilstm_perm': 1, 'fuse_net': 2}[modname]
    procs = [ctx.Process(target=_order_worker, args=(r, 3, port, q, modname)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank assembled the same predictions (row sums of its features) -> the same MAE, equal to the serial value
    feats = np.arange(7 * 6, dtype=np.float32).reshape(7, 3, 2) / 10
    pred = feats.reshape(7, -1).sum(1) * (3 if modname == 'fuse_net' else 1)      # the fusion stand-in outputs text + audio = 3 x
    expect = float(np.mean(np.abs(np.arange(7) - pred)))
    for r in range(3):
        assert abs(got[r] - expect) < 1e-5, (r, got[r], expect)


def test_device_feature_cache_notices_in_place_edits(monkeypatch):
    """the HBM-resident feature cache is keyed on a content check, not only on the array object.  Default check: a
    strided sample plus one probe per row; DEP_FEATURES_HASH=1: the whole array (xxh3)."""
    import torch
    from icassp2022_depression_amd import _common
    _common.invalidate_device_features()
    X = np.random.default_rng(0).standard_normal((40, 6, 8)).astype(np.float32)
    t0 = _common.device_features(X, 'cpu', role='probe_test')
    assert _common.device_features(X, 'cpu', role='probe_test') is t0                  # unchanged array: the cached copy
    r = 17
    col = (r * 2654435761) % 48
    X.reshape(40, -1)[r, col] += 1.0                                                   # a middle row, edited in place at its probe
    t1 = _common.device_features(X, 'cpu', role='probe_test')
    assert t1 is not t0 and torch.equal(t1, torch.from_numpy(X))
    monkeypatch.setenv('DEP_FEATURES_HASH', '1')
    t2 = _common.device_features(X, 'cpu', role='probe_test')
    X.reshape(40, -1)[r, (col + 1) % 48] += 1.0                                        # an element no probe looks at
    t3 = _common.device_features(X, 'cpu', role='probe_test')
    assert t3 is not t2 and torch.equal(t3, torch.from_numpy(X))
    # the fusion scripts' pair list: editing a MIDDLE pair in place refreshes the stacked copy
    pairs = [[np.full((3, 4), float(i), np.float32), np.full((3, 5), float(-i), np.float32)] for i in range(6)]
    monkeypatch.delenv('DEP_FEATURES_HASH')
    f0 = _common.PairFeeder(pairs, range(6), 'cpu')
    pairs[3][1][0, 0] = 99.0
    f1 = _common.PairFeeder(pairs, range(6), 'cpu')
    assert float(f1.Xt[3, 0, 0]) == 99.0 and f1.Xt is not f0.Xt
    _common.invalidate_device_features()
