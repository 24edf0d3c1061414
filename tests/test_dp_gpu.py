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


def _run(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    sys.path.insert(0, ROOT)
    from icassp2022_depression_amd import audio_gru_whole as m, nn, parallel
    if world > 1:
        parallel.init_from_env('gloo')
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
