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
    for world in (1, 2):
        q = ctx.Queue()
        port = 29700 + os.getpid() % 1000 + world
        procs = [ctx.Process(target=_run, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = q.get(timeout=240)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    sd1, acc1 = res[1]; sd2, acc2 = res[2]
    assert acc1 == acc2
    for k in sd1:
        assert np.abs(sd1[k] - sd2[k]).max() < 2e-6, k
