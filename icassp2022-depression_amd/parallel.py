"""Data parallelism over the utterance (volunteer) axis: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code; utterances are independent and every
loss is a batch mean, so the path shards naturally:
  * each global mini-batch is cut into contiguous per-rank slices (`shard_slice`);
  * every rank normalises its loss gradient by the GLOBAL batch size (`global_count`), so the SUM
    all-reduce of the single flat gradient bucket reproduces the reference's batch-mean gradient
    exactly, ragged last batches included;
  * one collective per step: `all_reduce_grads(model)` on `model.live_grad_bucket()`
    (3.4 MB audio / 6.4 MB text / 3 KB fusion) -- latency-bound, so one bucket, no ring tuning.
"""
import os

import torch

_state = {'init': False}


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* when launched by torchrun."""
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 or dist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=world)


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def world_size():
    d = _dist()
    return d.get_world_size() if d else 1


def rank():
    d = _dist()
    return d.get_rank() if d else 0


def shard_slice(n, r=None, w=None):
    """Contiguous slice [lo, hi) of an n-row mini-batch owned by rank r of w (sizes differ by <= 1)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


_count_override = [None]


def set_global_count(n):
    """train() announces the global mini-batch size before calling the criterion on its shard."""
    _count_override[0] = n


def global_count(local):
    n = _count_override[0]
    if world_size() == 1 or n is None:
        return local
    return n


def all_reduce_sum(t):
    d = _dist()
    if d:
        d.all_reduce(t, op=d.ReduceOp.SUM)
    return t


def all_reduce_grads(model):
    """SUM all-reduce of the model's single contiguous live-gradient bucket (RCCL over xGMI)."""
    d = _dist()
    if d and model._grad_ready:
        d.all_reduce(model.live_grad_bucket(), op=d.ReduceOp.SUM)


def broadcast_params(model, src=0):
    d = _dist()
    if d:
        d.broadcast(model._flat, src=src)


def barrier():
    d = _dist()
    if d:
        d.barrier()
