"""Data parallelism over the utterance (volunteer) axis: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code; utterances are independent and every
loss is a batch mean, so the path shards naturally:
  * each global mini-batch is cut into contiguous per-rank slices (`shard_slice`);
  * every rank normalises its loss gradient by the GLOBAL batch size (`global_count`), so the SUM
    all-reduce of the single flat gradient bucket reproduces the reference's batch-mean gradient
    exactly, ragged last batches included;
  * the gradient exchange is a SUM all-reduce of the model's flat gradient buffer.  With the native RCCL communicator
    (`init_native_comm`, the C-ABI's dep_comm_* entry points; built by `init_from_env` whenever the backend is nccl) it is cut at layer boundaries and
    overlapped with the backward pass: the top layer's range [layer L-1 | head] is enqueued on a communication stream behind
    the next layer's backward sweep and travels over xGMI beside that layer's weight-gradient GEMMs
    (dep_rnn_backward_overlapped); what becomes final later (layer 0, LayerNorm) follows as one grouped operation, and the
    compute stream waits for the communication stream before the optimizer reads the gradients.  Without it (gloo tests,
    DEP_COMM=torch) the whole bucket is reduced in one torch.distributed all-reduce after the backward.
    Every rank issues the same sequence of collectives each step -- also a rank whose shard of a small mini-batch is
    empty (`nn.empty_shard_step`).
"""
import os

import torch

_state = {'init': False}


def init_from_env(backend=None, native=True):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* when launched by torchrun.  On GPUs (backend "nccl")
    this also builds the C-ABI's own RCCL communicator, so that the training scripts -- which call only this -- take the
    overlapped gradient path bench.py measures; `native=False` / DEP_COMM=torch keeps torch.distributed's."""
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
    if backend == 'nccl' and native:
        init_native_comm()


def transport():
    """Which gradient transport the training step uses: 'none' (one rank), 'rccl-native' or 'torch.distributed'."""
    if world_size() == 1 and _native['comm'] is None:
        return 'none'
    return 'rccl-native' if _native['comm'] is not None else 'torch.distributed'


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
_comm_off = [False]


def set_comm_enabled(on):
    """Measurement only (bench.py `exposed_comm_ms`): with the gradient exchange off every rank steps on its LOCAL gradients --
    the replicas drift apart, so the caller snapshots and restores the parameters around such a leg."""
    _comm_off[0] = not on


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
    """SUM all-reduce of the model's single contiguous live-gradient bucket (torch.distributed transport)."""
    d = _dist()
    if d and model._grad_ready:
        from . import _lib as L
        L.order_note('C torch.distributed all_reduce of the live bucket (on the compute stream)')
        d.all_reduce(model.live_grad_bucket(), op=d.ReduceOp.SUM)


# ----------------------------------------------------------------------------- native RCCL communicator (C-ABI)
_native = {'comm': None, 'stream': None, 'world': 1, 'why': None}


def _agree(ok):
    """MIN over ranks of a local success flag (torch.distributed side channel): every rank takes the same branch."""
    d = _dist()
    if not d or world_size() == 1:
        return bool(ok)
    dev = 'cuda' if d.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    d.all_reduce(t, op=d.ReduceOp.MIN)
    return bool(int(t.item()))


def init_native_comm(force_single=False):
    """Create the RCCL communicator through the C-ABI (dep_comm_unique_id / dep_comm_init): rank 0 obtains the id, the
    torch.distributed group (already initialised by init_from_env) carries it to the other ranks.  force_single builds a
    one-rank communicator (tests on a single GPU).

    No rank may enter a collective the others will not reach: the ranks first agree that librccl resolves
    everywhere (dep_comm_available + MIN), rank 0's id travels WITH a status byte, and after ncclCommInitRank they agree
    again.  Any disagreement -> None on EVERY rank (the torch.distributed all-reduce of the whole bucket is used instead)."""
    import ctypes as C
    from . import _lib as L
    if _native['comm'] is not None:
        return _native['comm']
    d = _dist()
    world, rk = world_size(), rank()
    if world == 1 and not force_single:
        return None
    if os.environ.get('DEP_COMM', 'rccl') == 'torch':
        return None
    try:
        lib = L.load()
        have = bool(lib.dep_comm_available()) and torch.cuda.is_available()
    except Exception:                                       # noqa: BLE001 -- a rank without the library still has to vote
        lib, have = None, False
    if not _agree(have):
        _native['why'] = 'librccl / the HIP library is not available on every rank'
        return None
    dev = torch.cuda.current_device()
    msg = bytearray(129)                                    # [status | 128-byte id]
    if rk == 0:
        idbuf = (C.c_char * 128)()
        if lib.dep_comm_unique_id(idbuf, 128) == 0:
            msg[0] = 1; msg[1:] = idbuf.raw
    t = torch.frombuffer(msg, dtype=torch.uint8).clone().cuda()
    if d and world > 1:
        d.broadcast(t, src=0)
    raw = bytes(t.cpu().numpy().tobytes())
    if raw[0] != 1:
        _native['why'] = 'rank 0 could not create the RCCL id'
        return None
    comm = C.c_void_p()
    rc = lib.dep_comm_init(C.byref(comm), world, rk, raw[1:], 128, dev)
    if not _agree(rc == 0):
        if rc == 0:
            lib.dep_comm_destroy(comm)
        _native['why'] = 'ncclCommInitRank failed on some rank'
        return None
    _native.update(comm=comm, stream=torch.cuda.Stream(), world=world, why=None)
    return comm


def native_comm():
    return _native['comm']


def destroy_native_comm():
    from . import _lib as L
    if _native['comm'] is not None:
        torch.cuda.synchronize()
        L.load().dep_comm_destroy(_native['comm'])
        _native.update(comm=None, stream=None, world=1, why=None)


def layer_buckets(spans, n_live):
    """Pure helper (CPU-testable): `spans` = ordered list of (start, count) ranges of the flat live-gradient buffer in the
    order they become final during the backward; checks they are disjoint and cover [0, n_live) exactly."""
    marks = sorted((s, s + c) for s, c in spans if c > 0)
    pos = 0
    for a, b in marks:
        if a != pos:
            raise ValueError(f'gradient ranges leave a gap or overlap at {pos} (next range starts at {a})')
        pos = b
    if pos != n_live:
        raise ValueError(f'gradient ranges cover {pos} of {n_live} live floats')
    return [(s, c) for s, c in spans if c > 0]


def make_grad_sync(model, in_call):
    """GradSync for dep_rnn_backward_overlapped, or None without a native communicator.  in_call: {layer: (start, count)}."""
    if _native['comm'] is None or not in_call or _comm_off[0]:
        return None
    from . import _lib as L
    gs = L.GradSync()
    gs.comm = _native['comm']
    gs.comm_stream = _native['stream'].cuda_stream
    base = model._flat_grad.data_ptr()
    for l, (start, count) in in_call.items():
        gs.range_ptr[l] = base + 4 * start
        gs.range_count[l] = count
    return gs


def finish_grad_sync(model, in_call, post):
    """After the model's backward: reduce what is still local and join the streams.
    native communicator: `post` ranges (final only now) as one grouped RCCL operation on the communication stream, then the
    compute stream waits for that stream; otherwise one torch.distributed all-reduce of the whole bucket."""
    if _comm_off[0]:
        return
    if _native['comm'] is None:
        all_reduce_grads(model)
        return
    import ctypes as C
    from . import _lib as L
    layer_buckets(list(in_call.values()) + list(post), model._n_live)          # every live float is reduced exactly once
    cs = _native['stream']
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    cs.wait_event(ev)
    post = [(s, c) for s, c in post if c > 0]
    if post:
        base = model._flat_grad.data_ptr()
        ptrs = (C.c_void_p * len(post))(*[base + 4 * s for s, _ in post])
        cnts = (C.c_long * len(post))(*[c for _, c in post])
        L.check(L.load().dep_comm_allreduce_ranges(_native['comm'], ptrs, cnts, len(post), cs.cuda_stream),
                'dep_comm_allreduce_ranges')
    ev2 = torch.cuda.Event(); ev2.record(cs)
    cur.wait_event(ev2)
    L.order_note('join: the compute stream waits for the communication stream')


def reduce_zero_contribution(model, in_call, post):
    """A rank whose shard is empty: same collectives, in the same order, on a zeroed gradient buffer."""
    if _native['comm'] is None:
        all_reduce_grads(model)
        return
    from . import _lib as L
    cs = _native['stream']
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    cs.wait_event(ev)
    base = model._flat_grad.data_ptr()
    for l in sorted(in_call, reverse=True):                                    # the backward walks the layers top -> bottom
        s, c = in_call[l]
        if c > 0:
            L.check(L.load().dep_comm_allreduce(_native['comm'], base + 4 * s, c, cs.cuda_stream), 'dep_comm_allreduce')
    finish_grad_sync(model, in_call, post)


def broadcast_params(model, src=0):
    d = _dist()
    if d:
        d.broadcast(model._flat, src=src)


def barrier():
    d = _dist()
    if d:
        d.barrier()
