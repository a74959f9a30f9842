"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU
tests).  The only collective of the path is ONE all-reduce of the flat fp32 gradient buffer per step
 -- the reference's Lightning DDP fires one bucketed all-reduce per manual_backward
(main.py:112, lit_wrapper.py:49,56,72); summing locally first is mathematically identical."""
import os

import torch
import torch.distributed as dist


def local_device_index():
    """GPU of this rank: LOCAL_RANK, unless SININN_FORCE_DEVICE pins every rank to one device (rehearsals of the
    multi-rank path on a single-GPU box, together with SININN_DIST_BACKEND=gloo)."""
    forced = os.environ.get('SININN_FORCE_DEVICE')
    return int(forced) if forced is not None else int(os.environ.get('LOCAL_RANK', '0'))


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None, allow_single=False):
    """Initialise the default process group from torchrun's environment (no-op for a single process unless allow_single:
    a one-rank RCCL group, used by the test of the collective's stream ordering)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if (ws <= 1 and not allow_single) or dist.is_initialized():
        return world()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = os.environ.get('SININN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    kw = {}
    if backend == 'nccl':
        torch.cuda.set_device(local_device_index())
        try:        # RCCL's internal stream at high priority: the gradient all-reduce is the tail of every step
            kw['pg_options'] = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except (AttributeError, TypeError):
            pass
    try:
        dist.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=ws, **kw)
    except TypeError:                   # a torch build whose init_process_group does not take pg_options
        dist.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=ws)
    return world()


# bench.py sets TIMING[0] = [] around its timed region: every gradient all-reduce then appends a (start, end) pair of HIP events --
# start on the stream the collective is ordered behind (it fires when the last weight-gradient kernel has finished), end on the
# caller's stream right after it has been made to wait for the collective (it fires when the reduced buffer is visible to Adam).
TIMING = [None]


def allreduce_mean_(flat_buffers):
    """In-place average of each flat gradient buffer over the ranks (one collective per buffer)."""
    _, ws = world()
    if ws == 1:
        return
    for buf in flat_buffers:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf.div_(ws)


def allreduce_sum_(flat_buffers, after=None, _force=False):
    """In-place SUM of each flat gradient buffer over the ranks; the 1/world factor is folded into the fused Adam launch
    (sininn_adam_step's grad_scale) instead of a separate pass over the buffer.

    after: the HIP stream whose queued work produces the buffers (the weight-gradient stream).  The collective is then
    ordered behind THAT stream only -- it is issued with `after` as the current stream, so RCCL's own (high-priority)
    communication stream waits for the last weight-gradient kernel and not for whatever else the caller's stream still has
    in flight (tail of the data-gradient chains, loss logging) -- and the caller's current stream waits for the collective
    (`work.wait()` is a stream wait, the host does not block).  Adam, launched next on the caller's stream, is thereby
    ordered after the reduced gradients.  (_force: run the collective in a one-rank group too -- ordering test.)"""
    _, ws = world()
    if ws == 1 and not _force:
        return
    timed = TIMING[0] is not None and bool(flat_buffers) and flat_buffers[0].is_cuda
    if timed:                                   # bench.py: (start, end) HIP events per step, see the docstring of TIMING
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(after if after is not None else torch.cuda.current_stream())
    if after is not None and flat_buffers and flat_buffers[0].is_cuda and dist.get_backend() == 'nccl':
        with torch.cuda.stream(after):
            works = [dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True) for buf in flat_buffers]
        for w in works:
            w.wait()
    else:
        if after is not None and flat_buffers and flat_buffers[0].is_cuda:
            torch.cuda.current_stream().wait_stream(after)     # gloo rehearsal on GPU tensors: plain stream order
        for buf in flat_buffers:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if timed:
        e1.record(torch.cuda.current_stream())
        TIMING[0].append((e0, e1))


def broadcast_(tensors, src=0):
    _, ws = world()
    if ws == 1:
        return
    for t in tensors:
        dist.broadcast(t, src=src)


def shard_indices(global_indices, rank=None, world_size=None):
    """Rank r takes positions r::world of every global batch."""
    r, ws = world()
    rank = r if rank is None else rank
    world_size = ws if world_size is None else world_size
    return global_indices[rank::world_size]
