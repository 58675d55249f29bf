"""Multi-GPU sharding for the STFT path: one process per GPU, clips shard, RCCL only for the final exchange.

Frames and clips are independent, so the data path has NO collective: every rank runs
``sg_stft`` on its own contiguous run of clips.  ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests) is used for exactly two things:

  * ``global_max``   -- all-reduce(MAX) of the per-shard spectrogram maximum, the one place where a
                        cross-GPU exchange changes *values* (batch-global ``base`` of PlotEngine.py:126);
  * ``gather_*``     -- gathering small reduced products (band features ``[frames, 2]``, band powers) or,
                        on request, the sharded spectra themselves to one rank.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a one-shot gather to a root is bounded by the
root's 7 inbound links and gains nothing from a ring, so ``gather_to_root`` posts direct peer sends.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np


def shard_range(n_items: int, world: int, rank: int):
    """Contiguous split of ``range(n_items)``: first ``n % world`` ranks get one extra. -> (start, stop)"""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def n_frames(n_samples: int, n_fft: int, hop: int) -> int:
    """A2: ``(N - n)//hop + 1`` frames, none when the clip is shorter than a frame (scipy:2180-2188)."""
    return 0 if n_samples < n_fft else (n_samples - n_fft) // hop + 1


def stft_cost(n_samples: int, n_fft: int, hop: int) -> float:
    """Relative cost of one (clip, n_fft, hop) work item: frames * n log n (BASELINE cfg4 balancing)."""
    frames = 0 if n_samples < n_fft else (n_samples - n_fft) // hop + 1
    return frames * n_fft * max(np.log2(n_fft), 1.0)


def deal_work_items(costs: Sequence[float], world: int):
    """Greedy longest-processing-time deal of work items to ranks -> list of index lists (deterministic)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    return out


def _dist():
    import torch.distributed as dist
    return dist


def world_info():
    d = _dist()
    if d.is_available() and d.is_initialized():
        return d.get_world_size(), d.get_rank()
    return 1, 0


def global_max(local_max):
    """All-reduce(MAX) of a 0-d/1-d tensor across ranks (identity when not distributed)."""
    d = _dist()
    world, _ = world_info()
    if world > 1:
        d.all_reduce(local_max, op=d.ReduceOp.MAX)
    return local_max


def _backend():
    d = _dist()
    try:
        return str(d.get_backend()).lower()
    except Exception:
        return ""


def _comm_tensor(t):
    """RCCL moves device memory only: under the "nccl" backend a host tensor is staged on this rank's GPU first
    (gloo, the CPU tests' backend, takes host tensors as they are).  -> (tensor to communicate, came_from_host)"""
    import torch
    if "nccl" in _backend() and not t.is_cuda:
        return t.to(torch.device("cuda", torch.cuda.current_device())), True
    return t, False


def gather_equal(t, dst: int | None = None):
    """Gather equal-shaped tensors, one per rank: all ranks get the list when ``dst`` is None, else only ``dst``
    (the other ranks get None)."""
    import torch
    d = _dist()
    world, rank = world_info()
    if world == 1:
        return [t]
    if dst is None:
        ct, staged = _comm_tensor(t.contiguous())
        out = [torch.empty_like(ct) for _ in range(world)]
        d.all_gather(out, ct)
        return [o.cpu() for o in out] if staged else out
    parts = gather_to_root([t], dst, shapes=[[tuple(t.shape)]] * world)
    return [parts[r][0] for r in range(world)] if rank == dst else None


def gather_to_root(tensors, dst: int = 0, shapes=None):
    """Ragged gather by direct peer sends (RCCL has no gatherv): every rank sends its tensors to ``dst``.

    ``shapes[r]`` lists the shapes rank ``r`` sends; when None they are exchanged first with
    ``all_gather_object``.  The sends and receives of a rank are posted as ONE batch (``batch_isend_irecv``), so under
    RCCL the root's seven inbound xGMI links carry their shards concurrently instead of one peer after the other.
    Host tensors are staged through the rank's GPU when the backend is RCCL and come back as host tensors.
    Returns on ``dst`` a list (per rank) of lists of tensors, elsewhere None."""
    import torch
    d = _dist()
    world, rank = world_info()
    if world == 1:
        return [list(tensors)]
    if shapes is None:
        shapes = [None] * world
        d.all_gather_object(shapes, [tuple(t.shape) for t in tensors])
    staged_any = False
    comm = []
    for t in tensors:
        ct, staged = _comm_tensor(t.contiguous())
        staged_any |= staged
        comm.append(ct)
    if rank == dst:
        dtype = comm[0].dtype if comm else torch.float32
        if comm:
            device = comm[0].device
        else:
            device = torch.device("cuda", torch.cuda.current_device()) if "nccl" in _backend() else torch.device("cpu")
            staged_any = "nccl" in _backend()
        out, ops = [], []
        for r in range(world):
            if r == dst:
                out.append(list(tensors))
                continue
            bufs = [torch.empty(s, dtype=dtype, device=device) for s in shapes[r]]
            ops += [d.P2POp(d.irecv, b, r) for b in bufs if b.numel()]
            out.append(bufs)
        if ops:
            for q in d.batch_isend_irecv(ops):
                q.wait()
        if staged_any:
            out = [part if r == dst else [b.cpu() for b in part] for r, part in enumerate(out)]
        return out
    ops = [d.P2POp(d.isend, t, dst) for t in comm if t.numel()]
    if ops:
        for q in d.batch_isend_irecv(ops):
            q.wait()
    return None


def run_sharded(n_clips: int, compute: Callable[[int, int], "object"], reduce_max: bool = False):
    """Run ``compute(start, stop)`` on this rank's clip shard; optionally all-reduce(MAX) its scalar result."""
    world, rank = world_info()
    start, stop = shard_range(n_clips, world, rank)
    res = compute(start, stop)
    if reduce_max:
        res = global_max(res)
    return (start, stop), res
