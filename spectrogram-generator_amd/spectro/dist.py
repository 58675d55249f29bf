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


def gather_equal(t, dst: int | None = None):
    """Gather equal-shaped tensors: all ranks get the list when ``dst`` is None, else only ``dst``."""
    import torch
    d = _dist()
    world, rank = world_info()
    if world == 1:
        return [t]
    if dst is None:
        out = [torch.empty_like(t) for _ in range(world)]
        d.all_gather(out, t.contiguous())
        return out
    return gather_to_root([t], dst)[0] if rank == dst else (gather_to_root([t], dst) or None)


def gather_to_root(tensors, dst: int = 0, shapes=None):
    """Ragged gather by direct peer sends (RCCL has no gatherv): every rank sends its tensors to ``dst``.

    ``shapes[r]`` lists the shapes rank ``r`` sends; when None they are exchanged first with
    ``all_gather_object``.  Returns on ``dst`` a list (per rank) of lists of tensors, elsewhere None."""
    import torch
    d = _dist()
    world, rank = world_info()
    if world == 1:
        return [list(tensors)]
    if shapes is None:
        shapes = [None] * world
        d.all_gather_object(shapes, [tuple(t.shape) for t in tensors])
    if rank == dst:
        out, reqs = [], []
        for r in range(world):
            if r == dst:
                out.append(list(tensors))
                continue
            bufs = [torch.empty(s, dtype=tensors[0].dtype if tensors else torch.float32,
                                device=tensors[0].device if tensors else "cpu") for s in shapes[r]]
            reqs += [d.irecv(b, src=r) for b in bufs]
            out.append(bufs)
        for q in reqs:
            q.wait()
        return out
    for q in [d.isend(t.contiguous(), dst=dst) for t in tensors]:
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
