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

import os
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


# A world of ONE normally skips every collective.  With this set (or SPECTRO_DIST_FORCE_COLLECTIVES=1) and a process group
# initialised, the collectives run anyway -- RCCL executes them as one-rank collectives -- so that the code RCCL runs at N > 1
# (communicator set-up, host tensors staged through the GPU, all_reduce / all_gather on device memory) can be executed on a
# one-GPU box (tests/test_gpu_multiproc.py).
FORCE_COLLECTIVES = os.environ.get("SPECTRO_DIST_FORCE_COLLECTIVES") == "1"


def _alone(world: int) -> bool:
    if world > 1:
        return False
    d = _dist()
    return not (FORCE_COLLECTIVES and d.is_available() and d.is_initialized())


def global_max(local_max):
    """All-reduce(MAX) of a 0-d/1-d tensor across ranks (identity when not distributed)."""
    d = _dist()
    world, _ = world_info()
    if not _alone(world):
        t, staged = _comm_tensor(local_max)                  # RCCL reduces device memory only: a host tensor goes through this rank's GPU
        d.all_reduce(t, op=d.ReduceOp.MAX)
        if staged:
            local_max.copy_(t.cpu())
    return local_max


def all_agree(ok: bool) -> bool:
    """True on every rank iff ``ok`` is True on every rank (all-reduce MIN; identity when not distributed): lets ranks refuse TOGETHER,
    so that none of them walks into a collective the others have left."""
    d = _dist()
    world, _ = world_info()
    if _alone(world):
        return bool(ok)
    import torch
    flag, _ = _comm_tensor(torch.tensor([1 if ok else 0], dtype=torch.int32))
    d.all_reduce(flag, op=d.ReduceOp.MIN)
    return bool(int(flag.cpu()[0]))


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
    if _alone(world):
        return [t]
    if dst is None:
        ct, staged = _comm_tensor(t.contiguous())
        out = [torch.empty_like(ct) for _ in range(world)]
        d.all_gather(out, ct)
        return [o.cpu() for o in out] if staged else out
    parts = gather_to_root([t], dst, shapes=[[(tuple(t.shape), t.dtype)]] * world)
    return [parts[r][0] for r in range(world)] if rank == dst else None


def _dtype_name(dt) -> str:
    return str(dt).replace("torch.", "")


def _as_dtype(name):
    import torch
    if isinstance(name, torch.dtype):
        return name
    dt = getattr(torch, _dtype_name(name), None)
    if not isinstance(dt, torch.dtype):
        raise ValueError(f"gather_to_root: unknown dtype {name!r} in the shape table")
    return dt


def _split_entry(e):
    """A shape-table entry is ``shape`` or ``(shape, dtype)`` (dtype a torch.dtype or its name) -> (shape, dtype | None)."""
    if len(e) == 2 and isinstance(e[0], (tuple, list)) and not isinstance(e[1], (int, np.integer)):
        return tuple(int(v) for v in e[0]), _as_dtype(e[1])
    return tuple(int(v) for v in e), None


# dtype codes for the one small collective that settles a table given without dtypes (0: this rank sends nothing, -1: mixed)
_DTYPE_CODES = ("float32", "float64", "float16", "bfloat16", "int8", "uint8", "int16", "int32", "int64", "bool", "complex64", "complex128")


def gather_to_root(tensors, dst: int = 0, shapes=None, checked: bool = False):
    """Ragged gather by direct peer sends (RCCL has no gatherv): every rank sends its tensors to ``dst``.

    ``shapes[r]`` lists what rank ``r`` sends, one entry per tensor: ``(shape, dtype)`` -- then nothing is exchanged before the
    sends -- or a bare ``shape``; when ``shapes`` is None the table (shapes and dtypes) is exchanged first with ``all_gather_object``.
    The root sizes every receive buffer with the SENDER's dtype: a peer whose products are float64 while the root holds
    float32 (or nothing) arrives intact.  A table of bare shapes says nothing about dtypes, so the ranks settle them with one
    tiny all-gather of dtype codes first (a rank whose own tensors are of several dtypes, or two ranks that disagree, make
    EVERY rank raise -- before any send is posted, because a mismatched pair of P2P operations is wrong bytes under gloo and a
    hang under RCCL).  The sends and receives of a rank are posted as ONE batch (``batch_isend_irecv``), so under RCCL the
    root's seven inbound xGMI links carry their shards concurrently instead of one peer after the other.  Host tensors are
    staged through the rank's GPU when the backend is RCCL and come back as host tensors.
    Tensors that do not match their row of the table make every rank raise together (one all-reduce, skipped with
    ``checked=True`` when the caller's ranks have already agreed on exactly that).
    Returns on ``dst`` a list (per rank) of lists of tensors, elsewhere None."""
    import torch
    d = _dist()
    world, rank = world_info()
    if _alone(world):
        return [list(tensors)]
    if shapes is None:
        shapes = [None] * world
        d.all_gather_object(shapes, [(tuple(t.shape), _dtype_name(t.dtype)) for t in tensors])
    table = [[_split_entry(e) for e in shapes[r]] for r in range(world)]
    if any(dt is None for row in table for _, dt in row):
        mine = {_dtype_name(t.dtype) for t in tensors}
        code = 0 if not mine else (_DTYPE_CODES.index(next(iter(mine))) + 1 if len(mine) == 1 and next(iter(mine)) in _DTYPE_CODES else -1)
        ct, _ = _comm_tensor(torch.tensor([code], dtype=torch.int32))
        codes = [torch.empty_like(ct) for _ in range(world)]
        d.all_gather(codes, ct)
        codes = [int(c.cpu()[0]) for c in codes]             # the same list on every rank: they all raise, or none does
        used = {c for c in codes if c != 0}
        if -1 in used or len(used) > 1:
            raise ValueError("gather_to_root: a shape table without dtypes needs ONE dtype on every rank; got "
                             + ", ".join("none" if c == 0 else "mixed" if c < 0 else _DTYPE_CODES[c - 1] for c in codes)
                             + " -- pass (shape, dtype) entries")
        common = _as_dtype(_DTYPE_CODES[used.pop() - 1]) if used else torch.float32
        table = [[(s, common if dt is None else dt) for s, dt in row] for row in table]
    own = [(tuple(t.shape), t.dtype) for t in tensors]
    if not checked and not all_agree(own == table[rank]):    # one all-reduce; ``checked=True``: the caller's ranks have agreed already
        raise ValueError(f"gather_to_root: the tensors do not match the shape table" +
                         (f" (rank {rank} holds {own}, the table says {table[rank]})" if own != table[rank] else " (on another rank)"))
    staged_any = False
    comm = []
    for t in tensors:
        ct, staged = _comm_tensor(t.contiguous())
        staged_any |= staged
        comm.append(ct)
    if rank == dst:
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
            bufs = [torch.empty(s, dtype=dt, device=device) for s, dt in table[r]]
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


def frame_shards(n_samples: int, nperseg: int, hop: int, world: int):
    """One very long clip cut at frame boundaries: rank r owns frames ``[f_lo, f_hi)`` and reads samples
    ``[f_lo*hop, (f_hi-1)*hop + nperseg)`` -- neighbours both read the ``nperseg - hop`` samples around a cut (a halo
    READ of the caller's array, nothing is exchanged).  -> list of (f_lo, f_hi, s_lo, s_hi), one per rank."""
    total = n_frames(n_samples, nperseg, hop)
    out = []
    for r in range(world):
        lo, hi = shard_range(total, world, r)
        out.append((lo, hi, lo * hop, (hi - 1) * hop + nperseg) if hi > lo else (lo, lo, lo * hop, lo * hop))
    return out


def long_clip_spectrogram(x, fs=1.0, window=("tukey", .25), nperseg=None, noverlap=None, nfft=None, detrend="constant",
                          scaling="density", mode="psd", world=None, rank=None):
    """This rank's share of ``spectrogram(x, ...)`` for one long 1-D recording that every rank can read (a memory-mapped
    file, say): ``(f, t[f_lo:f_hi], Sxx[:, f_lo:f_hi], (f_lo, f_hi))``.  The shares of all ranks, concatenated along time,
    are the single-process result bit for bit: the frames are the same frames run by the same plan, and ``t`` is a slice of
    the full vector (A7), not recomputed from the shard."""
    from . import _capi
    from .signal import resolve_segments, spectrogram
    if getattr(x, "ndim", None) != 1:                          # ndarray or np.memmap: only this rank's samples are touched
        raise ValueError("long_clip_spectrogram takes one 1-D recording")
    if world is None or rank is None:
        world, rank = world_info()
    win, nps = resolve_segments(window, nperseg, input_length=x.shape[-1])
    nov = nps // 8 if noverlap is None else int(noverlap)
    if nov >= nps:
        raise ValueError("noverlap must be less than nperseg.")
    hop = nps - nov
    f_lo, f_hi, s_lo, s_hi = frame_shards(x.shape[0], nps, hop, world)[rank]
    t_all = _capi.times(x.shape[0], nps, hop, fs)
    if f_hi == f_lo:                                          # more ranks than frames
        f, _, s = spectrogram(x[:nps], fs, win, nps, nov, nfft, detrend, scaling=scaling, mode=mode)
        return f, t_all[:0], s[..., :0], (f_lo, f_hi)
    f, _, s = spectrogram(x[s_lo:s_hi], fs, win, nps, nov, nfft, detrend, scaling=scaling, mode=mode)
    assert s.shape[-1] == f_hi - f_lo
    return f, t_all[f_lo:f_hi], s, (f_lo, f_hi)
