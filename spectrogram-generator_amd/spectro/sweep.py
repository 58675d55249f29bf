"""Parameter sweep n_fft x hop over a batch of clips, sharded over the ranks of one node (BASELINE cfg4).

Default (batched) form: every rank owns a CONTIGUOUS run of clips for every ``(n_fft, hop)`` pair -- the remainders of
``n_clips / world`` rotate with the pair index so the ranks stay balanced -- uploads the clips it owns ONCE and runs
ONE batched device call per pair (``DeviceClips.band_log_power``: fused band power + log10, two launches), i.e. at most
``len(n_ffts) * len(hops)`` calls per rank instead of one launch and one upload per (clip, pair) item.  Plans (window /
twiddle tables) are replicated per GPU; there is no data-path collective.  What crosses xGMI at the end is a *reduced*
product per item (the per-frame log band power ``[n_frames]``, A11's fused kernel), gathered to the root with direct
peer sends (``gather_to_root``) -- a full-spectrum gather would dwarf the compute.

Per-item form (``compute=`` given, or ``batched=False``): work items ``(clip, n_fft, hop)`` sorted by cost
``frames * n log n`` and dealt longest-first (``spectro.dist.deal_work_items``), one call per item -- for reductions the
batched call does not offer, and the equality check of the batched path in the tests.

The reference's ``SweepManager`` is a file loader ("sweep" = recorded trial); this module is the batch API the
BASELINE config calls a "SweepManager parameter sweep" and has no reference behaviour beyond "each item equals the
offline spectrogram call on the same samples".
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np

from . import dist as sdist

__all__ = ["work_items", "clip_blocks", "sharded_sweep"]


def work_items(n_clips: int, n_samples: int, n_ffts: Sequence[int], hops: Sequence[int]):
    """All ``(clip, n_fft, hop)`` triples and their costs, in a deterministic order."""
    items, costs = [], []
    for n in n_ffts:
        for h in hops:
            c = sdist.stft_cost(n_samples, n, h)
            for clip in range(n_clips):
                items.append((clip, int(n), int(h)))
                costs.append(c)
    return items, costs


def clip_blocks(n_clips: int, n_ffts: Sequence[int], hops: Sequence[int], world: int):
    """Batched deal: for every pair (in ``work_items`` order) the clip range ``[start, stop)`` of each rank.

    Contiguous ranges in rank order, ``n_clips // world`` clips each; the ``n_clips % world`` extra clips go to the ranks
    ``(j + pair_index) % world``, so over the pairs every rank does the same work to within one clip of one pair and a
    rank's ranges for different pairs differ by at most that remainder (its clips are uploaded once, as their hull).
    -> ``{(n_fft, hop): [(start, stop)] * world}``"""
    base, rem = divmod(n_clips, world)
    out, p = {}, 0
    for n in n_ffts:
        for h in hops:
            extra = {(j + p) % world for j in range(rem)}
            ranges, start = [], 0
            for r in range(world):
                size = base + (1 if r in extra else 0)
                ranges.append((start, start + size))
                start += size
            out[(int(n), int(h))] = ranges
            p += 1
    return out


def _default_compute(clips, fs, fmin, fmax, window):
    from . import engine

    def run(clip, n_fft, hop):
        t, feats = engine.band_features(clips[clip], fs, n_fft, fmin, fmax, window=window, noverlap=n_fft - hop)
        return np.zeros(0, np.float32) if feats is None else np.ascontiguousarray(feats[:, 0], np.float32)
    return run


def _default_batch(fs, fmin, fmax, window):
    """``batch(host_clips[lo:hi]) -> runner(n_fft, hop, a, b) -> [b - a, n_frames] f32`` over clips ``a..b`` of that
    upload; clips cross PCIe once per rank"""
    from . import engine

    def open_batch(x):
        dev = engine.DeviceClips(x)

        def run(n_fft, hop, a, b):
            t, feats = dev.band_log_power(fs, n_fft, hop, fmin, fmax, window=window, clip_range=(a, b))
            if feats is None:
                return np.zeros((b - a, 0), np.float32)
            return np.ascontiguousarray(feats[..., 0], np.float32)
        run.close = dev.free
        return run
    return open_batch


def sharded_sweep(clips, fs: float, n_ffts: Sequence[int], hops: Sequence[int], fmin: float = 0.0, fmax: float = 1e9,
                  window="hann", compute: Callable | None = None, dst: int = 0, batched: bool = True,
                  batch_compute: Callable | None = None):
    """Run the sweep on this rank's share and gather the reduced results on ``dst``.

    ``clips``: ``[n_clips, n_samples]`` host array, identical on every rank (each rank only touches its own share).
    Batched (default): ``batch_compute(x_hull) -> run(n_fft, hop, a, b) -> [b - a, n_frames] f32`` over clips ``a..b`` of the
    hull it was given (default: log band power per frame through ``DeviceClips``); per item: ``compute(clip, n_fft, hop) -> 1-D float32`` (default: the same
    product, one call per item).  Returns on ``dst`` a dict ``{(clip, n_fft, hop): array}`` for ALL items, elsewhere None.
    """
    import torch
    clips = np.asarray(clips)
    world, rank = sdist.world_info()
    n_clips, n_samples = clips.shape
    if compute is not None or not batched:
        items, costs = work_items(n_clips, n_samples, n_ffts, hops)
        deal = sdist.deal_work_items(costs, world)
        run = compute or _default_compute(clips, fs, fmin, fmax, window)
        mine = [np.ascontiguousarray(run(*items[i]), np.float32) for i in deal[rank]]
        gathered = sdist.gather_to_root([torch.from_numpy(m) for m in mine], dst=dst)
        if rank != dst and world > 1:
            return None
        out = {}
        for r, part in enumerate(gathered):
            for i, t in zip(deal[r], part):
                out[items[i]] = t.numpy()
        return out

    blocks = clip_blocks(n_clips, n_ffts, hops, world)
    lo = min(b[rank][0] for b in blocks.values())
    hi = max(b[rank][1] for b in blocks.values())
    opener = batch_compute or _default_batch(fs, fmin, fmax, window)
    mine = []
    run = opener(clips[lo:hi]) if hi > lo else None             # ONE upload: the hull of this rank's clip ranges
    try:
        for pair, ranges in blocks.items():
            c0, c1 = ranges[rank]
            if c1 <= c0:
                mine.append(np.zeros((0, 0), np.float32))
                continue
            mine.append(np.ascontiguousarray(np.asarray(run(pair[0], pair[1], c0 - lo, c1 - lo), np.float32)))   # one batched call
    finally:
        if run is not None and hasattr(run, "close"):
            run.close()
    # every rank can derive every shape: frames follow from (n_samples, n_fft, hop), clips from the block table
    shapes = [[(ranges[r][1] - ranges[r][0], sdist.n_frames(n_samples, n, h)) if ranges[r][1] > ranges[r][0] else (0, 0)
               for (n, h), ranges in blocks.items()] for r in range(world)]
    gathered = sdist.gather_to_root([torch.from_numpy(m) for m in mine], dst=dst, shapes=shapes)
    if rank != dst and world > 1:
        return None
    out = {}
    for r, part in enumerate(gathered):
        for (pair, ranges), t in zip(blocks.items(), part):
            arr = t.numpy()
            for j, clip in enumerate(range(*ranges[r])):
                out[(clip, pair[0], pair[1])] = arr[j]
    return out
