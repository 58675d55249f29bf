"""Parameter sweep n_fft x hop over a batch of clips, sharded over the ranks of one node (BASELINE cfg4).

Work items are ``(clip, n_fft, hop)`` triples.  They are sorted by cost ``frames * n log n`` and dealt to the ranks
(longest-processing-time first, ``spectro.dist.deal_work_items``); plans (window / twiddle tables) are replicated per
GPU; every rank runs its items on its own device with no data-path collective.  What crosses xGMI at the end is a
*reduced* product per item (default: the per-frame band power ``[n_frames]``, A11's fused kernel), gathered to the
root with direct peer sends (``gather_to_root``) -- a full-spectrum gather would dwarf the compute.

The reference's ``SweepManager`` is a file loader ("sweep" = recorded trial); this module is the batch API the
BASELINE config calls a "SweepManager parameter sweep" and has no reference behaviour beyond "each item equals the
offline spectrogram call on the same samples".
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np

from . import dist as sdist

__all__ = ["work_items", "sharded_sweep"]


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


def _default_compute(clips, fs, fmin, fmax, window):
    from . import engine

    def run(clip, n_fft, hop):
        t, feats = engine.band_features(clips[clip], fs, n_fft, fmin, fmax, window=window, noverlap=n_fft - hop)
        return np.zeros(0, np.float32) if feats is None else np.ascontiguousarray(feats[:, 0], np.float32)
    return run


def sharded_sweep(clips, fs: float, n_ffts: Sequence[int], hops: Sequence[int], fmin: float = 0.0, fmax: float = 1e9,
                  window="hann", compute: Callable | None = None, dst: int = 0):
    """Run the sweep on this rank's share and gather the reduced results on ``dst``.

    ``clips``: ``[n_clips, n_samples]`` host array, identical on every rank (each rank only touches its own items).
    ``compute(clip, n_fft, hop) -> 1-D float32 array`` is the per-item device call; default = log band power per frame
    via the fused STFT kernel.  Returns on ``dst`` a dict ``{(clip, n_fft, hop): array}`` for ALL items, elsewhere None.
    """
    import torch
    clips = np.asarray(clips)
    world, rank = sdist.world_info()
    items, costs = work_items(clips.shape[0], clips.shape[1], n_ffts, hops)
    deal = sdist.deal_work_items(costs, world)
    run = compute or _default_compute(clips, fs, fmin, fmax, window)
    mine = [np.ascontiguousarray(run(*items[i]), np.float32) for i in deal[rank]]
    tensors = [torch.from_numpy(m) for m in mine]
    # shapes are known from the item list only for the default reduction; exchange them to stay generic
    gathered = sdist.gather_to_root(tensors, dst=dst)
    if rank != dst and world > 1:
        return None
    out = {}
    for r, part in enumerate(gathered):
        for i, t in zip(deal[r], part):
            out[items[i]] = t.numpy()
    return out
