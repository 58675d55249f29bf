"""Parameter sweep n_fft x hop over a batch of clips, sharded over the ranks of one node (BASELINE cfg4).

Default (batched) form: every rank owns a CONTIGUOUS run of clips for every ``(n_fft, hop)`` pair -- the remainders of
``n_clips / world`` rotate with the pair index so the ranks stay balanced -- uploads the clips it owns ONCE and runs
ONE batched device call per pair (``DeviceClips.band_log_power``: fused band power + log10, two launches), i.e. at most
``len(n_ffts) * len(hops)`` calls per rank instead of one launch and one upload per (clip, pair) item.  Hops of one n_fft that
divide each other share ONE transform at their gcd (``hop_families``: frame i at hop h IS frame i*h/g at hop g), so BASELINE cfg4's
15 pairs cost 5 transforms -- 1 unit of work per n_fft instead of 1 + 1/2 + 1/4.  Plans (window /
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

__all__ = ["work_items", "clip_blocks", "hop_families", "sweep_frames", "sharded_sweep"]


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


def hop_families(hops: Sequence[int]):
    """Hops that can share ONE transform per n_fft: frame ``i`` at hop ``h`` is frame ``i * h/g`` at hop ``g`` when ``g`` divides
    ``h`` (same samples, same arithmetic: bit-identical rows), so a family ``{h_i}`` is served by the hop-``g`` transform with
    ``g = gcd(h_i)`` and row subsampling -- cfg4's hops 64 / 128 / 256 cost 1 unit of work instead of 1 + 1/2 + 1/4.  A hop
    joins a family only while the shared transform stays cheaper than separate ones (``1/g <= sum 1/h_i``), and only if the
    family's gcd keeps the parity of its members: the register kernels of some plans (f64, nfft 256 / 512 / 2048 / 4096) serve
    even hops only and an odd hop runs on the LDS kernel, whose rows agree with theirs to rounding, not bit for bit -- a family
    must be one kernel's work for the rows to be IDENTICAL.
    -> ``[(g, [h, ...]), ...]``, deterministic."""
    from math import gcd
    rest = sorted({int(h) for h in hops})
    out = []
    while rest:
        fam = [rest.pop(0)]
        g = fam[0]
        for h in list(rest):
            g2 = gcd(g, h)
            if 1.0 / g2 <= sum(1.0 / m for m in fam) + 1.0 / h + 1e-12 and all(m % 2 == g2 % 2 for m in fam + [h]):
                fam.append(h)
                rest.remove(h)
                g = g2
        out.append((g, fam))
    return out


def _default_compute(clips, fs, fmin, fmax, window):
    from . import engine

    def run(clip, n_fft, hop):
        t, feats = engine.band_features(clips[clip], fs, n_fft, fmin, fmax, window=window, noverlap=n_fft - hop)
        return np.zeros(0, np.float32) if feats is None else np.ascontiguousarray(feats[:, 0], np.float32)
    return run


def sweep_frames(n_samples: int, n_fft: int, hop: int) -> int:
    """Frames of one sweep item as the device call produces them: ``nperseg`` is clamped to the clip length like scipy does
    (``_triage_segments``), the hop stays -- so ``n_fft > n_samples`` gives ONE frame, not none."""
    nps = min(int(n_fft), int(n_samples))
    return sdist.n_frames(n_samples, nps, hop) if nps > 0 else 0


def _default_batch(fs, fmin, fmax, window, on_device=False):
    """``batch(host_clips[lo:hi]) -> runner(n_fft, hop, a, b) -> [b - a, n_frames] f32`` over clips ``a..b`` of that
    upload; clips cross PCIe once per rank.  ``on_device``: the products stay in HBM as torch tensors (zero-copy views of the
    library's blocks) -- what a gather over RCCL sends; otherwise they come back as numpy arrays."""
    from . import engine

    def open_batch(x):
        dev = engine.DeviceClips(x)
        held = []                                           # device blocks behind tensors handed out

        def run(n_fft, hop, a, b):
            t, feats = dev.band_log_power(fs, n_fft, hop, fmin, fmax, window=window, clip_range=(a, b), keep_on_device=on_device)
            if feats is None:
                if on_device:
                    import torch
                    return torch.zeros((b - a, 0), dtype=torch.float32, device="cuda")
                return np.zeros((b - a, 0), np.float32)
            if on_device:
                import torch
                held.append(feats)
                return torch.as_tensor(feats, device="cuda")[..., 0].to(torch.float32).contiguous()
            return np.ascontiguousarray(feats[..., 0], np.float32)

        def family(n_fft, g, members):
            """``members = [(hop, a, b), ...]`` with ``g | hop``: one hop-``g`` transform over the hull of the clip ranges,
            every member's rows taken from it.  -> list of ``[b - a, n_frames(hop)]`` f32 in member order"""
            a0, b0 = min(m[1] for m in members), max(m[2] for m in members)
            base = run(n_fft, g, a0, b0)
            out = []
            for hop, a, b in members:
                nfr = sweep_frames(dev.n_samples, n_fft, hop)
                part = base[a - a0:b - a0, ::hop // g][:, :nfr]
                out.append(part.contiguous() if on_device else np.ascontiguousarray(part))
            return out

        def close():
            for h in held:
                h.free()
            held.clear()
            dev.free()
        run.family = family
        run.close = close
        return run
    return open_batch


def sharded_sweep(clips, fs: float, n_ffts: Sequence[int], hops: Sequence[int], fmin: float = 0.0, fmax: float = 1e9,
                  window="hann", compute: Callable | None = None, dst: int = 0, batched: bool = True,
                  batch_compute: Callable | None = None, share_hops: bool = True, device_products: bool | None = None):
    """Run the sweep on this rank's share and gather the reduced results on ``dst``.

    ``clips``: ``[n_clips, n_samples]`` host array, identical on every rank (each rank only touches its own share).
    Batched (default): ``batch_compute(x_hull) -> run(n_fft, hop, a, b) -> [b - a, n_frames] f32`` over clips ``a..b`` of the
    hull it was given (default: log band power per frame through ``DeviceClips``); per item: ``compute(clip, n_fft, hop) -> 1-D float32`` (default: the same
    product, one call per item).  ``share_hops`` (batched form, runners that offer ``family``): hops of one n_fft that divide
    each other are served by one transform at their gcd and row subsampling (``hop_families``) -- identical values, cfg4's
    15 pairs cost 5 transforms.  ``device_products`` (default runner): the reduced products stay on the GPU until the gather has
    moved them (default: yes under RCCL with more than one rank, where ``gather_to_root`` would otherwise stage host arrays
    back onto the device).  Returns on ``dst`` a dict ``{(clip, n_fft, hop): array}`` for ALL items, elsewhere None.
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
    if device_products is None:
        device_products = world > 1 and "nccl" in sdist._backend()
    on_device = bool(device_products) and batch_compute is None
    opener = batch_compute or _default_batch(fs, fmin, fmax, window, on_device=on_device)
    run = opener(clips[lo:hi]) if hi > lo else None             # ONE upload: the hull of this rank's clip ranges
    done = {}

    def as_tensor(v):
        return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32)))
    try:
        if run is not None and share_hops and hasattr(run, "family"):
            for n in dict.fromkeys(int(n) for n in n_ffts):
                for g, fam in hop_families(hops):
                    members = [(h, blocks[(n, h)][rank][0] - lo, blocks[(n, h)][rank][1] - lo) for h in fam
                               if blocks[(n, h)][rank][1] > blocks[(n, h)][rank][0]]
                    if len(members) > 1 and n <= n_samples:      # one transform at hop g for the whole family (a clamped
                        for (h, _, _), arr in zip(members, run.family(n, g, members)):     # nperseg shifts the hops: no sharing)
                            done[(n, h)] = as_tensor(arr)
        mine = []
        for pair, ranges in blocks.items():
            c0, c1 = ranges[rank]
            if c1 <= c0:
                mine.append(torch.zeros((0, 0), dtype=torch.float32))
            elif pair in done:
                mine.append(done[pair])
            else:
                mine.append(as_tensor(run(pair[0], pair[1], c0 - lo, c1 - lo)))   # one batched call
        # every rank can derive every shape: frames follow from (n_samples, n_fft, hop) -- with nperseg clamped to the clip length, as
        # the device call clamps it -- and clips from the block table.  A product of another shape would leave the root waiting for
        # bytes that never come (or a sender stuck): refuse it here, on every rank alike.
        shapes = [[((ranges[r][1] - ranges[r][0], sweep_frames(n_samples, n, h)) if ranges[r][1] > ranges[r][0] else (0, 0), torch.float32)
                   for (n, h), ranges in blocks.items()] for r in range(world)]
        bad = [(pair, (tuple(m.shape), m.dtype), want) for m, want, pair in zip(mine, shapes[rank], blocks) if (tuple(m.shape), m.dtype) != want]
        if not sdist.all_agree(not bad):                     # every rank leaves here together, whichever of them holds the odd product
            raise ValueError("sweep products do not have the shapes the gather expects" +
                             (f": item {bad[0][0]} has {bad[0][1]}, expected {bad[0][2]}" if bad else " (on another rank)"))
        gathered = sdist.gather_to_root(mine, dst=dst, shapes=shapes, checked=True)
    finally:
        if run is not None and hasattr(run, "close"):
            run.close()
    if rank != dst and world > 1:
        return None
    out = {}
    for r, part in enumerate(gathered):
        for (pair, ranges), t in zip(blocks.items(), part):
            arr = t.cpu().numpy()
            for j, clip in enumerate(range(*ranges[r])):
                out[(clip, pair[0], pair[1])] = arr[j]
    return out
