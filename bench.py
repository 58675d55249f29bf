#!/usr/bin/env python3
"""bench.py -- STFT frames/s at n_fft=1024, hop=256, 48 kHz mono on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path (framing -> detrend -> Hann window -> real FFT -> one-sided
density PSD) over one batch of synthetic clips already resident in HBM: BASELINE configs[1],
64 clips x 10 s x 48 kHz f32 per GPU (119 808 frames, 368.7 MB of algorithmic traffic per step).
For N > 1 the driver launches one rank per GPU with torch.distributed.run; clips shard across
ranks with no data-path collective (weak scaling: every rank owns its own 64-clip batch); RCCL
is used only for the barrier and the max-over-ranks time.  Rank 0 prints ONE JSON line.

Defaults (500 timed steps after 100 warm-up steps, ~60 ms of GPU time) are long enough to get past the chip's
power-management transient: the first ~12 launches run at 92 us, the next ~100 at up to 138 us, then the
clock settles (profiles/r01e_kernel_trace_durations.txt); shorter runs measure the transient, not the kernel.
For that reason ~60 ms of untimed launches (--settle-ms) precede the W warm-up steps whatever W and K are.

PyTorch is plumbing here (device memory, streams, torch.distributed); the measured work is
sg_stft from libspectro.so, called through the C ABI on torch's current stream.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spectrogram-generator_amd"))

FS, NPERSEG, HOP, N_SAMPLES, CLIPS_PER_GPU = 48000.0, 1024, 256, 480000, 64
N_BINS = NPERSEG // 2 + 1
BYTES_PER_FRAME = HOP * 4 + N_BINS * 4          # each sample read once, each bin written once
HBM_PEAK_GBS = 8000.0                           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_BUFFER_SETS = 4                               # rotate so the 256 MiB Infinity Cache cannot hold a step's data


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--clips", type=int, default=CLIPS_PER_GPU, help="clips per GPU (default: BASELINE cfg2 = 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU baseline work")
    ap.add_argument("--kernel", default=None, help="force a kernel family (debug): r8x3 | stockham")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed back-to-back launches before the warm-up steps, so that the power-management transient "
                         "(first ~200 launches) is over whatever --warmup/--steps are (0 disables)")
    return ap.parse_args()


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        if d.get("workload_frames") == CLIPS_PER_GPU * ((N_SAMPLES - NPERSEG) // HOP + 1):
            return d.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    # Rehearsal switch for a 1-GPU box: SPECTRO_BENCH_SAME_GPU=1 puts every rank on cuda:0 and uses gloo for the
    # barrier / max-reduce (RCCL refuses two ranks on one device).  The driver's real runs use nccl, one GPU per rank.
    same_gpu = os.environ.get("SPECTRO_BENCH_SAME_GPU") == "1"
    if same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from spectro import _capi
    from spectro.windows import get_window
    _capi.ensure_device(local_rank)

    n_clips = args.clips
    plan = _capi.Plan(NPERSEG, NPERSEG, HOP, get_window("hann", NPERSEG), _capi.DETREND["constant"], FS,
                      _capi.SCALING["density"], _capi.MODE["psd"], _capi.F32)
    if args.kernel:
        plan.force_kernel(args.kernel)
    n_frames = plan.n_frames(N_SAMPLES)
    frames_per_step = n_clips * n_frames

    # synthetic input: default_rng(1234 + rank) white noise * 0.1, f32, resident in HBM
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    xs = [torch.randn((n_clips, N_SAMPLES), device=dev, dtype=torch.float32, generator=gen) * 0.1
          for _ in range(N_BUFFER_SETS)]
    outs = [torch.empty((n_clips, n_frames, N_BINS), device=dev, dtype=torch.float32) for _ in range(N_BUFFER_SETS)]
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(i):
        b = i % N_BUFFER_SETS
        plan.stft(xs[b].data_ptr(), N_SAMPLES, N_SAMPLES, n_clips, outs[b].data_ptr(), n_frames * N_BINS, stream=stream)

    if args.settle_ms > 0:                        # untimed: let the clocks settle (see module docstring)
        t_settle = time.perf_counter()
        i = 0
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for _ in range(16):
                step(i)
                i += 1
            torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(i)
    ev1.record()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)                    # HIP events on the launch stream

    t = torch.tensor([elapsed, dev_ms], device="cpu" if same_gpu else dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max, dev_ms_max = float(t[0]), float(t[1])

    # cheap sanity on the last output: finite and positive energy (full parity lives in tests/)
    last = outs[(args.steps - 1) % N_BUFFER_SETS]
    ok = bool(torch.isfinite(last).all()) and float(last.sum()) > 0
    if not ok:
        raise SystemExit("bench output is not finite/positive")

    if rank == 0:
        value = world * frames_per_step * args.steps / elapsed_max
        launch_s = dev_ms_max / 1e3 / args.steps
        achieved = frames_per_step * BYTES_PER_FRAME / launch_s / 1e9
        res = {
            "metric": "STFT frames/sec at n_fft=1024 hop=256, 48kHz mono; % HBM roofline",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg2: {n_clips} clips x 10 s x 48 kHz f32 per GPU, n_fft=1024 hop=256 Hann, "
                                   "detrend=constant, one-sided density PSD (linear power), inputs resident in HBM",
                       "clips_per_gpu": n_clips, "frames_per_step_per_gpu": frames_per_step,
                       "sharding": "clips over ranks, no data-path collective", "kernel": plan.kernel,
                       "buffer_sets": N_BUFFER_SETS},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(),
                         "kernel": "stft1024_r8x3_kernel", "us_per_launch": launch_s * 1e6,
                         "algorithmic_bytes_per_frame": BYTES_PER_FRAME,
                         "read_only_frac": frames_per_step * HOP * 4 / launch_s / 1e9 / HBM_PEAK_GBS},
            "pct_hbm_roofline": 100.0 * achieved / HBM_PEAK_GBS,
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_baseline import time_cpu_baseline
            clips = (np.random.default_rng(1234).standard_normal((CLIPS_PER_GPU, N_SAMPLES)).astype(np.float32)
                     * np.float32(0.1))
            res["cpu_baseline"] = time_cpu_baseline(clips, FS, NPERSEG, HOP, budget_s=args.cpu_budget)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
