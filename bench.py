#!/usr/bin/env python3
"""bench.py -- STFT frames/s at n_fft=1024, hop=256, 48 kHz mono on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path (framing -> detrend -> Hann window -> real FFT -> one-sided
density PSD) over one batch of synthetic clips already resident in HBM: BASELINE configs[1],
64 clips x 10 s x 48 kHz f32 per GPU (119 808 frames, 368.7 MB of algorithmic traffic per step).
For N > 1 the driver launches one rank per GPU with torch.distributed.run; clips shard across
ranks with no data-path collective (weak scaling: every rank owns its own 64-clip batch;
``--scaling strong`` shards ONE 64-clip batch instead); RCCL is used only for the barrier, the
max-over-ranks time and -- after the timed region, reported separately as ``gather`` -- the
gather of a reduced product (per-frame band power) to rank 0, which is the one exchange the path
has (``--gather-full`` also times the full spectra).  Rank 0 prints ONE JSON line.

What bounds the kernel is part of the line: ``roofline`` is the HBM roof (the graded fraction, priced with the
SAME clock as ``value``: the host clock around the barrier-bracketed region; the HIP-event time of the same region
is reported beside it under its own name), ``roofline.valu`` the VALU-issue roof, ``fp32_flops`` the arithmetic
rate, ``power`` the board power, power cap and shader clock read from the card's hwmon sensors during an untimed
leg of back-to-back launches after the timed region, and ``limiter`` what an untimed A/B of the same launch on
zero-filled inputs says (MI355X_MICROARCH.md, DVFS give-back item 1: zero data frees the clock; if the launch does
not get faster with it, neither the power cap nor the issue rate is what bounds it).  DESIGN.md section 5.

With ``--gpus N`` (N > 1) and no torch.distributed environment the script starts its own N ranks
(``python -m torch.distributed.run --nproc-per-node N``) as a child process before touching the GPU and relays
rank 0's line; under a launcher whose WORLD_SIZE differs from ``--gpus`` it exits non-zero.

Defaults (500 timed steps after 100 warm-up steps, ~60 ms of GPU time) are long enough to get past the chip's
power-management transient: the first ~12 launches run at 92 us, the next ~100 at up to 138 us, then the
clock settles (profiles/r01e_kernel_trace_durations.txt); shorter runs measure the transient, not the kernel.
For that reason ~60 ms of untimed launches (--settle-ms) precede the W warm-up steps whatever W and K are.

PyTorch is plumbing here (device memory, streams, torch.distributed); the measured work is
sg_stft from libspectro.so, called through the C ABI on torch's current stream.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spectrogram-generator_amd"))

FS, NPERSEG, HOP, N_SAMPLES, CLIPS_PER_GPU = 48000.0, 1024, 256, 480000, 64
N_BINS = NPERSEG // 2 + 1
BYTES_PER_FRAME = HOP * 4 + N_BINS * 4          # each sample read once, each bin written once
HBM_PEAK_GBS = 8000.0                           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_BUFFER_SETS = 4                               # rotate so the 256 MiB Infinity Cache cannot hold a step's data
# Arithmetic of one frame.  ALGORITHMIC: real FFT 2.5 n log2 n = 25 600 + detrend/window 3 n + epilogue
# 4 (n/2 + 1) = 30 724 flop.  EXECUTED by stft1024_r8x3_kernel per wave = per frame (instruction mix of the frame loop,
# profiles/r02_isa_mix_r8x3.txt, from `hipcc -S`): 320 VALU instructions = 366 issue slots (v_pk_add_f32 and v_mov_b64
# hold the SIMD for two), 101 of them FMAs -> 101*2 + 163 + 32*2 = 429 flop per lane x 64 lanes.
FLOP_PER_FRAME_ALGORITHMIC = 30724
FLOP_PER_FRAME_EXECUTED = 429 * 64
VALU_ISSUE_SLOTS_PER_FRAME = 366
CYCLES_PER_SLOT = 2                              # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md)
SCLK_MAX_MHZ = 2400.0
FP32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs (= ranks) of this node; default: WORLD_SIZE under a launcher, else 1.  Given explicitly, a launcher whose "
                         "WORLD_SIZE differs is refused")
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--clips", type=int, default=CLIPS_PER_GPU, help="clips per GPU (default: BASELINE cfg2 = 64)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank owns a --clips batch; strong: ONE --clips batch is sharded over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-mode", action="store_true", help="skip the untimed reference-mode (Tukey, hop 896; f32 and f64) leg")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU baseline work")
    ap.add_argument("--kernel", default=None, help="force a kernel family (debug): r8x3 | stockham")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed back-to-back launches before the warm-up steps, so that the power-management transient "
                         "(first ~200 launches) is over whatever --warmup/--steps are (0 disables)")
    ap.add_argument("--telemetry-s", type=float, default=1.5,
                    help="seconds of untimed back-to-back launches AFTER the timed region during which rank 0 reads board "
                         "power and shader clock from the card's hwmon nodes (0 disables)")
    ap.add_argument("--no-limiter-leg", action="store_true", help="skip the untimed zero-filled-input A/B the `limiter` entry is derived from")
    ap.add_argument("--master-port", type=int, default=0, help="--gpus N > 1 without a launcher: rendezvous port of the ranks this script starts (0: pick a free one)")
    ap.add_argument("--dry-run-spawn", action="store_true", help="print the launcher command --gpus N would start, as JSON, and exit (no GPU needed)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the (untimed-region) gather measurement")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the untimed `secondary` legs (cfg3 fused mel, cfg4 sweep, cfg5 streaming, the >= 4 GB batch; about 3 s)")
    ap.add_argument("--gather-full", action="store_true", help="N > 1: also time the gather of the full spectra to rank 0")
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


class Telemetry:
    """Board power / shader clock of ONE card (by PCI address) from its amdgpu hwmon nodes, polled from a thread."""

    KEYS = ("freq1_input", "power1_average", "power1_input", "power1_cap")

    def __init__(self, pci_bus_id: str, period: float = 0.02):
        self.period, self.samples, self._stop, self._th = period, [], threading.Event(), None
        self.paths = {}
        for hw in glob.glob(f"/sys/bus/pci/devices/{pci_bus_id}/hwmon/hwmon*"):
            for k in self.KEYS:
                try:
                    with open(f"{hw}/{k}") as fh:
                        fh.read()
                    self.paths[k] = f"{hw}/{k}"
                except OSError:
                    pass

    @staticmethod
    def _read(p):
        try:
            with open(p) as fh:
                return float(fh.read().strip())
        except (OSError, ValueError):
            return None

    def start(self):
        def poll():
            while not self._stop.is_set():
                self.samples.append((time.perf_counter(), {k: self._read(p) for k, p in self.paths.items()}))
                time.sleep(self.period)
        if self.paths:
            self._th = threading.Thread(target=poll, daemon=True)
            self._th.start()
        return self

    def stop(self, skip_s: float):
        """-> dict of medians over the samples taken after the first ``skip_s`` seconds (sensor averaging window), or None"""
        if self._th is None:
            return None
        self._stop.set()
        self._th.join()
        if not self.samples:
            return None
        t0 = self.samples[0][0] + skip_s
        rows = [s for t, s in self.samples if t >= t0] or [s for _, s in self.samples]
        pkey = "power1_average" if rows[0].get("power1_average") is not None else "power1_input"

        def med(key, div):
            v = [r[key] / div for r in rows if r.get(key) is not None]
            return statistics.median(v) if v else None
        return {"board_W": med(pkey, 1e6), "cap_W": med("power1_cap", 1e6), "sclk_MHz": med("freq1_input", 1e6),
                "n_samples": len(rows)}


def spawn_command(args, argv):
    """The launcher command for --gpus N > 1 when this process is not already a rank: one rank per GPU on this node."""
    port = args.master_port
    if not port:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def spawn_ranks(args, argv):
    """Start the ranks as a CHILD process (this parent has not touched the GPU: no torch import yet) and relay rank 0's line."""
    import subprocess
    cmd = spawn_command(args, argv)
    if args.dry_run_spawn:
        print(json.dumps({"spawn": cmd, "n_ranks": args.gpus}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"[bench] the {args.gpus}-rank child run failed (exit code {proc.returncode}, JSON line {'found' if line else 'missing'})", file=sys.stderr)
        return proc.returncode or 1
    if json.loads(line).get("n_gpus") != args.gpus:
        print(f"[bench] the child run reports n_gpus={json.loads(line).get('n_gpus')}, wanted {args.gpus}", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def main():
    args = parse()
    in_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus is None:                                  # `torchrun --nproc-per-node=8 bench.py`: the launcher's world is the answer
        args.gpus = int(os.environ["WORLD_SIZE"]) if in_launcher else 1
    if args.dry_run_spawn or (args.gpus > 1 and not in_launcher):
        raise SystemExit(spawn_ranks(args, [a for a in sys.argv[1:] if a != "--dry-run-spawn"]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a line "
                         f"whose n_gpus is not the one asked for")
    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    # Rehearsal switch for a 1-GPU box: SPECTRO_BENCH_SAME_GPU=1 puts every rank on cuda:0 and uses gloo for the
    # barrier / max-reduce (RCCL refuses two ranks on one device).  The driver's real runs use nccl, one GPU per rank.
    same_gpu = os.environ.get("SPECTRO_BENCH_SAME_GPU") == "1"
    if same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SPECTRO_BENCH_FORCE_DIST=1 under a launcher with WORLD_SIZE=1: the process group, the barrier, the all_reduce(MAX) of the
    # times and the gather run through RCCL with one rank -- the N > 1 code path on a one-GPU box (tests/test_gpu_multiproc.py)
    use_dist = world > 1 or (in_launcher and os.environ.get("SPECTRO_BENCH_FORCE_DIST") == "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from spectro import _capi
    from spectro import dist as sdist
    from spectro.windows import get_window
    _capi.ensure_device(local_rank)

    if args.scaling == "strong":
        c0, c1 = sdist.shard_range(args.clips, world, rank)
        n_clips = c1 - c0
    else:
        n_clips = args.clips
    plan = _capi.Plan(NPERSEG, NPERSEG, HOP, get_window("hann", NPERSEG), _capi.DETREND["constant"], FS,
                      _capi.SCALING["density"], _capi.MODE["psd"], _capi.F32)
    if args.kernel:
        plan.force_kernel(args.kernel)
    n_frames = plan.n_frames(N_SAMPLES)
    frames_per_step = n_clips * n_frames                       # this rank
    total_frames_per_step = (args.clips if args.scaling == "strong" else args.clips * world) * n_frames

    # synthetic input: default_rng(1234 + rank) white noise * 0.1, f32, resident in HBM
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    xs = [torch.randn((max(n_clips, 1), N_SAMPLES), device=dev, dtype=torch.float32, generator=gen) * 0.1
          for _ in range(N_BUFFER_SETS)]
    outs = [torch.empty((max(n_clips, 1), n_frames, N_BINS), device=dev, dtype=torch.float32) for _ in range(N_BUFFER_SETS)]
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
    # (the process's first event record costs ~65 us of one-time set-up -- tools/bench_overhead.py; keep it out of the timed region)
    _e0, _e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _e0.record()
    _e1.record()
    torch.cuda.synchronize(dev)
    _e0.elapsed_time(_e1)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()                                      # instrumentation, not workload: ~6 us of host time the idle GPU would wait through
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    ev1.record()
    while not ev1.query():                            # poll instead of sleeping in the driver: the blocking synchronize below wakes
        pass                                          # up tens of us late, which a 20-step region would book as kernel time
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize(dev)                   # (RCCL's barrier is device work; with one rank there is nothing more to wait for)
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)                    # HIP events on the launch stream

    t = torch.tensor([elapsed, dev_ms], device="cpu" if same_gpu else dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max, dev_ms_max = float(t[0]), float(t[1])

    # cheap sanity on the last output: finite and positive energy (full parity lives in tests/)
    last = outs[(args.steps - 1) % N_BUFFER_SETS]
    ok = n_clips == 0 or (bool(torch.isfinite(last).all()) and float(last.sum()) > 0)
    if not ok:
        raise SystemExit("bench output is not finite/positive")

    # ---- untimed legs -------------------------------------------------------------------------------------------
    power = None
    if rank == 0 and args.telemetry_s > 0 and n_clips > 0:
        try:
            tel = Telemetry(_capi.device_pci_bus_id()).start()
            t_tel, i = time.perf_counter(), 0
            while time.perf_counter() - t_tel < args.telemetry_s:
                for _ in range(64):
                    step(i)
                    i += 1
                torch.cuda.synchronize(dev)
            power = tel.stop(skip_s=min(0.6, args.telemetry_s / 2))
            if power:
                power["us_per_launch_during_sample"] = (time.perf_counter() - t_tel) * 1e6 / i
                power["sample"] = (f"{args.telemetry_s:g} s of back-to-back launches after the timed region; hwmon of "
                                   f"{_capi.device_pci_bus_id()}, medians after the sensor's first 0.6 s")
        except Exception as e:                                    # sensors are a report, never a failure
            print(f"[bench] telemetry unavailable: {e}", file=sys.stderr)

    # what bounds the launch: the same binary on zero-filled inputs, back to back with the random ones, each with its own clock reading
    limiter = None
    if rank == 0 and n_clips > 0 and not args.no_limiter_leg:
        try:
            limiter = time_limiter_leg(_capi, plan, xs, outs, n_clips, n_frames, dev, stream)
        except Exception as e:
            print(f"[bench] limiter leg failed: {e}", file=sys.stderr)

    # the reference's literal call (PlotEngine.py:113: Tukey(0.25), hop 896) on the same clips, f32 and -- what the reference's
    # loaders hand over -- f64: reported beside cpu_baseline.reference_mode, never part of `value`
    ref_mode = None
    if rank == 0 and world == 1 and n_clips > 0 and not args.no_reference_mode:
        try:
            ref_mode = time_reference_mode(_capi, get_window, xs, n_clips, dev, stream)
        except Exception as e:
            print(f"[bench] reference-mode leg failed: {e}", file=sys.stderr)

    # BASELINE's other configs and SURVEY H7's >= 4 GB batch, untimed legs after the headline region (never part of `value`)
    secondary = None
    if rank == 0 and world == 1 and n_clips == CLIPS_PER_GPU and not args.no_secondary:
        try:
            secondary = time_secondary(_capi, get_window, xs, dev, stream)
        except Exception as e:
            print(f"[bench] secondary legs failed: {type(e).__name__}: {e}", file=sys.stderr)

    gather = None
    if use_dist and not args.no_gather:
        gather = time_gather(args, plan, xs[0], outs[0], n_clips, n_frames, dev, same_gpu, world, rank, stream)

    if rank == 0:
        # ONE clock for the graded figures: the host clock around the barrier + synchronize bracketed region (the contract's clock)
        # prices `value`, `ms_per_step` and `roofline.achieved / frac`; the HIP-event time of the same region is named as such.
        value = total_frames_per_step * args.steps / elapsed_max
        launch_s = elapsed_max / args.steps
        launch_s_events = dev_ms_max / 1e3 / args.steps
        per_gpu_fps = frames_per_step / launch_s
        achieved = per_gpu_fps * BYTES_PER_FRAME / 1e9
        sclk = (power or {}).get("sclk_MHz") or SCLK_MAX_MHZ
        n_simd = _capi.device_info()["compute_units"] * 4
        slots_per_s = per_gpu_fps * VALU_ISSUE_SLOTS_PER_FRAME
        valu = {
            "issue_slots_per_frame": VALU_ISSUE_SLOTS_PER_FRAME, "cycles_per_slot": CYCLES_PER_SLOT,
            "achieved_slots_per_s": slots_per_s,
            "peak_slots_per_s_at_sclk": n_simd * sclk * 1e6 / CYCLES_PER_SLOT, "sustained_clk_MHz": sclk,
            "frac_at_sclk": slots_per_s / (n_simd * sclk * 1e6 / CYCLES_PER_SLOT),
            "frac_at_2400MHz": slots_per_s / (n_simd * SCLK_MAX_MHZ * 1e6 / CYCLES_PER_SLOT),
            "lane_ops_per_s": slots_per_s * 64, "peak_lane_ops_per_s": n_simd * 32 * sclk * 1e6,
        }
        hbm_frac = achieved / HBM_PEAK_GBS
        res = {
            "metric": "STFT frames/sec at n_fft=1024 hop=256, 48kHz mono; % HBM roofline",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max * 1e3 / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg2: {args.clips} clips x 10 s x 48 kHz f32 "
                                   f"{'per GPU' if args.scaling == 'weak' else 'in all, sharded over the ranks'}, "
                                   "n_fft=1024 hop=256 Hann, detrend=constant, one-sided density PSD (linear power), "
                                   "inputs resident in HBM",
                       "clips_per_gpu": n_clips, "frames_per_step_per_gpu": frames_per_step,
                       "sharding": "clips over ranks, no data-path collective", "kernel": plan.kernel,
                       "buffer_sets": N_BUFFER_SETS},
            # priced against the HBM roof (achieved / peak / frac are bytes); "valu" is the VALU issue roof, "limiter" what a
            # zero-input A/B of the same launch says about the clock's part in it
            "roofline": {"bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": hbm_frac, "traffic": load_traffic(),
                         "traffic_source": "profiles/traffic_latest.json: FETCH_SIZE x 2 + WRITE_SIZE per launch from a committed rocprofv3 "
                                           "--pmc run of this kernel on this workload (tools/profile_round.sh); NOT measured in this run",
                         "clock": "host clock around the barrier-bracketed timed region, the same one as `value` and `ms_per_step`",
                         "kernel": "stft1024_r8x3_kernel", "us_per_launch": launch_s * 1e6,
                         "us_per_launch_hip_events": launch_s_events * 1e6,
                         "frac_hip_events": frames_per_step / launch_s_events * BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_frame": BYTES_PER_FRAME,
                         "read_only_frac": per_gpu_fps * HOP * 4 / 1e9 / HBM_PEAK_GBS,
                         "practical_hbm_GBs_for_this_mix": 5500.0,      # profiles/r02_ubench_hbm_peaks.txt: 1 read : 2 write
                         "frac_of_practical": achieved / 5500.0,
                         "valu": valu},
            "fp32_flops": {"algorithmic_TFLOPs": per_gpu_fps * FLOP_PER_FRAME_ALGORITHMIC / 1e12,
                           "executed_TFLOPs": per_gpu_fps * FLOP_PER_FRAME_EXECUTED / 1e12,
                           "peak_TFLOPs": FP32_PEAK_TFLOPS, "flop_per_frame_algorithmic": FLOP_PER_FRAME_ALGORITHMIC,
                           "flop_per_frame_executed": FLOP_PER_FRAME_EXECUTED,
                           "frac_executed": per_gpu_fps * FLOP_PER_FRAME_EXECUTED / 1e12 / FP32_PEAK_TFLOPS},
            "pct_hbm_roofline": 100.0 * hbm_frac,
        }
        if power:
            res["power"] = power
            if power.get("board_W"):
                res["power"]["uJ_per_frame"] = power["board_W"] * power["us_per_launch_during_sample"] / max(frames_per_step, 1)
        if limiter:
            res["limiter"] = limiter
        if ref_mode:
            res["reference_mode"] = ref_mode
        if secondary:
            res["secondary"] = secondary
        if gather:
            res["gather"] = gather
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_baseline import time_cpu_baseline
            clips = (np.random.default_rng(1234).standard_normal((CLIPS_PER_GPU, N_SAMPLES)).astype(np.float32)
                     * np.float32(0.1))
            res["cpu_baseline"] = time_cpu_baseline(clips, FS, NPERSEG, HOP, budget_s=args.cpu_budget)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


def time_limiter_leg(_capi, plan, xs, outs, n_clips, n_frames, dev, stream, secs=0.5):
    """Untimed A/B after the headline region: the SAME launch on the random clips and on zero-filled copies of them, alternating, `secs`
    of back-to-back launches each, with the card's shader clock and board power sampled during each leg.  Zero data draws less power, so
    the clock rises to its maximum; a launch bounded by the power cap or by instruction issue speeds up with the clock, one bounded by
    the memory system does not.  -> dict (per-leg figures + the reading derived from them)."""
    import torch
    zeros = [torch.zeros_like(x) for x in xs[:2]]
    legs = {"random": xs, "zeros": zeros}
    rows = {k: [] for k in legs}
    for _ in range(2):
        for name, ins in legs.items():
            def step(i, ins=ins):
                b = i % len(ins)
                plan.stft(ins[b].data_ptr(), N_SAMPLES, N_SAMPLES, n_clips, outs[b].data_ptr(), n_frames * N_BINS, stream=stream)
            tel = Telemetry(_capi.device_pci_bus_id()).start()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < secs * 0.6:            # settle
                for _ in range(32):
                    step(n)
                    n += 1
                torch.cuda.synchronize(dev)
            ev0.record()
            m, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < secs * 0.4:
                for _ in range(32):
                    step(m)
                    m += 1
            ev1.record()
            torch.cuda.synchronize(dev)
            pw = tel.stop(skip_s=secs * 0.3) or {}
            rows[name].append({"us_per_launch": ev0.elapsed_time(ev1) * 1e3 / m, "sclk_MHz": pw.get("sclk_MHz"), "board_W": pw.get("board_W")})
    del zeros

    def med(name, key):
        v = [r[key] for r in rows[name] if r.get(key) is not None]
        return statistics.median(v) if v else None
    out = {"method": "same launch on random and on zero-filled inputs, alternating, 2 x %.1f s each, HIP events; hwmon medians per leg" % secs,
           "random": {k: med("random", k) for k in ("us_per_launch", "sclk_MHz", "board_W")},
           "zeros": {k: med("zeros", k) for k in ("us_per_launch", "sclk_MHz", "board_W")}}
    ur, uz = out["random"]["us_per_launch"], out["zeros"]["us_per_launch"]
    cr, cz = out["random"]["sclk_MHz"], out["zeros"]["sclk_MHz"]
    out["time_ratio_zeros_over_random"] = uz / ur
    if cr and cz:
        out["clock_ratio_zeros_over_random"] = cz / cr
        # share of the launch time that scales with the shader clock: t = t_fixed + t_clk / f  ->  (1 - uz/ur) / (1 - cr/cz)
        out["clock_bound_share"] = max(0.0, min(1.0, (1.0 - uz / ur) / (1.0 - cr / cz))) if cz > cr * 1.02 else None
    share = out.get("clock_bound_share")
    if share is None:
        out["reading"] = "the clock did not move between the legs: no statement"
    elif share < 0.35:
        out["reading"] = ("mostly NOT clock-bound: with the clock %.0f %% higher on zero data the launch is only %.1f %% shorter, so neither the "
                          "power cap nor instruction issue sets most of its time; the memory side and the launch's ramp and uneven finish do "
                          "(in-kernel stamps: profiles/r03_limiter.txt)" % ((cz / cr - 1) * 100, (1 - uz / ur) * 100))
    else:
        out["reading"] = ("clock-bound share %.2f: with the clock %.0f %% higher on zero data the launch is %.1f %% shorter -- the power cap "
                          "(through the clock it allows) sets a large part of its time" % (share, (cz / cr - 1) * 100, (1 - uz / ur) * 100))
    return out


def time_secondary(_capi, get_window, xs, dev, stream):
    """BASELINE configs 3, 4 and 5 and SURVEY H7's >= 4 GB batch as UNTIMED legs after the headline region (about 3 s in all; the
    headline's `value` / `ms_per_step` / `roofline` never see them).  Same clips as the headline leg (64 x 480 000 f32, the four
    rotating input sets), HIP events on the launch stream, outputs rotating through a pool larger than the Infinity Cache.
    tools/bench_extra.py is the long form of the same measurements (profiles/r0N_extra_*.json)."""
    import numpy as np
    import torch
    from spectro.mel import MelBank
    from spectro.stream import StreamingSTFT
    n_clips = xs[0].shape[0]
    out = {"note": "untimed legs after the headline region; inputs resident in HBM (cfg5: host to host), rotating buffers, HIP events"}

    def timed(fn, min_iters, settle_s=0.08, budget_s=0.12):
        """us per call: `settle_s` of back-to-back calls, then at least `min_iters` calls between two events"""
        k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < settle_s:
            for _ in range(4):
                fn(k)
                k += 1
            torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n, t0 = 0, time.perf_counter()
        ev0.record()
        while n < min_iters or time.perf_counter() - t0 < budget_s:
            for _ in range(4):
                fn(k)
                k += 1
                n += 1
            if n >= 4096:
                break
        ev1.record()
        torch.cuda.synchronize(dev)
        return ev0.elapsed_time(ev1) * 1e3 / n

    # ---- cfg3: the same batch + 80-band mel, fused (the linear spectrum never reaches HBM): hop*4 + 80*4 = 1344 B per frame
    plan = _capi.Plan(NPERSEG, NPERSEG, HOP, get_window("hann", NPERSEG), _capi.DETREND["constant"], FS, _capi.SCALING["density"], _capi.MODE["psd"], _capi.F32)
    nfr = plan.n_frames(N_SAMPLES)
    bank = MelBank(NPERSEG, FS, 80, 0.0, FS / 2)
    mels = [torch.empty((n_clips, nfr, 80), device=dev, dtype=torch.float32) for _ in range(2)]
    us = timed(lambda i: bank.stft_mel_ptr(plan, xs[i % len(xs)].data_ptr(), N_SAMPLES, N_SAMPLES, n_clips, mels[i % 2].data_ptr(), nfr * 80, True, stream=stream), 64)
    out["cfg3_fused_mel"] = {"us": us, "frames": n_clips * nfr, "frames_per_s": n_clips * nfr / us * 1e6, "bytes_per_frame": HOP * 4 + 80 * 4,
                             "achieved_GBs": n_clips * nfr * (HOP * 4 + 80 * 4) / us / 1e3, "n_mels": 80, "log": True,
                             "kernel": "stft1024_r8x3_kernel OUT_MEL (band-sparse epilogue)"}
    bank.close()
    plan.close()
    del mels

    # ---- cfg4: n_fft x hop sweep, this GPU's 64-clip share of the 256-clip job: spectra, and the fused band power the sweep gathers
    n_ffts, hops = (256, 512, 1024, 2048, 4096), (64, 128, 256)
    biggest = max(n_clips * ((N_SAMPLES - n) // h + 1) * (n // 2 + 1) for n in n_ffts for h in hops)
    pool = torch.empty(2 * biggest, device=dev, dtype=torch.float32)           # two output sets of the largest shape (2 x 3.9 GB)
    bp = torch.empty(n_clips * ((N_SAMPLES - 256) // 64 + 1), device=dev, dtype=torch.float32)
    per_pair, total, shared, band_total, band_shared = [], 0.0, 0.0, 0.0, 0.0
    for n in n_ffts:
        for h in hops:
            p = _capi.Plan(n, n, h, get_window("hann", n), _capi.DETREND["constant"], FS, _capi.SCALING["density"], _capi.MODE["psd"], _capi.F32)
            nf, nb = p.n_frames(N_SAMPLES), n // 2 + 1
            sz = n_clips * nf * nb
            n_out = max(2, min(8, int(2 * biggest // sz)))                   # small shapes rotate through more sets: past the cache
            est = 1e-9 * n_clips * nf * (n / 256) * 1.2                       # s per launch, roughly
            iters = max(8, min(256, int(0.03 / est)))
            us = timed(lambda i: p.stft(xs[i % len(xs)].data_ptr(), N_SAMPLES, N_SAMPLES, n_clips, pool[(i % n_out) * sz:].data_ptr(), nf * nb, stream=stream),
                       iters, settle_s=0.03, budget_s=0.0)
            us_b = timed(lambda i: p.band_power(xs[i % len(xs)].data_ptr(), N_SAMPLES, N_SAMPLES, n_clips, 1, n // 4, bp.data_ptr(), nf, stream=stream),
                         iters, settle_s=0.03, budget_s=0.0)
            bpf = h * 4 + nb * 4
            per_pair.append({"n_fft": n, "hop": h, "kernel": p.kernel, "frames": n_clips * nf, "us": us, "frames_per_s": n_clips * nf / us * 1e6,
                             "bytes_per_frame": bpf, "frac_of_hbm_peak": n_clips * nf * bpf / us / 1e3 / HBM_PEAK_GBS, "band_power_us": us_b})
            total += us
            band_total += us_b
            if h == min(hops):
                shared += us
                band_shared += us_b
            p.close()
    out["cfg4_sweep_64clips"] = {"total_ms": total / 1e3, "shared_hops_ms": shared / 1e3, "band_power_ms": band_total / 1e3,
                                 "band_power_shared_hops_ms": band_shared / 1e3, "per_pair": per_pair,
                                 "note": "15 (n_fft, hop) pairs on 64 clips x 480 000 samples; shared_hops: hops 128 / 256 taken as row subsets of the "
                                         "hop-64 transform (spectro.sweep.hop_families); band_power: the fused reduced product sharded_sweep gathers"}
    del pool, bp

    # ---- one launch over >= 4 GB (768 clips: 1.47 GB in, 2.95 GB out), the headline plan
    big_clips = 768
    plan = _capi.Plan(NPERSEG, NPERSEG, HOP, get_window("hann", NPERSEG), _capi.DETREND["constant"], FS, _capi.SCALING["density"], _capi.MODE["psd"], _capi.F32)
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    xb = torch.randn((big_clips, N_SAMPLES), device=dev, dtype=torch.float32, generator=gen) * 0.1
    ob = torch.empty((big_clips, nfr, N_BINS), device=dev, dtype=torch.float32)
    us = timed(lambda i: plan.stft(xb.data_ptr(), N_SAMPLES, N_SAMPLES, big_clips, ob.data_ptr(), nfr * N_BINS, stream=stream), 12, settle_s=0.1, budget_s=0.0)
    fr = big_clips * nfr
    out["large_batch"] = {"clips": big_clips, "GB": (xb.numel() + ob.numel()) * 4 / 1e9, "frames": fr, "us": us, "frames_per_s": fr / us * 1e6,
                          "frac": fr * BYTES_PER_FRAME / us / 1e3 / HBM_PEAK_GBS, "bytes_per_frame": BYTES_PER_FRAME}
    plan.close()
    del xb, ob

    # ---- the reference's non-power-of-two nperseg (GUI.py:87-89): nperseg 1000, hop 250 and the reference's own hop 875, register chirp-z kernel
    np2 = {}
    for hop_np2 in (250, 875):
        pn = _capi.Plan(1000, 1000, hop_np2, get_window("hann" if hop_np2 == 250 else ("tukey", 0.25), 1000), _capi.DETREND["constant"], FS,
                        _capi.SCALING["density"], _capi.MODE["psd"], _capi.F32)
        nf = pn.n_frames(N_SAMPLES)
        o2 = [torch.empty((n_clips, nf, 501), device=dev, dtype=torch.float32) for _ in range(2)]
        us = timed(lambda i: pn.stft(xs[i % len(xs)].data_ptr(), N_SAMPLES, N_SAMPLES, n_clips, o2[i % 2].data_ptr(), nf * 501, stream=stream), 24, settle_s=0.05, budget_s=0.0)
        np2[f"hop_{hop_np2}"] = {"kernel": pn.kernel, "frames": n_clips * nf, "us": us, "frames_per_s": n_clips * nf / us * 1e6,
                                 "bytes_per_frame": hop_np2 * 4 + 501 * 4, "frac_of_hbm_peak": n_clips * nf * (hop_np2 * 4 + 501 * 4) / us / 1e3 / HBM_PEAK_GBS}
        pn.close()
        del o2
    out["nperseg_1000"] = np2
    # ... and above 2048: nperseg 4000 at hop 1000 and at the reference's own hop 3500 (two wavefronts per frame, stft_rbluew.hip)
    np2w = {}
    for hop_np2 in (1000, 3500):
        pn = _capi.Plan(4000, 4000, hop_np2, get_window("hann" if hop_np2 == 1000 else ("tukey", 0.25), 4000), _capi.DETREND["constant"], FS,
                        _capi.SCALING["density"], _capi.MODE["psd"], _capi.F32)
        nf = pn.n_frames(N_SAMPLES)
        o2 = [torch.empty((n_clips, nf, 2001), device=dev, dtype=torch.float32) for _ in range(2)]
        us = timed(lambda i: pn.stft(xs[i % len(xs)].data_ptr(), N_SAMPLES, N_SAMPLES, n_clips, o2[i % 2].data_ptr(), nf * 2001, stream=stream), 24, settle_s=0.05, budget_s=0.0)
        np2w[f"hop_{hop_np2}"] = {"kernel": pn.kernel, "frames": n_clips * nf, "us": us, "frames_per_s": n_clips * nf / us * 1e6,
                                  "bytes_per_frame": hop_np2 * 4 + 2001 * 4, "frac_of_hbm_peak": n_clips * nf * (hop_np2 * 4 + 2001 * 4) / us / 1e3 / HBM_PEAK_GBS}
        pn.close()
        del o2
    out["nperseg_4000"] = np2w
    # ... and the reference's OWN call on its own dtype across the spin box (float64 recordings, SweepManager.py:135-136; Tukey, hop n - n // 8,
    # PlotEngine.py:113): 32 clips, one kernel family per size
    ref64 = {}
    c64 = min(32, n_clips)
    x64 = xs[0][:c64].double()
    for n64 in (64, 1024, 4000, 8192):
        hop64, nb64 = n64 - n64 // 8, n64 // 2 + 1
        pn = _capi.Plan(n64, n64, hop64, get_window(("tukey", 0.25), n64), _capi.DETREND["constant"], FS, _capi.SCALING["density"], _capi.MODE["psd"], _capi.F64)
        nf = pn.n_frames(N_SAMPLES)
        o2 = [torch.empty((c64, nf, nb64), device=dev, dtype=torch.float64) for _ in range(2)]
        us = timed(lambda i: pn.stft(x64.data_ptr(), N_SAMPLES, N_SAMPLES, c64, o2[i % 2].data_ptr(), nf * nb64, stream=stream), 16, settle_s=0.05, budget_s=0.0)
        ref64[f"nperseg_{n64}"] = {"kernel": pn.kernel, "hop": hop64, "clips": c64, "frames": c64 * nf, "us": us, "frames_per_s": c64 * nf / us * 1e6,
                                   "bytes_per_frame": (hop64 + nb64) * 8, "frac_of_hbm_peak": c64 * nf * (hop64 + nb64) * 8 / us / 1e3 / HBM_PEAK_GBS}
        pn.close()
        del o2
    del x64
    out["f64_reference_call"] = ref64

    # ---- cfg5: streaming 8 ch x 96 kHz, n_fft 4096 hop 1024, 4096-sample chunks, synchronous feed() host to host (PCIe inclusive)
    st = StreamingSTFT(8, 96000.0, 4096, 1024, window="hann")
    chunk = (np.random.default_rng(5).standard_normal((8, 4096)) * 0.1).astype(np.float32)
    for _ in range(8):
        st.feed(chunk)
    lat, frames, t0 = [], 0, time.perf_counter()
    for _ in range(300):
        c0 = time.perf_counter()
        _, sx = st.feed(chunk)
        lat.append(time.perf_counter() - c0)
        frames += sx.shape[-1] * 8
    dt = time.perf_counter() - t0
    st.close()
    out["cfg5_streaming"] = {"chunks": 300, "chunk_ms_p50": float(np.median(lat) * 1e3), "chunk_ms_p99": float(np.percentile(lat, 99) * 1e3),
                             "frames_per_s": frames / dt, "realtime_factor": (300 * 4096 / 96000.0) / dt,
                             "note": "8 ch x 96 kHz, n_fft 4096 hop 1024, numpy chunk in -> numpy frames out per feed()"}
    torch.cuda.empty_cache()
    return out


def time_reference_mode(_capi, get_window, xs, n_clips, dev, stream, secs=0.4):
    """spectrogram(x, fs, nperseg=1024, scaling='density', mode='psd') as the reference calls it (window ('tukey', .25),
    noverlap = nperseg // 8 -> hop 896, detrend 'constant') on the bench's clips, f32 and f64, inputs resident; rotating buffer sets
    as in the headline leg.  HIP events on the launch stream."""
    import torch
    hop = NPERSEG - NPERSEG // 8
    out = {"call": "scipy.signal.spectrogram(x, fs, nperseg=1024, scaling='density', mode='psd') -> Tukey(0.25), hop 896, "
                   "detrend constant (PlotEngine.py:113); untimed leg after the headline region, inputs resident in HBM"}
    for name, tdt, code, isz in (("f32", torch.float32, _capi.F32, 4), ("f64", torch.float64, _capi.F64, 8)):
        plan = _capi.Plan(NPERSEG, NPERSEG, hop, get_window(("tukey", 0.25), NPERSEG), _capi.DETREND["constant"], FS,
                          _capi.SCALING["density"], _capi.MODE["psd"], code)
        nfr = plan.n_frames(N_SAMPLES)
        ins = xs if tdt == torch.float32 else [x.to(torch.float64) for x in xs[:2]]
        outs = [torch.empty((n_clips, nfr, N_BINS), device=dev, dtype=tdt) for _ in range(len(ins))]

        def step(i):
            b = i % len(ins)
            plan.stft(ins[b].data_ptr(), N_SAMPLES, N_SAMPLES, n_clips, outs[b].data_ptr(), nfr * N_BINS, stream=stream)
        for i in range(20):
            step(i)
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n, t0 = 0, time.perf_counter()
        ev0.record()
        while time.perf_counter() - t0 < secs:
            for _ in range(50):
                step(n)
                n += 1
            torch.cuda.synchronize(dev)
        ev1.record()
        torch.cuda.synchronize(dev)
        us = ev0.elapsed_time(ev1) * 1e3 / n
        bpf = hop * isz + N_BINS * isz
        out[name] = {"kernel": plan.kernel, "frames": n_clips * nfr, "us_per_launch": us, "frames_per_s": n_clips * nfr / us * 1e6,
                     "algorithmic_bytes_per_frame": bpf, "achieved_GBs": n_clips * nfr * bpf / us / 1e3,
                     "frac_of_hbm_peak": n_clips * nfr * bpf / us / 1e3 / HBM_PEAK_GBS, "buffer_sets": len(ins)}
        plan.close()
        del outs
        if tdt != torch.float32:
            del ins
    return out


def time_gather(args, plan, x, out, n_clips, n_frames, dev, same_gpu, world, rank, stream):
    """The path's one exchange, timed OUTSIDE the throughput region: every rank reduces its shard to the per-frame band
    power (fused kernel, [clips, frames] f32) and the shards are gathered -- the default product by ONE all_gather collective
    (RCCL's most ordinary call; shards padded to the largest), the full spectra [clips, frames, 513] (--gather-full) by direct
    peer sends to rank 0 (spectro.dist.gather_to_root).  -> dict for rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    from spectro import dist as sdist
    band = torch.zeros((max(n_clips, 1), n_frames), device=dev, dtype=torch.float32)
    if n_clips:
        plan.band_power(x.data_ptr(), N_SAMPLES, N_SAMPLES, n_clips, 0, N_BINS - 1, band.data_ptr(), n_frames, stream=stream)
    torch.cuda.synchronize(dev)
    res = {}

    def best_of(fn, reps=3):
        best = None
        for _ in range(reps):
            dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            got = fn()
            torch.cuda.synchronize(dev)
            tt = torch.tensor([time.perf_counter() - t0], device="cpu" if same_gpu else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            best = float(tt[0]) if best is None else min(best, float(tt[0]))
        return best, got

    # reduced product: all_gather of equal (padded) shards
    counts = [None] * world
    dist.all_gather_object(counts, int(n_clips))
    pad = max(max(counts), 1)
    send = torch.zeros((pad, n_frames), device="cpu" if same_gpu else dev, dtype=torch.float32)
    send[:n_clips] = band[:n_clips].to(send.device)
    recv = torch.empty((world * pad, n_frames), device=send.device, dtype=torch.float32)
    t_best, _ = best_of(lambda: dist.all_gather_into_tensor(recv, send))
    if rank == 0:
        nbytes = (world - 1) * pad * n_frames * 4
        ok = bool(torch.isfinite(recv).all()) and all(float(recv[r * pad:r * pad + c].abs().sum()) > 0 for r, c in enumerate(counts) if c)
        res["band_power"] = {"gather_ms": t_best * 1e3, "bytes_in_per_rank": nbytes, "gather_GBps": nbytes / t_best / 1e9 if t_best > 0 else None,
                             "collective": "all_gather_into_tensor", "values_ok": ok}
    if args.gather_full:
        sendf = out[:n_clips].cpu() if same_gpu else out[:n_clips]
        shapes = [None] * world
        dist.all_gather_object(shapes, [(tuple(sendf.shape), "float32")])
        t_best, got = best_of(lambda: sdist.gather_to_root([sendf], dst=0, shapes=shapes, checked=True))
        if rank == 0:
            nbytes = sum(int(torch.tensor(s[0][0]).prod()) * 4 for r, s in enumerate(shapes) if r != 0)
            ok = all(tuple(g[0].shape) == tuple(shapes[r][0][0]) for r, g in enumerate(got))
            res["full_spectra"] = {"gather_ms": t_best * 1e3, "bytes_to_root": nbytes, "gather_GBps": nbytes / t_best / 1e9 if t_best > 0 else None,
                                   "collective": "batched isend / irecv to rank 0", "shapes_ok": ok}
    if rank == 0:
        res["note"] = ("after the timed region, best of 3, max over ranks; xGMI is point-to-point: a one-shot gather is bounded by a "
                       "rank's inbound links, a ring gains nothing")
        return res
    return None


if __name__ == "__main__":
    main()
