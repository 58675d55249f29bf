#!/usr/bin/env python3
"""What bounds stft1024_r8x3_kernel?  The three one-run experiments VERDICT r02 item 1 asks for, in one process on one box.

    python tools/limiter.py [--clips 64] [--secs 2.0] [--legs data,duty,big]

  data   the same binary on random and on zero-filled inputs, back to back, twice (MI355X_MICROARCH.md, DVFS give-back item 1):
         `--secs` of sustained launches per leg (board power and sclk sampled meanwhile), then 400 timed launches.
  duty   duty-cycled launches: one launch, then an idle gap of >= 2x its duration, 300 times.  Run the script under
         `rocprofv3 --kernel-trace` to get the per-dispatch durations from the tracer as well.
  big    a >= 4 GB-per-launch batch: 768 clips, two buffer sets.

With a diagnostic build (tools/build_variant.sh stamp ... -DSG_R8_STAMP=1; SPECTRO_LIB=.../lib_stamp/libspectro.so) every leg also
reads the kernel's own clock stamps (s_memtime / s_memrealtime per wave, DVFS give-back item 6): the in-kernel shader clock, the
launch's span from first wave start to last wave end, and how the waves' starts (ramp) and ends (tail) are spread.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spectrogram-generator_amd"))
from spectro import _capi  # noqa: E402
from spectro.windows import get_window  # noqa: E402
from bench import Telemetry  # noqa: E402

N, NFFT, HOP, NB = 480000, 1024, 256, 4
BINS = NFFT // 2 + 1
BPF = HOP * 4 + BINS * 4
DT, ISZ, CODE = np.float32, 4, _capi.F32            # --nfft / --hop / --f64 replace these (any kernel family: the data leg only)


def pct(v, q):
    return float(np.percentile(v, q))


class Rig:
    def __init__(self, clips, n_sets=NB):
        self.clips = clips
        self.plan = _capi.Plan(NFFT, NFFT, HOP, get_window("hann", NFFT), 1, 48000.0, 0, 0, CODE)
        self.nfr = self.plan.n_frames(N)
        self.frames = clips * self.nfr
        self.n_sets = n_sets
        self.ins = [_capi.DeviceBuffer(clips * N * ISZ) for _ in range(n_sets)]
        self.outs = [_capi.DeviceBuffer(clips * self.nfr * BINS * ISZ) for _ in range(n_sets)]
        self.stamped = "stamp" in os.path.basename(os.path.dirname(_capi.LIB_PATH)) and self.plan.kernel == "r8x3"
        self.n_waves = 0
        if self.stamped:
            cu = _capi.device_info()["compute_units"]
            self.n_waves = min(cu * 4 * 3, (self.frames + 3) // 4)
            self.stamps = _capi.DeviceBuffer(65536 * 64)
            _capi.check(_capi.lib().sg_memset(self.stamps.ptr, 0, 65536 * 64, None))
            os.environ["SPECTRO_R8_STAMP_PTR"] = hex(self.stamps.ptr)

    def fill(self, kind, seed=1234):
        rng = np.random.default_rng(seed)
        for b, buf in enumerate(self.ins):
            if kind == "zeros":
                _capi.check(_capi.lib().sg_memset(buf.ptr, 0, self.clips * N * ISZ, None))
            else:
                chunk = 64
                x = (rng.standard_normal((min(chunk, self.clips), N)) * 0.1).astype(DT)
                for c0 in range(0, self.clips, chunk):       # big batches: the same 64 clips repeated (values do not matter here)
                    n = min(chunk, self.clips - c0)
                    _capi.check(_capi.lib().sg_memcpy_h2d(buf.ptr + c0 * N * ISZ, x.ctypes.data, n * N * ISZ, None))
                _capi.stream_sync()
        _capi.stream_sync()

    def launch(self, i):
        b = i % self.n_sets
        self.plan.stft(self.ins[b].ptr, N, N, self.clips, self.outs[b].ptr, self.nfr * BINS)

    def sustained(self, secs):
        t0, i = time.perf_counter(), 0
        while time.perf_counter() - t0 < secs:
            for _ in range(32):
                self.launch(i)
                i += 1
            _capi.stream_sync()
        return (time.perf_counter() - t0) * 1e6 / i

    def timed(self, n=400):
        _capi.stream_sync()
        t0 = time.perf_counter()
        for i in range(n):
            self.launch(i)
        _capi.stream_sync()
        return (time.perf_counter() - t0) * 1e6 / n

    def read_stamps(self, keep=None):
        """-> dict from the stamps of the LAST launch, or None without the diagnostic build"""
        if not self.stamped:
            return None
        raw = np.zeros((self.n_waves, 16), np.uint64)
        self.stamps.download(raw)
        _capi.stream_sync()
        if keep is not None:
            keep.append(raw.copy())
        raw = raw[raw[:, 0] != 0]
        if len(raw) == 0:
            return None
        c0, r0, c1, r1, c2, r2, fr = (raw[:, k].astype(np.int64) for k in range(7))
        dc, dr = (c2 - c0).astype(np.float64), (r2 - r0).astype(np.float64)
        clk = dc / dr * 100.0                                          # MHz: shader cycles per 10 ns tick
        lc, lr = (c2 - c1).astype(np.float64), (r2 - r1).astype(np.float64)
        start = (r0 - r0.min()) * 0.01                                  # us after the first wave started
        end = (r2.max() - r2) * 0.01                                    # us before the last wave ended
        life = dr * 0.01
        span = float((r2.max() - r0.min()) * 0.01)
        xcc = (raw[:, 7] >> np.uint64(32)).astype(np.int64) & 0xf
        return {
            "waves": int(len(raw)), "frames_per_wave": [int(fr.min()), int(fr.max())],
            "clock_MHz_median": pct(clk, 50), "clock_MHz_p5_p95": [pct(clk, 5), pct(clk, 95)],
            "loop_clock_MHz_median": pct(lc / lr * 100.0, 50),
            "span_us": span, "wave_life_us_mean": float(life.mean()), "life_over_span": float(life.mean() / span),
            "start_us_p50_p90_max": [pct(start, 50), pct(start, 90), float(start.max())],
            "end_gap_us_p50_p90_max": [pct(end, 50), pct(end, 90), float(end.max())],
            "prologue_us_median": pct((r1 - r0) * 0.01, 50),
            "cycles_per_frame_median": pct(lc / fr, 50), "ns_per_frame_per_wave_median": pct(lr * 10.0 / fr, 50),
            "xcc_ids_seen": sorted(set(int(v) for v in xcc)),
            "tail": None if not raw[:, 10].any() else {
                "static_end_us_p50": pct((raw[:, 8].astype(np.int64) - r0.min()) * 0.01, 50),
                "fetches_per_wave_mean": float(raw[:, 10].mean()),
                "fetch_us_mean": float(raw[:, 9].astype(np.int64).sum() * 0.01 / max(raw[:, 10].sum(), 1)),
            },
        }


def leg_data(rig, secs, out):
    tel_id = _capi.device_pci_bus_id()
    for rep in (1, 2):
        for kind in ("random", "zeros"):
            rig.fill(kind)
            tel = Telemetry(tel_id).start()
            us_sus = rig.sustained(secs)
            power = tel.stop(skip_s=min(0.6, secs / 2))
            us = rig.timed(400)
            row = {"leg": "data", "kernel": rig.plan.kernel, "nfft": NFFT, "hop": HOP, "rep": rep, "input": kind, "us_per_launch": us, "us_per_launch_sustained": us_sus,
                   "frac_of_8TBs": rig.frames * BPF / us / 8e6, "power": power}
            if rig.stamped:
                samples = []
                for _ in range(5):
                    for i in range(20):
                        rig.launch(i)
                    _capi.stream_sync()
                    samples.append(rig.read_stamps())
                row["stamps_last_of_20_x5"] = samples
            out.append(row)
            print(json.dumps(row), flush=True)


def leg_duty(rig, out, n=300, gap_factor=2.5):
    rig.fill("random")
    us = rig.timed(100)
    gap = gap_factor * us * 1e-6
    rig.sustained(0.3)
    time.sleep(0.5)
    host, spans, clocks = [], [], []
    for i in range(n):
        t0 = time.perf_counter()
        rig.launch(i)
        _capi.stream_sync()
        t1 = time.perf_counter()
        host.append((t1 - t0) * 1e6)
        if rig.stamped and i % 10 == 9:
            st = rig.read_stamps()
            if st:
                spans.append(st["span_us"])
                clocks.append(st["clock_MHz_median"])
        t_end = time.perf_counter() + gap
        while time.perf_counter() < t_end:
            pass
    row = {"leg": "duty", "launches": n, "idle_gap_us": gap * 1e6, "back_to_back_us_per_launch": us,
           "host_launch_plus_sync_us_p10_p50_p90": [pct(host, 10), pct(host, 50), pct(host, 90)],
           "note": "host figure includes launch + sync overhead; the tracer's per-dispatch durations are the kernel time"}
    if spans:
        row["stamp_span_us_p10_p50_p90"] = [pct(spans, 10), pct(spans, 50), pct(spans, 90)]
        row["stamp_clock_MHz_p10_p50_p90"] = [pct(clocks, 10), pct(clocks, 50), pct(clocks, 90)]
    out.append(row)
    print(json.dumps(row), flush=True)


def leg_big(clips, secs, out):
    rig = Rig(clips, n_sets=2)
    rig.fill("random")
    tel = Telemetry(_capi.device_pci_bus_id()).start()
    us_sus = rig.sustained(secs)
    power = tel.stop(skip_s=min(0.6, secs / 2))
    us = rig.timed(60)
    row = {"leg": "big", "clips": clips, "frames": rig.frames, "GB_per_launch": rig.frames * BPF / 1e9, "us_per_launch": us,
           "us_per_launch_sustained": us_sus, "frames_per_s": rig.frames / us * 1e6, "frac_of_8TBs": rig.frames * BPF / us / 8e6,
           "power": power}
    if rig.stamped:
        row["stamps"] = rig.read_stamps()
    out.append(row)
    print(json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--big-clips", type=int, default=768)
    ap.add_argument("--secs", type=float, default=2.0)
    ap.add_argument("--legs", default="data,duty,big")
    ap.add_argument("--out", default=None)
    ap.add_argument("--nfft", type=int, default=1024)
    ap.add_argument("--hop", type=int, default=256)
    ap.add_argument("--f64", action="store_true")
    ap.add_argument("--dump-stamps", default=None, help="diagnostic build: .npy of the raw stamps [launch][wave][8] of 6 launches (3 back to back, 3 after idle)")
    a = ap.parse_args()
    global NFFT, HOP, BINS, BPF, DT, ISZ, CODE
    NFFT, HOP = a.nfft, a.hop
    BINS = NFFT // 2 + 1
    if a.f64:
        DT, ISZ, CODE = np.float64, 8, _capi.F64
    BPF = HOP * ISZ + BINS * ISZ
    _capi.ensure_device()
    out = [{"lib": _capi.LIB_PATH, "device": _capi.device_info()}]
    print(json.dumps(out[0]), flush=True)
    legs = a.legs.split(",")
    if "data" in legs or "duty" in legs or a.dump_stamps:
        rig = Rig(a.clips)
        if "data" in legs:
            leg_data(rig, a.secs, out)
        if "duty" in legs:
            leg_duty(rig, out)
        if a.dump_stamps and rig.stamped:
            keep = []
            rig.fill("random")
            rig.sustained(1.0)
            for _ in range(3):
                for i in range(8):
                    rig.launch(i)
                _capi.stream_sync()
                rig.read_stamps(keep)
            for i in range(3):
                time.sleep(0.01)
                rig.launch(i)
                _capi.stream_sync()
                rig.read_stamps(keep)
            np.save(a.dump_stamps, np.stack(keep))
        del rig
        _capi.device_pool_clear()
    if "big" in legs:
        leg_big(a.big_clips, a.secs, out)
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
