#!/usr/bin/env python3
"""Shader clock and board power while a command runs (GPU box):

    python tools/telemetry.py [--period 0.05] -- python bench.py --steps 20000 --no-cpu-baseline

Polls the amdgpu hwmon nodes (freq1_input = sclk, freq2_input = mclk, power1_average / power1_input, power1_cap) of
every card from a thread while the child runs, prints min / median / max over the samples taken after the first
``--skip`` seconds, then the child's stdout.  Read-only sysfs, no privileges needed; falls back to one ``rocm-smi``
snapshot mid-run if the nodes are not readable.
"""
from __future__ import annotations

import argparse
import glob
import statistics
import subprocess
import sys
import threading
import time


def nodes():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "power1_cap", "temp1_input"):
            p = f"{hw}/{name}"
            try:
                with open(p) as fh:
                    fh.read()
                out.setdefault(hw, {})[name] = p
            except OSError:
                pass
    return out


def read(p):
    try:
        with open(p) as fh:
            return float(fh.read().strip())
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--period", type=float, default=0.05)
    ap.add_argument("--skip", type=float, default=0.5, help="seconds after the first busy sample to drop")
    ap.add_argument("--all", action="store_true", help="print every card, not only the busiest")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    nd = nodes()
    samples = []
    stop = threading.Event()

    def poll():
        while not stop.is_set():
            t = time.perf_counter()
            for hw, d in nd.items():
                samples.append((t, hw, {k: read(p) for k, p in d.items()}))
            time.sleep(a.period)

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t0 = time.perf_counter()
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    smi = None
    if not nd:
        time.sleep(3.0)
        smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    out, err = child.communicate()
    stop.set()
    th.join()
    print(f"[telemetry] child rc={child.returncode} wall={time.perf_counter() - t0:.2f}s, hwmon nodes: {len(nd)}")
    def peak(hw):
        return max([(s.get("power1_average") or s.get("power1_input") or 0.0) for (t, h, s) in samples if h == hw] or [0.0])
    cards = list(nd) if a.all else sorted(nd, key=peak)[-1:]
    for hw in cards:
        rows = [(t, s) for (t, h, s) in samples if h == hw]
        cap = rows[0][1].get("power1_cap") if rows else None
        pkey = "power1_average" if rows and rows[0][1].get("power1_average") is not None else "power1_input"
        # busy = power above 40 % of the maximum seen
        pw = [s.get(pkey) or 0.0 for _, s in rows]
        if not pw:
            continue
        thr = 0.4 * max(pw)
        busy = [(t, s) for (t, s) in rows if (s.get(pkey) or 0.0) >= thr]
        if busy:
            tb = busy[0][0] + a.skip
            busy = [(t, s) for (t, s) in busy if t >= tb] or busy
        for key, unit, div in (("freq1_input", "MHz sclk", 1e6), ("freq2_input", "MHz mclk", 1e6), (pkey, "W", 1e6), ("temp1_input", "C", 1e3)):
            v = [s[key] / div for _, s in busy if s.get(key) is not None]
            if v:
                print(f"[telemetry] {hw.split('/')[4]} {key:15s} busy n={len(v):4d} min {min(v):8.1f} med {statistics.median(v):8.1f} max {max(v):8.1f} {unit}")
        if cap:
            print(f"[telemetry] {hw.split('/')[4]} power cap {cap / 1e6:.0f} W")
    if smi:
        print(smi)
    sys.stdout.write(out)
    sys.stderr.write(err[-2000:])
    return child.returncode


if __name__ == "__main__":
    sys.exit(main())
