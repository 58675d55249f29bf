#!/usr/bin/env python3
"""Where do the ~110 us go that a 20-launch timed region of bench.py spends beyond its kernels?  python tools/bench_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spectrogram-generator_amd"))
import torch
from spectro import _capi
from spectro.windows import get_window
dev = torch.device("cuda", 0); torch.cuda.set_device(0); _capi.ensure_device(0)
plan = _capi.Plan(1024, 1024, 256, get_window("hann", 1024), 1, 48000.0, 0, 0, _capi.F32)
N, C = 480000, 64
nfr = plan.n_frames(N)
xs = [torch.randn((C, N), device=dev) * 0.1 for _ in range(4)]
outs = [torch.empty((C, nfr, 513), device=dev) for _ in range(4)]
stream = torch.cuda.current_stream(dev).cuda_stream
def step(i):
    b = i % 4
    plan.stft(xs[b].data_ptr(), N, N, C, outs[b].data_ptr(), nfr * 513, stream=stream)
for i in range(300): step(i)
torch.cuda.synchronize(dev)
rows = []
for K in (1, 5, 20, 20, 20, 100):
    for i in range(5): step(i)
    torch.cuda.synchronize(dev); torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); ev0.record(); t1 = time.perf_counter()
    for i in range(K): step(i)
    t2 = time.perf_counter(); ev1.record(); t3 = time.perf_counter()
    while not ev1.query(): pass
    t4 = time.perf_counter(); torch.cuda.synchronize(dev); t5 = time.perf_counter()
    ev_us = ev0.elapsed_time(ev1) * 1e3
    rows.append((K, ev_us, (t5 - t0) * 1e6, (t4 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6 / K, (t3 - t2) * 1e6, (t5 - t4) * 1e6))
print("K  events_us  host_total_us  host_to_poll_us  ev0.record_us  per_step_submit_us  ev1.record_us  final_sync_us  | fixed = host_total - events")
for r in rows:
    print("%3d %9.1f %12.1f %14.1f %12.1f %16.2f %12.1f %12.1f | %.1f (to poll: %.1f)" % (*r, r[2] - r[1], r[3] - r[1]))
