#!/usr/bin/env python3
"""Beside tools/save_profile.py: the dispatch-duration summary of the kernel trace and the SQ counters per frame of a tools/profile_round.sh run.
    tools/save_profile_extras.py <tag>      (reads gpurun_out/prof_<tag>/, writes profiles/<tag>_kernel_trace_durations.txt, _sq_counters.txt, _bench_driver_shape.json)"""
import csv, glob, shutil, statistics as st, sys
tag = sys.argv[1]
O = f"gpurun_out/prof_{tag}"
f = glob.glob(O + "/kt/**/*kernel_trace.csv*", recursive=True)[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "stft1024_r8x3" in r["Kernel_Name"]]
q = lambda v, p: sorted(v)[int(p * (len(v) - 1))]
line = lambda name, v: f"{name}: mean {st.mean(v):.2f} us, p10 {q(v, .1):.2f}, p50 {q(v, .5):.2f}, p90 {q(v, .9):.2f}"
open(f"profiles/{tag}_kernel_trace_durations.txt", "w").write(
    f"stft1024_r8x3_kernel dispatches in the kernel trace of `python3 bench.py` (tools/profile_round.sh {tag}): n={len(d)}\n" + line("all", d) + "\n" +
    line("last 500 (the timed region)", d[-500:]) + "\n")
f = glob.glob(O + "/sq/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "stft1024_r8x3" in r["Kernel_Name"]]
out = f"rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CU_CYCLES -- python3 bench.py --steps 40 --warmup 10 ... (tools/profile_round.sh {tag}); stft1024_r8x3_kernel, per dispatch\n"
for name in sorted(set(r["Counter_Name"] for r in rows)):
    v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name]
    out += f"{name:24s} n={len(v):3d} avg={st.mean(v):14.1f}   per frame {st.mean(v) / 119808:.2f}\n"
open(f"profiles/{tag}_sq_counters.txt", "w").write(out)
shutil.copy(O + "/bench_driver_shape.json", f"profiles/{tag}_bench_driver_shape.json")
print(open(f"profiles/{tag}_kernel_trace_durations.txt").read() + out)
