#!/bin/bash
# On the GPU box: the four results profiles/r03_limiter.txt is made of.  tools/run_limiter.sh <tag>
tag=${1:-r3}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -e
timeout -k 10 240 python3 $R/tools/limiter.py --out $O/limiter_plain.json > $O/limiter_plain.log 2>&1
SPECTRO_LIB=$R/spectrogram-generator_amd/lib_stamp/libspectro.so timeout -k 10 240 python3 $R/tools/limiter.py --out $O/limiter_stamp.json > $O/limiter_stamp.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $O/duty_trace -- python3 $R/tools/limiter.py --legs duty > $O/duty_trace.log 2>&1
python3 - <<PY
import csv, glob
import numpy as np
rows = []
for f in glob.glob("$O/duty_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stft1024_r8x3" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
d = np.array([(e - s) / 1e3 for s, e in rows])
gap = np.array([(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)])
duty = gap > 150.0           # the dispatch that FOLLOWS an idle gap
b2b = ~duty
dd, db = d[1:][duty], d[1:][b2b]
with open("$O/duty_trace_summary.txt", "w") as fh:
    for name, v in (("after an idle gap >= 150 us", dd), ("back to back (gap < 150 us)", db)):
        if len(v):
            print(f"{name}: n={len(v)} p10={np.percentile(v,10):.1f} p50={np.percentile(v,50):.1f} p90={np.percentile(v,90):.1f} mean={v.mean():.1f} us", file=fh)
    if duty.any():
        print(f"idle gaps: p50={np.percentile(gap[duty],50):.0f} us", file=fh)
print(open("$O/duty_trace_summary.txt").read())
PY
rm -rf $O/duty_trace
timeout -k 10 300 python3 $R/bench.py --clips 768 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-mode > $O/bench_clips768.json 2> $O/bench_clips768.err
tail -c 1500 $O/bench_clips768.json
