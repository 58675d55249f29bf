#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <tag> "<counters pass1>" ["<counters pass2>" ...]
# Runs tools/quick_bench.py under rocprofv3 --pmc (one run per pass) and prints the per-dispatch averages for the stft kernel.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/tools/quick_bench.py ${QB_ARGS:-64 256} > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_${tag}_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stft' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print(f"{k:28s} n={len(v):3d} avg={sum(v)/len(v):14.1f}")
PY
