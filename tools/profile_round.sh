#!/bin/bash
# On the GPU box: the round's evidence for the headline kernel -- bench line, kernel trace + stats of the SAME command,
# HBM traffic counters in separate passes (no trace domains with --pmc).   tools/profile_round.sh <tag>
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2>> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-reference-mode --no-limiter-leg --no-secondary --telemetry-s 0 > $O/kt.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 40 --warmup 10 --settle-ms 0 --no-cpu-baseline --no-reference-mode --no-limiter-leg --no-secondary --telemetry-s 0 > $O/fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 40 --warmup 10 --settle-ms 0 --no-cpu-baseline --no-reference-mode --no-limiter-leg --no-secondary --telemetry-s 0 > $O/write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CU_CYCLES --output-format csv -d $O/sq -- python3 $R/bench.py --steps 40 --warmup 10 --settle-ms 0 --no-cpu-baseline --no-reference-mode --no-limiter-leg --no-secondary --telemetry-s 0 > $O/sq.log 2>&1
# keep only the summaries (the raw traces are tens of MB)
find $O -name "*kernel_trace.csv" -size +4M -exec sh -c 'head -c 400000 "$1" > "$1.head"; rm "$1"' _ {} \;
ls -la $O $O/kt/* 2>/dev/null | head -40
cat $O/bench.json
