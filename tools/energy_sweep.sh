#!/bin/bash
# Energy price list on the GPU box: board power of each tools/ubench/energy_rate.hip variant and of the steady HBM streams.
R=${GRAFT_REPO_ROOT:-$PWD}
SECS=${1:-4}
for v in 7 0 1 2 3 4 5 6 8 9 10 11; do
  python3 $R/tools/telemetry.py --skip 1.0 -- $R/tools/ubench/energy_rate.bin $v $SECS 2>&1 | grep -E "variant|power1|freq1"
done
for m in rows read write mix; do
  python3 $R/tools/telemetry.py --skip 1.0 -- $R/tools/ubench/hbm_peaks.bin sustain $m $SECS 2>&1 | grep -E "sustain|power1|freq1"
done
