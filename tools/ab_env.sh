#!/bin/bash
# Same-box A/B of run-time knobs of the headline kernel, each with board power / sclk beside it:
#   tools/ab_env.sh "VAR=val VAR2=val" "VAR=val" ...     ("" = defaults)
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for v in "$@"; do
    echo "== [$v]"
    env $v python3 $R/tools/telemetry.py --skip 0.7 -- python3 $R/bench.py --steps ${STEPS:-20000} --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('[telemetry]') and ('freq1' in l or 'power1_input' in l): print('   ', ' '.join(l.split()[2:]))
    if l.startswith('{'):
        d=json.loads(l); print('    us/launch', round(d['roofline']['us_per_launch'],2), 'frac', round(d['roofline']['frac'],4))
"
  done
done
