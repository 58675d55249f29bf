#!/bin/bash
# compact same-box A/B of run-time knobs: tools/ab_env2.sh "VAR=val" ... (2 rounds, prints us/launch)
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for v in "$@"; do
    echo -n "[$v] "
    env $v python3 $R/bench.py --steps ${STEPS:-3000} --warmup 100 --no-cpu-baseline --telemetry-s 0 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['roofline']['us_per_launch'],2), round(d['roofline']['frac'],4))"
  done
done
