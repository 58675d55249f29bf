#!/bin/bash
# Same-box A/B of the dynamic tail (SPECTRO_TAIL_PCT / SPECTRO_TAIL_CHUNK) on the headline bench, then the in-kernel stamps of each setting.
#   tools/ab_tail.sh "PCT CHUNK" "PCT CHUNK" ...
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for v in "$@"; do
    set -- $v
    echo -n "pct=$1 chunk=$2: "
    SPECTRO_TAIL_PCT=$1 SPECTRO_TAIL_CHUNK=$2 python3 $R/bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-reference-mode --telemetry-s 0 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(round(d['roofline']['us_per_launch'],2), 'us', round(d['roofline']['frac'],4))"
    set --
  done
done
