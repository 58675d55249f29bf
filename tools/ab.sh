#!/bin/bash
# A/B several builds of libspectro in ONE process sequence on the GPU box (interleaved, 2 rounds):
#   tools/ab.sh name1 name2 ...     (libs prebuilt in spectrogram-generator_amd/lib_<name>/)
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for v in "$@"; do
    echo -n "$v: "
    SPECTRO_LIB=$R/spectrogram-generator_amd/lib${v:+_$v}/libspectro.so python $R/bench.py --steps 400 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['roofline']['us_per_launch'],1), round(d['roofline']['frac'],3))"
  done
done
