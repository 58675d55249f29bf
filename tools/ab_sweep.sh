#!/bin/bash
# A/B several builds of libspectro on cfg4 shapes (on the GPU box):  tools/ab_sweep.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-$PWD}
for shape in "64 256" "128 256" "64 512" "256 512" "64 2048" "256 2048" "64 4096" "256 4096"; do
  for v in "$@"; do
    hop=${shape% *}; n=${shape#* }
    echo -n "n=$n hop=$hop $v: "
    SPECTRO_LIB=$R/spectrogram-generator_amd/lib_$v/libspectro.so python $R/tools/quick_bench.py 64 $hop - $n 2>/dev/null | grep "^kernel" | tail -1 | awk '{print $5, $6}'
  done
done
