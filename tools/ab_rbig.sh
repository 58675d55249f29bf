#!/bin/bash
# A/B several builds of libspectro on the rbig shapes (on the GPU box):  tools/ab_rbig.sh name1 name2 ...
# (libs prebuilt in spectrogram-generator_amd/lib_<name>/ with SG_RBIG_DEFS=... python build.py --force)
R=${GRAFT_REPO_ROOT:-$PWD}
for shape in "64 4096" "256 4096" "64 2048"; do
  set -- "$@"
  for v in "$@"; do
    hop=${shape% *}; n=${shape#* }
    echo -n "n=$n hop=$hop $v: "
    SPECTRO_LIB=$R/spectrogram-generator_amd/lib_$v/libspectro.so python $R/tools/quick_bench.py 64 $hop - $n 2>/dev/null | grep "^kernel" | tail -1 | awk '{print $5, $6}'
  done
done
