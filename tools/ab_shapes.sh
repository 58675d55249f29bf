#!/bin/bash
# Interleaved A/B of libspectro builds on arbitrary (nfft, hop) shapes with tools/quick_bench.py (on the GPU box):
#   tools/ab_shapes.sh <out.txt> "<nfft:hop nfft:hop ...>" name1 name2 ...      ("base" = lib/, others = lib_<name>/); two rounds, best of each
R=${GRAFT_REPO_ROOT:-$PWD}
out=$1; shapes=$2; shift 2
mkdir -p $(dirname $out); : > $out
for rep in 1 2; do for sh in $shapes; do for v in "$@"; do
  n=${sh%:*}; hop=${sh#*:}
  if [ "$v" == "base" ]; then lib=$R/spectrogram-generator_amd/lib/libspectro.so; else lib=$R/spectrogram-generator_amd/lib_$v/libspectro.so; fi
  us=$(SPECTRO_LIB=$lib QB_SECS=${QB_SECS:-0.4} python3 $R/tools/quick_bench.py 64 $hop - $n 2>/dev/null | grep "^kernel" | awk '{print $5}' | sort -n | head -1)
  echo "n$n hop $hop $v $us" >> $out
done; done; done
python3 - $out <<'PY'
import sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ln in open(sys.argv[1]):
    a = ln.split()
    if len(a) == 5: acc[(a[0], a[2])][a[3]].append(float(a[4]))
for shape, d in acc.items():
    print(f"{shape[0]} hop {shape[1]:>4s}: " + "   ".join(f"{k} {min(v):8.1f} us" for k, v in d.items()))
PY
