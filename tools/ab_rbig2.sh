#!/bin/bash
# Interleaved sustained A/B of several libspectro builds on the rbig shapes (on the GPU box): tools/ab_rbig2.sh <out.txt> name1 name2 ...
# ("base" = the product lib/; others = spectrogram-generator_amd/lib_<name>/ from tools/build_variant.sh); two rounds.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$1; shift
mkdir -p $(dirname $out)
: > $out
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" == "base" ]; then lib=$R/spectrogram-generator_amd/lib/libspectro.so; else lib=$R/spectrogram-generator_amd/lib_$v/libspectro.so; fi
    echo "== $v (round $rep)" >> $out
    SPECTRO_LIB=$lib QB_SECS=${QB_SECS:-0.4} python3 $R/tools/quick_rbig.py 2>/dev/null | grep "^n" >> $out || exit 1
  done
done
python3 - $out <<'PY'
import re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
name = None
for ln in open(sys.argv[1]):
    m = re.match(r"== (\S+)", ln)
    if m: name = m.group(1); continue
    m = re.match(r"n(\d+) hop (\d+): spectrum ([\d.]+) ms .* band ([\d.]+) ms", ln)
    if m: acc[(int(m.group(1)), int(m.group(2)))][name].append((float(m.group(3)), float(m.group(4))))
print("shape            " + "  ".join(f"{n:>22s}" for n in next(iter(acc.values()))))
for shape, d in sorted(acc.items()):
    print(f"n{shape[0]} hop {shape[1]:3d}:  " + "  ".join(f"{min(x[0] for x in v)*1e3:7.1f} / {min(x[1] for x in v)*1e3:7.1f} us    " for v in d.values()))
PY
