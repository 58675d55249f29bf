#!/bin/bash
# Build a variant of libspectro.so into spectrogram-generator_amd/lib_<name>/ for tools/ab.sh:
#   tools/build_variant.sh <name> <file.hip> "<extra hipcc flags>"      (other objects are taken from lib/; a copy named <file>_tmp.hip stands in for <file>.hip)
set -e
name=$1; src=$2; defs=$3
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/spectrogram-generator_amd
mkdir -p $P/lib_$name
base=$(basename $src .hip); base=${base%_tmp}
noslp=""
case $base in stft_r8x3|stft_rsmall|stft_rbig|stft_rblue|stft_mel_fused) noslp="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-gpu-rdc -Wno-unused-function -Wno-unused-result $noslp $defs \
  -I $R/include -I $P/csrc -c $src -o $P/lib_$name/$base.o
objs=""
for o in $P/lib/*.o; do
  if [ "$(basename $o)" == "$base.o" ]; then objs="$objs $P/lib_$name/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $P/lib_$name/libspectro.so
echo "built lib_$name"
