#!/bin/bash
# A/B two builds of libspectro.so with any measuring tool, interleaved, two rounds, on the GPU box:
#   tools/ab_lib.sh <variant> <out.txt> <command ...>      (variant built by tools/build_variant.sh; "" = lib/)
R=${GRAFT_REPO_ROOT:-$PWD}
v=$1; out=$2; shift 2
mkdir -p $(dirname $out)
{
for rep in 1 2; do for name in "" "$v"; do
  echo "== build: ${name:-product}"
  SPECTRO_LIB=$R/spectrogram-generator_amd/lib${name:+_$name}/libspectro.so "$@"
done; done
} > $out 2>&1
