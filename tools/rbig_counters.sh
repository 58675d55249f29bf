#!/bin/bash
# On the GPU box: SQ counters of stft_rbig_kernel for nfft 4096 / 2048 at hops 256 / 64 (three rocprofv3 --pmc passes per shape,
# program after `--`, no trace domains), then the sustained timings of tools/quick_rbig.py.   tools/rbig_counters.sh <tag>
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r4
out=$R/gpurun_out/r4/rbig_counters_$tag.txt
: > $out
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
for shape in "256 4096" "64 4096" "256 2048" "64 2048"; do
  hop=${shape% *}; n=${shape#* }
  echo "== nfft $n hop $hop" >> $out
  QB_ARGS="64 $hop - $n" bash $R/tools/pmc.sh rbig_${tag}_${n}_${hop} "$P1" "$P2" "$P3" >> $out 2>&1 || exit 1
  grep -h "^kernel" $R/gpurun_out/pmc_rbig_${tag}_${n}_${hop}_1.log | tail -1 >> $out
done
echo "== sustained timings (tools/quick_rbig.py, un-profiled)" >> $out
cd $R && python3 tools/quick_rbig.py >> $out 2>&1
