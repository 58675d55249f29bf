#!/bin/bash
# On the GPU box: SQ counters of the round-4 kernels (three rocprofv3 --pmc passes per shape, program after `--`, no trace domains).
#   tools/new_kernel_counters.sh <out.txt>
R=${GRAFT_REPO_ROOT:-$PWD}
out=${1:-$R/gpurun_out/r4/new_kernel_counters.txt}
case $out in /*) ;; *) out=$R/$out;; esac
mkdir -p $(dirname $out)
: > $out
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
cd /tmp && export TMPDIR=/tmp
for spec in 64:16:f32:64 128:32:f32:64 1000:250:f32:64 4000:1000:f32:64 8160:2040:f32:64 8192:2048:f32:64 4000:1000:f64:32 8160:2040:f64:32 8192:2048:f64:32; do
  echo "== $spec" >> $out
  tag=$(echo $spec | tr ':' '_')
  i=0
  for ctrs in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    QA_SECS=0.03 timeout -k 5 120 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_new_${tag}_$i -- python3 $R/tools/quick_any.py $spec > $R/gpurun_out/pmc_new_${tag}_$i.log 2>&1
  done
  grep -h "^f" $R/gpurun_out/pmc_new_${tag}_1.log >> $out
  python3 - $R/gpurun_out/pmc_new_${tag} >> $out <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stft' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(f"{k:28s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
PY
  rm -rf $R/gpurun_out/pmc_new_${tag}_*
done
cat $out
