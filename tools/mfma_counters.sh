#!/bin/bash
# MFMA counters of the mel kernels (the MFMA tile kernels are opt-in since round 2; the default fused kernel has none)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA" | head -20 > $R/gpurun_out/r2/mfma_counter_list.txt
for ctr in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  SPECTRO_FUSED_MFMA=1 timeout -k 10 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/r2/mfma_$tag -- python3 $R/tools/quick_fused.py 0.3 > $R/gpurun_out/r2/mfma_$tag.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/r2/mfma_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if 'mel' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:60],r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, "n=%d avg=%.1f" % (len(v), sum(v)/len(v)))
PY
