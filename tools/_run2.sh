mkdir -p gpurun_out/r3
: > gpurun_out/r3/repro_malloc_async.txt
for v in 0 1 2 3 4; do timeout -k 10 200 tools/repro_malloc_async.bin 1500 0 $v >> gpurun_out/r3/repro_malloc_async.txt 2>&1; echo "variant $v rc=$?"; done
grep variant gpurun_out/r3/repro_malloc_async.txt
