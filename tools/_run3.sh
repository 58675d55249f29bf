mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_stft.py -x -q -m gpu -k "rbig or f64 or random_shapes" > gpurun_out/r3/pytest_rbigd.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r3/pytest_rbigd.log
for cfg in "64 2048 256" "64 2048 1792" "64 4096 256" "64 4096 3584" "64 2048 64"; do echo "== $cfg"; QB_SECS=0.5 timeout -k 10 200 python3 tools/quick_f64.py $cfg; done 2>&1 | tee gpurun_out/r3/quick_f64_rbigd.txt
