R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out/r3
{
SPECTRO_LIB=$R/spectrogram-generator_amd/lib_f64twl/libspectro.so timeout -k 10 400 python -m pytest tests/test_gpu_stft.py -q -x -k "f64 or float64 or double" 2>&1 | tail -3
for rep in 1 2; do for v in "" f64twl; do for sh in "1024 256" "1024 64" "512 128" "256 64"; do
echo -n "[$v] $sh: "; QF_LEGS=0 QB_SECS=1.5 SPECTRO_LIB=$R/spectrogram-generator_amd/lib${v:+_$v}/libspectro.so python tools/quick_f64.py 64 $sh
done; done; done
} > gpurun_out/r3/f64twl.txt 2>&1
