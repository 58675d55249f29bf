mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3/pytest_full.log 2>&1; echo "pytest rc=$?" 
tail -3 gpurun_out/r3/pytest_full.log
SPECTRO_LIB=$PWD/spectrogram-generator_amd/lib_stamp/libspectro.so timeout -k 10 100 python3 tools/limiter.py --legs data --secs 0.5 --out gpurun_out/r3/prologue_stamp.json > gpurun_out/r3/prologue_stamp.log 2>&1; python3 - <<PY
import json
d=json.load(open("gpurun_out/r3/prologue_stamp.json"))
for r in d[1:3]:
    s=r["stamps_last_of_20_x5"]
    print(r["input"], round(r["us_per_launch"],1), "span", [round(x["span_us"],1) for x in s], "life/span", [round(x["life_over_span"],3) for x in s], "prologue", [round(x["prologue_us_median"],2) for x in s], "ns/frame", round(s[0]["ns_per_frame_per_wave_median"]))
PY
tools/ab.sh base ""
