#!/usr/bin/env python3
"""Copy a rocprofv3 run from gpurun_out/ into profiles/ (trimmed):  tools/save_profile.py <tag> <kt_dir> <fetch_dir> <write_dir> [bench.json]"""
import csv, glob, json, os, shutil, sys
tag, kt, fe, wr = sys.argv[1:5]
os.makedirs('profiles', exist_ok=True)
rows = list(csv.reader(open(glob.glob(f'{kt}/**/*kernel_stats.csv', recursive=True)[0])))
with open(f'profiles/{tag}_kernel_stats.csv', 'w', newline='') as fh:
    w = csv.writer(fh)
    for r in rows:
        r[0] = r[0][:110]
        w.writerow(r)
def avg(path, name):
    rows = list(csv.DictReader(open(glob.glob(f'{path}/**/*counter_collection.csv', recursive=True)[0])))
    v = [float(r['Counter_Value']) for r in rows if 'stft1024_r8x3' in r['Kernel_Name'] and r['Counter_Name'] == name]
    return sum(v) / len(v), len(v)
f, nf = avg(fe, 'FETCH_SIZE')
w_, nw = avg(wr, 'WRITE_SIZE')
stft = [r for r in rows if 'stft1024' in r[0]][0]
d = {"tag": tag, "kernel": stft[0], "calls": int(stft[1]), "avg_ns": float(stft[3]), "min_ns": float(stft[5]), "max_ns": float(stft[6]),
     "workload_frames": 119808, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w_, "dispatches": [nf, nw],
     "correction": "gfx950: FETCH_SIZE reports 1/2 of coalesced streaming reads (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact",
     "fetch_bytes_per_launch": 2 * f * 1024, "write_bytes_per_launch": w_ * 1024,
     "hbm_bytes_per_launch": 2 * f * 1024 + w_ * 1024, "algorithmic_bytes_per_launch": 119808 * 3076 + 64 * 768 * 4}
json.dump(d, open(f'profiles/{tag}_traffic.json', 'w'), indent=1)
json.dump({"source": f"profiles/{tag}_traffic.json", "workload_frames": 119808, "hbm_bytes_per_launch": d["hbm_bytes_per_launch"]},
          open('profiles/traffic_latest.json', 'w'))
if len(sys.argv) > 5:
    shutil.copy(sys.argv[5], f'profiles/{tag}_bench.json')
print(json.dumps(d, indent=1))
