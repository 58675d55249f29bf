import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, os.path.join(ROOT, "spectrogram-generator_amd"))
from spectro.stream import StreamingSTFT
rng = np.random.default_rng(1)
chunk = (rng.standard_normal((8, 4096)) * 0.1).astype(np.float32)
for rep in range(2):
    for tr in ("device", "host"):
        st = StreamingSTFT(8, 96000.0, 4096, 1024, window="hann", transport=tr)
        for _ in range(20): st.feed(chunk)
        lat = []
        t0 = time.perf_counter()
        for _ in range(400):
            c0 = time.perf_counter(); t, s = st.feed(chunk); lat.append(time.perf_counter() - c0)
        dt = time.perf_counter() - t0
        print(f"cfg5 transport={tr}: median {np.median(lat)*1e6:.1f} us  p99 {np.percentile(lat,99)*1e6:.1f} us per 4096-sample chunk x 8 ch; {400*4096/96000.0/dt:.0f}x real time", flush=True)
        st.close()
