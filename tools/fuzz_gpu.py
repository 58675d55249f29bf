#!/usr/bin/env python3
"""One-off fuzz of the device path against the oracle (on the GPU box): batches whose persistent-kernel runs cross many
clip boundaries, every register-kernel family, modes / scalings / padding / int16.   python tools/fuzz_gpu.py [cases] [seed]"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spectrogram-generator_amd"))
import spectro
from oracle import stft_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
fam = {}
for it in range(cases):
    nper = int(rng.choice([32, 64, 128, 128, 256, 256, 512, 512, 1024, 1024, 2048, 2048, 4096, 4096, 1000, 384, 6000, 96, 1504, 2016, 34, 960, 3008, 4100, 2082, 1500, 8192, 160, 192, 224]))      # even non-powers of two: the register chirp-z kernels (round 4: one wave per frame up to 2048 / 1024 in f64, two to eight above; 2082, 4100: not a multiple of 4 / 8, LDS kernel)
    r = rng.random()
    hop = int(rng.choice([64, 128, 256, 512])) if r < 0.5 else (nper - nper // 8 if r < 0.7 else int(rng.integers(1, nper + 1)))
    hop = max(1, min(hop, nper))
    n_clips = int(rng.choice([1, 2, 3, 7, 16, 33, 70]))
    n_frames = int(rng.choice([1, 2, 3, 5, 17, 64, 131, 400]))
    if nper >= 3000:
        n_clips, n_frames = min(n_clips, 3), min(n_frames, 17)
    N = nper + hop * (n_frames - 1) + int(rng.integers(0, hop))
    mode = str(rng.choice(["psd", "psd", "psd", "magnitude", "complex", "angle"]))
    scaling = str(rng.choice(["density", "spectrum"]))
    detrend = [False, "constant", "constant", "linear"][int(rng.integers(0, 4))]
    nfft = nper if rng.random() < 0.8 else int(2 ** np.ceil(np.log2(nper)) * rng.choice([1, 2]))
    nfft = max(nfft, nper)
    dt = rng.choice(["f32", "f32", "f32", "f64", "i16"])
    x = rng.standard_normal((n_clips, N)) * rng.uniform(0.01, 3.0) + rng.uniform(-1, 1)
    x = (x * 3000).astype(np.int16) if dt == "i16" else x.astype(np.float32 if dt == "f32" else np.float64)
    kw = dict(fs=float(rng.choice([500.0, 16000.0, 48000.0])), nperseg=nper, noverlap=nper - hop, nfft=nfft,
              window=[("tukey", 0.25), "hann", "boxcar"][int(rng.integers(0, 3))], detrend=detrend, scaling=scaling, mode=mode)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            f, t, s = spectro.spectrogram(x, **kw)
            fo, to, so = orc.spectrogram(x, **kw)
        ok = np.array_equal(f, fo) and np.array_equal(t, to) and s.shape == so.shape and s.dtype == so.dtype
        if ok and s.size:
            if mode == "angle":
                # phase of weak bins is ill-conditioned: compare where the magnitude is not tiny
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    mag = np.abs(orc.spectrogram(x, **{**kw, "mode": "complex"})[2])
                strong = mag > 1e-3 * mag.max(axis=-2, keepdims=True)
                d = np.abs(np.angle(np.exp(1j * (s - so))))
                ok = bool(np.all(d[strong] < (2e-2 if so.dtype == np.float32 else 1e-6)))
            else:
                ref = np.abs(so).max(axis=-2, keepdims=True)
                tol = 2e-4 if np.abs(so).dtype == np.float32 or so.dtype == np.complex64 else 1e-9
                ok = bool(np.all(np.abs(s - so) <= tol * ref + 1e-30))
    except Exception as e:          # noqa: BLE001
        ok = False
        print("EXC", repr(e))
    if not ok:
        bad += 1
        print("FAIL", it, dict(n_clips=n_clips, N=N, dt=str(dt), **kw))
print(f"{cases - bad}/{cases} cases agree with the oracle")

# ---- fused products of the nfft-1024 register kernel: band features, log display, mel ----
from spectro import engine, _capi
from spectro.mel import MelBank
from oracle import mel_oracle
pbad, pcases = 0, max(cases // 5, 20)
for it in range(pcases):
    hop = int(rng.choice([64, 128, 256, 512, 896, 100, 255, 1024]))
    n_clips = int(rng.choice([1, 2, 5, 16, 37]))
    n_frames = int(rng.choice([1, 2, 7, 33, 150]))
    N = 1024 + hop * (n_frames - 1) + int(rng.integers(0, hop))
    N += N % 2
    fs = float(rng.choice([16000.0, 48000.0]))
    detrend = ["constant", False][int(rng.integers(0, 2))]
    x = (rng.standard_normal((n_clips, N)) * rng.uniform(0.05, 2.0) + rng.uniform(-0.5, 0.5)).astype(np.float32)
    fmin, fmax = sorted(rng.uniform(0, fs / 2, 2))
    if fmax - fmin < 3 * fs / 1024:
        fmin, fmax = 0.0, fs / 2
    ok = True
    try:
        fo, to, so = orc.spectrogram(x, fs=fs, nperseg=1024, window="hann", noverlap=1024 - hop, detrend=detrend)
        m = (fo >= fmin) & (fo <= fmax)
        dc = engine.DeviceClips(x)
        t, feats = dc.band_log_power(fs, 1024, hop, fmin, fmax, window="hann", detrend=detrend)
        lp = np.log10(so[:, m, :].sum(axis=1) + 1e-20)
        ok &= np.array_equal(t, to) and bool(np.allclose(feats[..., 0], lp, atol=3e-5))
        gmax = float(so.max()) * 1e8
        fb, tt, img = dc.log_image(fs, 1024, hop, fmin, fmax, gmax, window="hann", detrend=detrend)
        db = 10.0 * np.log10(np.clip(so[:, m, :] / (np.float32(gmax) + np.float32(1e-20)), 0, 1) + np.float32(1e-12))
        ref = (db - db.min()) / (db.max() - db.min()) if db.max() - db.min() > 1e-6 else np.zeros_like(db)
        ok &= img.shape == ref.shape and bool(np.abs(img - ref).max() <= 3e-3)
        dc.free()
        if hop % 2 == 0:
            n_mels = int(rng.choice([8, 40, 80, 128]))
            bank = MelBank(1024, fs, n_mels, 20.0, fs / 2 - 100.0)
            plan = engine.plan_for(spectro.get_window("hann", 1024), 1024, 1024, hop, _capi.DETREND[detrend], fs, 0, 0, _capi.F32)
            got = bank.stft_mel(x, plan, log_scale=False)
            mref = np.moveaxis(mel_oracle.mel_spectrogram(np.moveaxis(so, -1, -2), bank.weights.astype(np.float32), False), -1, -2)
            ok &= got.shape == mref.shape and bool(np.abs(got - mref).max() <= 1e-4 * mref.max())
            bank.close()
    except Exception as e:          # noqa: BLE001
        ok = False
        print("EXC", repr(e))
    if not ok:
        pbad += 1
        print("PRODUCT FAIL", it, dict(n_clips=n_clips, N=N, hop=hop, fs=fs, detrend=detrend, band=(fmin, fmax)))
print(f"{pcases - pbad}/{pcases} fused-product cases agree with the oracle")

# ---- batch APIs: pipelined ingest, resident clips, hop-shared sweep -- against the plain call on the same values ----
from spectro.pipeline import stft_pipelined
from spectro.sweep import hop_families
bbad, bcases = 0, max(cases // 10, 20)
for it in range(bcases):
    nper = int(rng.choice([256, 512, 1024, 2048, 4096, 1000]))
    hops = sorted({int(h) for h in rng.choice([16, 32, 64, 128, 256, 96, nper - nper // 8, int(rng.integers(1, nper + 1))], 3)})
    n_clips = int(rng.choice([1, 3, 8, 21, 40]))
    N = nper + int(rng.integers(0, 60)) * max(hops) + int(rng.integers(0, 200))
    dt = rng.choice(["f32", "f64", "i16"])
    x = rng.standard_normal((n_clips, N)) * rng.uniform(0.05, 2.0) + rng.uniform(-0.3, 0.3)
    x = (x * 4000).astype(np.int16) if dt == "i16" else x.astype(np.float32 if dt == "f32" else np.float64)
    fs = float(rng.choice([500.0, 48000.0]))
    ok = True
    why = []
    try:
        dc = engine.DeviceClips(x)
        for hop in hops:
            kw = dict(fs=fs, nperseg=nper, window="hann", noverlap=nper - hop)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                f, t, s = spectro.spectrogram(x, **kw)
                fo, to, so = orc.spectrogram(x, **kw)
            tol = 2e-4 if s.dtype == np.float32 else 1e-9
            ref = np.abs(so).max(axis=-2, keepdims=True)
            c = np.array_equal(t, to) and bool(np.all(np.abs(s - so) <= tol * ref + 1e-30)); ok &= c; why += [] if c else [f"plain hop {hop}"]
            fp, tp, sp_ = stft_pipelined(x, chunk_bytes=int(rng.choice([1, 1 << 16, 1 << 20, 1 << 26])), **kw)
            c = np.array_equal(tp, to) and bool(np.all(np.abs(sp_ - so) <= tol * ref + 1e-30)); ok &= c; why += [] if c else [f"pipelined hop {hop}"]
            dev = dc.stft(fs=fs, window="hann", nperseg=nper, hop=hop)
            sd = dev.to_host()
            dev.free()
            c = sd.shape == so.shape and bool(np.all(np.abs(sd - so) <= tol * ref + 1e-30)); ok &= c; why += [] if c else [f"DeviceClips hop {hop}"]
        for g, fam in hop_families(hops):                      # coarser hops are row subsets of the family's transform
            if len(fam) > 1:
                dev = dc.stft(fs=fs, window="hann", nperseg=nper, hop=g)
                base = dev.to_host()
                dev.free()
                for h in fam:
                    dev = dc.stft(fs=fs, window="hann", nperseg=nper, hop=h)
                    own = dev.to_host()
                    dev.free()
                    c = bool(np.array_equal(base[..., ::h // g][..., :own.shape[-1]], own)); ok &= c; why += [] if c else [f"family g={g} h={h} not identical"]
        dc.free()
    except Exception as e:          # noqa: BLE001
        ok = False
        print("EXC", repr(e))
    if not ok:
        bbad += 1
        print("BATCH FAIL", it, why, dict(n_clips=n_clips, N=N, dt=str(dt), nperseg=nper, hops=hops, fs=fs))
print(f"{bcases - bbad}/{bcases} batch-API cases agree with the oracle")

# ---- what PlotEngine does with one GUI-sized signal (A8-A13): mask, store, display image, HMM features, band powers ----
gbad, gcases = 0, max(cases // 5, 40)
for it in range(gcases):
    nper = int(rng.integers(1, 257)) * 32                      # the GUI's spin box: 32 ... 8192 in steps of 32
    N = int(rng.integers(200, 60000))
    fs = float(rng.choice([500.0, 2000.0, 10000.0, 20000.0]))
    f64 = rng.random() < 0.7                                   # the reference's recordings are float64
    x = rng.standard_normal(N) * rng.uniform(0.01, 50.0) + rng.uniform(-5, 5)
    x = x if f64 else x.astype(np.float32)
    fmin, fmax = sorted(rng.uniform(0, fs / 2 * 1.1, 2))
    if rng.random() < 0.3:
        fmin = 0.0
    log_scale = bool(rng.random() < 0.6)
    ok, why = True, []

    def chk(name, cond):
        global ok
        if not cond:
            ok = False
            why.append(name)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fo, to, so = orc.spectrogram(x, fs=fs, nperseg=nper)
            dev = engine.stft(x, fs=fs, nperseg=nper)
        gm = None if rng.random() < 0.5 else float(so.max() * rng.uniform(0.5, 3.0)) if so.size else None
        lf, lt, lsxx, img = orc.plot_image(fo, to, so, fmin, fmax, log_scale, gm)
        k_lo, k_hi = engine.bin_range(dev.f, fmin, fmax)
        chk("f/t/mask", np.array_equal(dev.f, fo) and np.array_equal(dev.t, to) and (k_hi - k_lo + 1) == lf.size)
        if lf.size and so.shape[-1]:
            tol = 1e-9 if f64 else 2e-4
            sl = dev.band_slice(k_lo, k_hi)
            chk("band_slice", sl.shape == lsxx.shape and bool(np.all(np.abs(sl - lsxx) <= tol * np.abs(so).max() + 1e-300)))
            if f64 or not log_scale:                           # the f32 log image is ill-conditioned at its darkest bin (tests/test_gpu_db.py)
                got = dev.image(k_lo, k_hi, log_scale, gm)
                chk("image", got.shape == img.shape and bool(np.abs(got - img).max() <= (1e-6 if f64 else 2e-4)))
            feats = dev.features(k_lo, k_hi)
            lp = np.log10(lsxx.sum(axis=0) + 1e-20)
            chk("features", bool(np.allclose(np.asarray(feats)[..., 0].reshape(-1), lp, atol=1e-9 if f64 else 3e-5)))
            bands = [(0.0, 4.0), (4.0, 8.0), (8.0, 13.0), (13.0, 30.0), (30.0, 80.0), (80.0, 250.0)]
            bp = orc.band_powers(lf, lsxx, {str(i): b for i, b in enumerate(bands)})
            ranges = [(int(np.searchsorted(lf, lo, "left")) + k_lo, int(np.searchsorted(lf, hi, "left")) + k_lo) for lo, hi in bands]
            tot = dev.band_totals([(k_lo, k_hi + 1)] + ranges)
            if tot[0] >= 1e-18:
                # f32: a bin is good to 1e-4 of the frame's LARGEST bin, not of itself -- a band of one or
                # two weak bins (case 157 of seed 3141: one 210-sample frame through the f32 chirp-z kernel) cannot be held to rtol 1e-4
                want = np.array([bp[str(i)] for i in range(len(bands))])
                got_r = np.asarray(tot[1:]) / tot[0]
                if f64:
                    chk("band_powers", bool(np.allclose(got_r, want, rtol=1e-9, atol=1e-12)))
                else:
                    n_el = np.array([max(hi - lo, 0) * lsxx.shape[-1] for lo, hi in ranges], np.float64)
                    chk("band_powers", bool(np.all(np.abs(got_r - want) <= 1e-4 * np.abs(want) + 2e-4 * float(np.abs(so).max()) * n_el / tot[0] + 1e-12)))
        dev.free()
    except Exception as e:          # noqa: BLE001
        ok = False
        print("EXC", repr(e))
    if not ok:
        gbad += 1
        print("GUI FAIL", it, why, dict(N=N, nperseg=nper, fs=fs, f64=bool(f64), band=(fmin, fmax), log_scale=log_scale))
print(f"{gcases - gbad}/{gcases} GUI-flow cases agree with the oracle")

# ---- streaming (cfg5): ragged chunks in, frames out == the offline call on the whole recording ----
from spectro.stream import StreamingSTFT
sbad, scases = 0, max(cases // 20, 10)
for it in range(scases):
    nper = int(rng.choice([96, 256, 512, 1024, 2048, 4096, 1000]))
    hop = int(rng.choice([nper // 4, nper // 2, nper, max(1, nper // 16)])) if rng.random() < 0.6 else int(rng.integers(1, nper + 1))
    n_ch = int(rng.choice([1, 2, 8]))
    total = nper * int(rng.integers(2, 9)) + int(rng.integers(0, 999))
    max_chunk = int(rng.choice([64, 500, 4096, 10000]))
    fs = float(rng.choice([8000.0, 96000.0]))
    x = (rng.standard_normal((n_ch, total)) * 0.3 + 0.05).astype(np.float32)
    ok = True
    try:
        st = StreamingSTFT(n_ch, fs, nper, hop, window="hann", max_chunk=max_chunk, transport=("auto", "device")[it % 2])   # pinned-host rows / HBM rows
        pos, ts, outs = 0, [], []
        while pos < total:
            n = min(int(rng.integers(0, max_chunk + 1)), total - pos)
            t, sx = st.feed(x[:, pos:pos + n])
            ts.append(t); outs.append(sx)
            pos += n
        st.close()
        t_all, s_all = np.concatenate(ts), np.concatenate(outs, axis=-1)
        fo, to, so = orc.spectrogram(x, fs=fs, nperseg=nper, window="hann", noverlap=nper - hop)
        ref = np.abs(so).max(axis=-2, keepdims=True)
        ok = s_all.shape == so.shape and np.array_equal(t_all, to) and bool(np.all(np.abs(s_all - so) <= 2e-4 * ref + 1e-30))
    except Exception as e:          # noqa: BLE001
        ok = False
        print("EXC", repr(e))
    if not ok:
        sbad += 1
        print("STREAM FAIL", it, dict(n_ch=n_ch, total=total, nperseg=nper, hop=hop, max_chunk=max_chunk, fs=fs))
print(f"{scases - sbad}/{scases} streaming cases agree with the oracle")
sys.exit(1 if (bad or pbad or bbad or gbad or sbad) else 0)
