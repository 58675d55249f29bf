"""CPU-only tests: the C ABI loads and exports every declared symbol, host-side logic, window tables,
the SweepManager surface, WAV decoding, sharding arithmetic, and a world_size-2 gloo run of the
distributed plumbing.  No device compute here (there is no GPU in the build container)."""
import os
import re
import struct
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, PKG


def test_abi_exports_every_declared_symbol():
    """Every function declared in include/spectro.h is exported by libspectro.so and bound in _capi."""
    import ctypes
    from spectro import _capi
    header = open(os.path.join(ROOT, "include", "spectro.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = _capi.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in spectro.h but not exported"
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    assert lib.sg_version() == 103


def test_host_shim_under_address_and_ub_sanitizers():
    """the host side of the ABI (argument triage, f / t vectors, mel bank construction, jet table, error
    strings -- csrc/host_shim.cpp, the very file linked into libspectro.so) built with g++ -fsanitize=address,undefined
    and driven by tests/asan_driver.cpp.  GPU sanitizers are not available on the pool; this is the CPU build."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("spectro_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    exe = mod.build_sanitizer_driver()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "checks passed under AddressSanitizer" in r.stdout and "runtime error" not in r.stderr


def test_abi_host_side_helpers_without_gpu():
    from spectro import _capi
    import scipy.fft
    for n, fs in [(512, 16000.0), (33, 1000.0), (1000, 500.0), (1024, 48000.0), (96, 44100.0)]:
        np.testing.assert_array_equal(_capi.freqs(n, fs), scipy.fft.rfftfreq(n, 1 / fs))
    for N, n, hop, fs in [(16000, 512, 448, 16000.0), (480000, 1024, 256, 48000.0), (500, 33, 29, 1000.0),
                          (10000, 1000, 875, 500.0), (256, 256, 224, 1000.0), (100, 256, 10, 1.0)]:
        ref = np.arange(n / 2, N - n / 2 + 1, hop) / float(fs) if N >= n else np.zeros(0)
        np.testing.assert_array_equal(_capi.times(N, n, hop, fs), ref)


def test_no_cpu_fallback_without_device():
    """Product path must fail loudly when no gfx950 device is present (this container has none)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import spectro
    with pytest.raises(Exception) as ei:
        spectro.spectrogram(np.zeros(4096, np.float32), fs=1000.0, nperseg=256)
    assert not isinstance(ei.value, (AssertionError,))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn
                assert "scipy" not in src or fn in ("windows.py", "signal.py", "PlotEngine.py", "SweepManager.py", "engine.py") \
                    or "import scipy" not in src
    for fn in ("signal.py", "engine.py", "windows.py", "_capi.py", "dist.py"):
        src = open(os.path.join(PKG, "spectro", fn)).read()
        assert "import scipy" not in src and "from scipy" not in src, fn


def test_windows_match_scipy():
    import scipy.signal as ss
    from spectro.windows import get_window
    for name in ["boxcar", "triang", "bartlett", "hann", "hamming", "blackman", "nuttall", "blackmanharris", "flattop",
                 "cosine", ("tukey", 0.25), ("tukey", 0.5), ("tukey", 0.0), ("tukey", 1.0), ("kaiser", 8.6),
                 ("gaussian", 7.0), ("general_hamming", 0.6)]:
        for n in (1, 2, 32, 33, 1000, 1024):
            assert np.abs(get_window(name, n) - ss.get_window(name, n)).max() < 1e-14, (name, n)
    np.testing.assert_array_equal(get_window(("tukey", 0.25), 1024), ss.get_window(("tukey", 0.25), 1024))
    with pytest.raises(ValueError):
        get_window("nope", 16)
    with pytest.raises(ValueError):
        get_window("kaiser", 16)


def test_resolve_segments_rules():
    from spectro.signal import resolve_segments, compute_dtype
    with pytest.warns(UserWarning, match="nperseg = 256 is greater than input length"):
        w, n = resolve_segments(("tukey", .25), None, 100)
    assert n == 100 and len(w) == 100
    w, n = resolve_segments(np.ones(64), None, 100)
    assert n == 64
    with pytest.raises(ValueError):
        resolve_segments(np.ones(64), 32, 100)
    with pytest.raises(ValueError):
        resolve_segments(np.ones(200), None, 100)
    with pytest.raises(ValueError):
        resolve_segments(np.ones((2, 2)), None, 100)
    assert compute_dtype(np.float32) == np.float32 and compute_dtype(np.int16) == np.float32
    assert compute_dtype(np.float64) == np.float64 and compute_dtype(np.int32) == np.float64


def test_bin_range_equals_mask():
    from spectro.engine import bin_range
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(2, 600))
        fs = float(rng.choice([100.0, 500.0, 16000.0, 48000.0]))
        f = np.fft.rfftfreq(n, 1 / fs)
        fmin, fmax = sorted(rng.uniform(-10, fs / 2 + 10, 2))
        if rng.random() < 0.3:
            fmin = float(f[rng.integers(0, len(f))])
        if rng.random() < 0.3:
            fmax = float(f[rng.integers(0, len(f))])
        lo, hi = bin_range(f, fmin, fmax)
        mask = (f >= fmin) & (f <= fmax)
        idx = np.nonzero(mask)[0]
        if len(idx) == 0:
            assert lo > hi
        else:
            assert (lo, hi) == (idx[0], idx[-1])


def test_sweepmanager_surface(tmp_path):
    from SweepManager import SweepManager
    from oracle import stft_oracle as orc
    sm = SweepManager()
    assert sm.data == {}
    with pytest.raises(ValueError, match="Unsupported file type: .txt"):
        sm.load_file("x.txt")
    raw, proc = np.arange(3.0), np.arange(4.0)
    sm.data.update({
        "abf": {"fs_raw": 10.0, "fs": 10.0, "raw": raw, "processed": None},
        "h5": {"fs_raw": 20.0, "fs": 5.0, "raw": raw, "processed": proc},
        "nofsraw": {"fs": 7.0, "raw": raw, "processed": proc},
        "noraw": {"fs": 7.0, "raw": None, "processed": None},
        "nofs": {"raw": raw, "processed": proc},
    })
    for name in ("abf", "h5", "nofsraw"):
        for p in (False, True):
            got, want = sm.get_signal(name, processed=p), orc.get_signal(sm.data, name, processed=p)
            assert got[0] is want[0] and got[1] == want[1]
    for name, p in (("missing", False), ("noraw", False), ("noraw", True), ("nofs", False), ("nofs", True)):
        with pytest.raises(KeyError) as e1:
            sm.get_signal(name, processed=p)
        with pytest.raises(KeyError) as e2:
            orc.get_signal(sm.data, name, processed=p)
        assert str(e1.value) == str(e2.value)


def _write_wav(path, fs, data, kind=1, bits=16):
    n_ch = data.shape[1]
    if kind == 1 and bits == 16:
        payload = data.astype("<i2").tobytes()
    elif kind == 1 and bits == 24:
        v = data.astype(np.int32).reshape(-1)
        payload = b"".join(int(x & 0xFFFFFF).to_bytes(3, "little") for x in v)
    elif kind == 1 and bits == 8:
        payload = (data + 128).astype(np.uint8).tobytes()
    elif kind == 1 and bits == 32:
        payload = data.astype("<i4").tobytes()
    else:
        payload = data.astype("<f4").tobytes()
    fmt = struct.pack("<HHIIHH", kind, n_ch, fs, fs * n_ch * bits // 8, n_ch * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\0" \
        + b"data" + struct.pack("<I", len(payload)) + payload
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_wav_loader(tmp_path):
    from SweepManager import SweepManager
    rng = np.random.default_rng(1)
    for kind, bits, lo, hi in [(1, 16, -32768, 32767), (1, 24, -(1 << 23), (1 << 23) - 1), (1, 8, -128, 127),
                               (1, 32, -(1 << 31), (1 << 31) - 1), (3, 32, -1, 1)]:
        data = rng.integers(lo, hi, size=(777, 2)) if kind == 1 else rng.uniform(lo, hi, size=(777, 2)).astype(np.float32)
        p = tmp_path / f"clip{bits}_{kind}.wav"
        _write_wav(str(p), 16000, data, kind, bits)
        sm = SweepManager()
        names = sm.load_file(str(p))
        assert names == [f"clip{bits}_{kind}_sweep0", f"clip{bits}_{kind}_sweep1"]
        for c, n in enumerate(names):
            sig, fs = sm.get_signal(n)
            assert fs == 16000.0
            np.testing.assert_array_equal(sig, data[:, c])
            assert sm.data[n]["sweep_idx"] == c and sm.data[n]["processed"] is None
        if bits == 16:
            assert sm.get_signal(names[0])[0].dtype == np.int16
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"nope")
    with pytest.raises(ValueError):
        SweepManager().load_file(str(bad))


def test_plotengine_surface_without_device():
    """Attribute / method surface GUI.py and ExportManager.py rely on."""
    from PlotEngine import PlotEngine
    pe = PlotEngine(parent=None)
    for attr in ("burst_patches", "spec_data_source", "currently_plotted_items", "last_Sxx", "last_t", "last_f",
                 "segment_map", "fig", "ax_signal", "ax_spec", "is_model_refined", "last_fs", "last_settings",
                 "editing_enabled", "ROI_COLOR", "HOVER_COLOR"):
        assert hasattr(pe, attr), attr
    for meth in ("plot_sweeps", "plot_extra", "set_editing_enabled", "draw", "clear", "reset_model", "unsupervised_detect",
                 "learn_and_detect", "plot_detection_lines", "calculate_absolute_power", "calculate_band_powers",
                 "_plot_spectrogram", "_calculate_features", "plot_single_signal", "remove_patch"):
        assert callable(getattr(pe, meth)), meth
    assert not hasattr(pe, "last_detected_events") and not hasattr(pe, "last_raw_t")     # H9: created in clear()
    pe.clear()
    assert pe.last_detected_events == [] and len(pe.last_raw_t) == 0
    assert pe.calculate_absolute_power() is None and pe.calculate_band_powers() is None
    with pytest.raises(ValueError, match="Please plot a spectrogram before detecting."):
        pe.unsupervised_detect()
    with pytest.raises(ValueError, match="Please plot a spectrogram before learning."):
        pe.learn_and_detect()
    pe.plot_detection_lines([(0.1, 0.2), (0.5, 0.9)])
    assert len(pe.burst_patches) == 2 and pe.burst_patches[0][0].event_data == (0.1, 0.2)
    pe.plot_detection_lines([])
    assert pe.burst_patches == []
    assert PlotEngine._merge_overlapping_events([(3, 4), (0, 1), (0.5, 2)]) == [(0, 2), (3, 4)]


def test_shard_arithmetic():
    from spectro.dist import shard_range, deal_work_items, stft_cost
    for n in (0, 1, 7, 64, 65, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    costs = [stft_cost(480000, n, h) for n in (256, 512, 1024, 2048, 4096) for h in (64, 128, 256)] * 4
    deal = deal_work_items(costs, 8)
    assert sorted(i for part in deal for i in part) == list(range(len(costs)))
    loads = [sum(costs[i] for i in part) for part in deal]
    assert max(loads) / (sum(loads) / 8) < 1.15


def test_hop_families():
    from spectro.sweep import hop_families
    assert hop_families([64, 128, 256]) == [(64, [64, 128, 256])]            # BASELINE cfg4: one transform per n_fft
    assert hop_families([256, 64, 128, 64]) == [(64, [64, 128, 256])]
    assert hop_families([64, 96]) == [(64, [64]), (96, [96])]                  # gcd 32 would cost more than both
    assert hop_families([100]) == [(100, [100])]
    assert hop_families([32, 48, 64]) == [(32, [32, 64]), (48, [48])]
    assert hop_families([6, 10, 15]) == [(6, [6]), (10, [10]), (15, [15])]
    for g, fam in hop_families([3, 6, 7, 14, 12, 28]):
        assert all(h % g == 0 for h in fam) and 1.0 / g <= sum(1.0 / h for h in fam) + 1e-12
        assert all(h % 2 == g % 2 for h in fam)
    # an odd hop never shares with even ones: on f64 / nfft 256, 512, 2048, 4096 plans it runs on another kernel, and a family's
    # rows are only IDENTICAL to the separate transforms' when one kernel computes them all (found by the GPU fuzz, seed 1618)
    assert hop_families([1, 64, 256]) == [(1, [1]), (64, [64, 256])]
    assert hop_families([3, 9, 27]) == [(3, [3, 9, 27])] and hop_families([1, 2, 4]) == [(1, [1]), (2, [2, 4])]


def test_frame_shards_of_one_long_clip():
    """a single long recording is cut at frame boundaries; every frame has one owner and an owner's sample range
    holds exactly its frames (the halo of nperseg - hop samples is read by both neighbours)."""
    from spectro.dist import frame_shards, n_frames
    for n_samples, nperseg, hop in [(480000, 1024, 256), (30000, 256, 224), (1024, 1024, 1), (1023, 1024, 7), (5000, 1000, 999),
                                    (100000, 4096, 64)]:
        total = n_frames(n_samples, nperseg, hop)
        for world in (1, 2, 3, 8, 50):
            sh = frame_shards(n_samples, nperseg, hop, world)
            assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
            for f_lo, f_hi, s_lo, s_hi in sh:
                assert 0 <= s_lo <= s_hi <= n_samples
                assert n_frames(s_hi - s_lo, nperseg, hop) == f_hi - f_lo
                if f_hi > f_lo:
                    assert s_lo == f_lo * hop and s_hi == (f_hi - 1) * hop + nperseg
            for a, b in zip(sh, sh[1:]):
                if a[1] > a[0] and b[1] > b[0]:
                    assert a[3] - b[2] == nperseg - hop          # the halo both neighbours read


GLOO_SCRIPT = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
    import numpy as np, torch, torch.distributed as dist
    from spectro import dist as sd
    from oracle import stft_oracle as orc
    dist.init_process_group("gloo")
    world, rank = sd.world_info()
    assert world == 2
    # 5 clips shard 3 + 2; the per-shard "compute" is the CPU oracle standing in for the device call
    x = (np.random.default_rng(3).standard_normal((5, 6000)) * 0.1).astype(np.float32)
    def compute(a, b):
        f, t, s = orc.spectrogram(x[a:b], fs=8000.0, nperseg=256, window="hann", noverlap=192)
        return torch.from_numpy(np.ascontiguousarray(np.moveaxis(s, -1, -2)))      # [clip, frame, bin]
    (a, b), local = sd.run_sharded(5, compute)
    assert (a, b) == sd.shard_range(5, 2, rank)
    gmax = sd.global_max(local.max().reshape(1).clone())
    feats = torch.log10(local[..., 3:40].sum(-1) + 1e-20)                            # reduced product [clip, frame]
    gathered = sd.gather_to_root([feats, local], dst=0)
    _, _, full = orc.spectrogram(x, fs=8000.0, nperseg=256, window="hann", noverlap=192)
    full = np.moveaxis(full, -1, -2)
    assert abs(float(gmax) - full.max()) == 0.0
    if rank == 0:
        got = torch.cat([g[1] for g in gathered]).numpy()
        assert got.shape == full.shape and np.array_equal(got, full), "sharded != single"
        gf = torch.cat([g[0] for g in gathered]).numpy()
        assert np.allclose(gf, np.log10(full[..., 3:40].sum(-1) + 1e-20))
        eq = sd.gather_equal(torch.full((2,), float(rank)))
    else:
        assert gathered is None
        eq = sd.gather_equal(torch.full((2,), float(rank)))
    assert [float(e[0]) for e in eq] == [0.0, 1.0]
    # gather_equal to ONE rank: that rank gets one tensor per rank, in rank order; the others get None
    only1 = sd.gather_equal(torch.full((3,), float(10 + rank)), dst=1)
    if rank == 1:
        assert len(only1) == 2 and [float(e[0]) for e in only1] == [10.0, 11.0] and all(e.shape == (3,) for e in only1)
    else:
        assert only1 is None
    # cfg4: sharded parameter sweep; the per-item "device call" is the oracle here (no GPU in this container)
    from spectro import sweep
    clips = (np.random.default_rng(9).standard_normal((3, 4000)) * 0.1).astype(np.float32)
    def item(clip, n, h):
        f, t, s = orc.spectrogram(clips[clip], fs=8000.0, nperseg=n, window="hann", noverlap=n - h)
        return np.log10(s.sum(axis=0) + 1e-20).astype(np.float32)
    res = sweep.sharded_sweep(clips, 8000.0, [128, 256, 512], [32, 64], compute=item, dst=0)
    if rank == 0:
        assert len(res) == 3 * 3 * 2
        for (clip, n, h), v in res.items():
            assert np.array_equal(v, item(clip, n, h)), (clip, n, h)
    else:
        assert res is None
    # the batched deal (contiguous clip ranges per pair, one upload per rank, one call per pair) equals the per-item one;
    # the oracle stands in for DeviceClips here and counts its calls
    calls = {"open": 0, "run": 0}
    def open_batch(xs):
        calls["open"] += 1
        def run(n, h, a, b):
            calls["run"] += 1
            f, t, s = orc.spectrogram(xs[a:b], fs=8000.0, nperseg=n, window="hann", noverlap=n - h)
            return np.log10(s.sum(axis=-2) + 1e-20).astype(np.float32)          # [clip, frame]
        return run
    blocks = sweep.clip_blocks(3, [128, 256, 512], [32, 64], 2)
    for pair, ranges in blocks.items():
        assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 3
    sizes = [sum(r[k][1] - r[k][0] for r in blocks.values()) for k in (0, 1)]
    assert sizes[0] == sizes[1] == 9                                              # 6 pairs x 3 clips, the odd clip alternates
    res_b = sweep.sharded_sweep(clips, 8000.0, [128, 256, 512], [32, 64], batch_compute=open_batch, dst=0)
    assert calls["open"] == 1 and calls["run"] == 6                               # NOT one call per (clip, pair)
    if rank == 0:
        assert res_b.keys() == res.keys()
        for k in res:
            assert np.array_equal(res_b[k], res[k]), k
    else:
        assert res_b is None
    # hops that divide each other share one transform per n_fft (sweep.hop_families): 3 calls instead of 6, same values
    calls2 = {"run": 0}
    def open_family_batch(xs):
        def run(n, h, a, b):
            calls2["run"] += 1
            f, t, s = orc.spectrogram(xs[a:b], fs=8000.0, nperseg=n, window="hann", noverlap=n - h)
            return np.log10(s.sum(axis=-2) + 1e-20).astype(np.float32)
        def family(n, g, members):
            a0, b0 = min(m[1] for m in members), max(m[2] for m in members)
            base = run(n, g, a0, b0)
            return [base[a - a0:b - a0, ::h // g][:, :sd.n_frames(xs.shape[1], n, h)] for h, a, b in members]
        run.family = family
        return run
    res_f = sweep.sharded_sweep(clips, 8000.0, [128, 256, 512], [32, 64], batch_compute=open_family_batch, dst=0)
    assert calls2["run"] == 3
    if rank == 0:
        assert res_f.keys() == res.keys()
        for k in res:
            assert np.array_equal(res_f[k], res[k]), k
    # n_fft beyond the clip length: the device call clamps nperseg to the clip (one frame), and so must the gather's shape table
    short = (np.random.default_rng(11).standard_normal((3, 400)) * 0.1).astype(np.float32)
    def open_clamped(xs):
        def run(n, h, a, b):
            nps = min(n, xs.shape[1])
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                f, t, s = orc.spectrogram(xs[a:b], fs=8000.0, nperseg=nps, window="hann", noverlap=nps - h)
            return np.log10(s.sum(axis=-2) + 1e-20).astype(np.float32)
        return run
    assert sweep.sweep_frames(400, 512, 64) == 1 and sweep.sweep_frames(400, 256, 64) == 3 and sweep.sweep_frames(0, 256, 64) == 0
    res_c = sweep.sharded_sweep(short, 8000.0, [256, 512], [64], batch_compute=open_clamped, dst=0)
    if rank == 0:
        assert len(res_c) == 6 and all(v.shape == ((1,) if k[1] == 512 else (3,)) for k, v in res_c.items())
    # a product whose shape is not the one every rank derives is refused on every rank, before anything is sent
    def open_wrong(xs):
        return lambda n, h, a, b: np.zeros((b - a, 2), np.float32)
    try:
        sweep.sharded_sweep(short, 8000.0, [512], [64], batch_compute=open_wrong, dst=0)
        raise SystemExit("a mis-shaped sweep product was not refused")
    except ValueError:
        pass
    # ... also when only ONE rank holds the odd product: the others must not walk into the gather and wait there
    def open_wrong_on_rank1(xs):
        return lambda n, h, a, b: np.zeros((b - a, 2 if rank == 1 else 1), np.float32)
    try:
        sweep.sharded_sweep(short, 8000.0, [512], [64], batch_compute=open_wrong_on_rank1, dst=0)
        raise SystemExit("rank %d went on although rank 1's product was mis-shaped" % rank)
    except ValueError:
        pass
    assert sd.all_agree(True) and not sd.all_agree(rank == 0)
    # gather_to_root sizes a receive buffer with the SENDER's dtype: rank 1 sends float64 (and an int16 block) while the root
    # holds float32 / nothing -- with the table exchanged, and with a table of (shape, dtype) entries given by the caller
    f64 = torch.arange(6, dtype=torch.float64).reshape(2, 3) / 7.0
    i16 = torch.arange(-3, 4, dtype=torch.int16)
    for table in (None, [[((4,), torch.float32)], [((2, 3), "float64"), ((7,), torch.int16)]]):
        mine = [torch.full((4,), 0.5, dtype=torch.float32)] if rank == 0 else [f64, i16]
        got = sd.gather_to_root(mine, dst=0, shapes=table)
        if rank == 0:
            assert got[1][0].dtype == torch.float64 and torch.equal(got[1][0], f64), got[1][0]
            assert got[1][1].dtype == torch.int16 and torch.equal(got[1][1], i16)
        else:
            assert got is None
    got = sd.gather_to_root([] if rank == 0 else [f64], dst=0)                       # the root holds nothing at all
    assert got is None if rank else (got[0] == [] and torch.equal(got[1][0], f64))
    # a table of bare shapes: one dtype everywhere is settled by a small all-gather (the root's own dtype no longer decides) ...
    got = sd.gather_to_root([] if rank == 0 else [f64], dst=0, shapes=[[], [(2, 3)]])
    assert got is None if rank else (got[1][0].dtype == torch.float64 and torch.equal(got[1][0], f64))
    # ... and ranks that disagree (f32 on the root, f64 on the peer) all raise, before any send is posted
    try:
        sd.gather_to_root([torch.zeros(2, 3)] if rank == 0 else [f64], dst=0, shapes=[[(2, 3)], [(2, 3)]])
        raise SystemExit("rank %d: mixed dtypes under a table without dtypes were not refused" % rank)
    except ValueError:
        pass
    # tensors that contradict the table: every rank raises together, whichever rank holds them
    try:
        sd.gather_to_root([torch.zeros(2, 3)] if rank == 0 else [f64], dst=0, shapes=[[((2, 3), "float32")], [((2, 3), "float32")]])
        raise SystemExit("rank %d: a tensor that contradicts the table was not refused" % rank)
    except ValueError:
        pass
    dist.barrier(); dist.destroy_process_group()
    os.write(1, ("rank %d ok" % rank + chr(10)).encode())   # one write per rank: print() pieces of two ranks can interleave
''')


def test_gloo_world2_sharding(tmp_path):
    script = tmp_path / "gloo_shard.py"
    script.write_text(GLOO_SCRIPT)
    import socket
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for attempt in range(2):                         # the rendezvous port can be taken between probing and binding
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, PKG],
                           capture_output=True, text=True, timeout=300, env=env)
        if r.returncode == 0:
            break
        print(f"[gloo test] attempt {attempt} failed (rc {r.returncode}); stderr tail:\n{r.stderr[-1500:]}", file=sys.stderr)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_mel_weights_host_function_matches_own_oracle():
    import ctypes as C
    from oracle import mel_oracle
    from spectro import _capi
    for nfft, fs, nm, lo, hi in [(1024, 48000.0, 80, 0.0, 24000.0), (512, 16000.0, 40, 50.0, 7600.0), (4096, 96000.0, 128, 20.0, 20000.0)]:
        w = np.empty((nfft // 2 + 1, nm))
        _capi.check(_capi.lib().sg_mel_weights(nfft, fs, nm, lo, hi, w.ctypes.data_as(C.POINTER(C.c_double))))
        ref = mel_oracle.mel_weights(nfft, fs, nm, lo, hi)
        assert np.abs(w - ref).max() < 1e-12
        assert w.min() >= 0 and w.max() <= 1.0 + 1e-12
        nt = (nm + 15) // 16
        klo, khi = (C.c_int * nt)(), (C.c_int * nt)()
        _capi.check(_capi.lib().sg_mel_tile_ranges(w.ctypes.data_as(C.POINTER(C.c_double)), nfft // 2 + 1, nm, klo, khi))
        for t in range(nt):
            blk = w[:, 16 * t:16 * t + 16]
            nz = np.nonzero(blk.any(axis=1))[0]
            assert klo[t] <= nz[0] and khi[t] >= nz[-1] + 1 and klo[t] % 4 == 0 and khi[t] % 4 == 0


def test_mel_weights_pinned_to_the_published_htk_definition():
    """X1: the mel bank against the PUBLISHED definition, not against this repository's own restatement: HTK mel scale
    m = 2595 log10(1 + f/700) (Young et al., HTK Book, eq. 5.13), n_mels + 2 band edges equally spaced in mel between
    fmin and fmax, triangular filters of unit peak evaluated at the rfft bin frequencies (what
    torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") and librosa.filters.mel(htk=True, norm=None)
    compute).  Hand-derived vectors (pocket-calculator arithmetic from the closed form, cfg3's bank: nfft 1024, 48 kHz,
    80 bands, 0..24 kHz) plus the closed form evaluated independently here."""
    import ctypes as C
    import math
    from spectro import _capi
    nfft, fs, nm, lo, hi = 1024, 48000.0, 80, 0.0, 24000.0
    w = np.empty((nfft // 2 + 1, nm))
    _capi.check(_capi.lib().sg_mel_weights(nfft, fs, nm, lo, hi, w.ctypes.data_as(C.POINTER(C.c_double))))
    # mel(24 kHz) = 2595 log10(1 + 24000/700) = 2595 * 1.5475989... = 4016.0192; 81 equal steps of 49.58048 mel;
    # edge j sits at 700 (10^(j * 49.58048 / 2595) - 1) Hz:
    edges = {1: 31.48293609949191, 40: 3367.657865996134, 41: 3550.603312667209, 80: 22936.91502114318, 81: 24000.0}
    mel_hi = 2595.0 * math.log10(1.0 + 24000.0 / 700.0)
    assert abs(mel_hi - 4016.019179871836) < 1e-9
    for j, f_hz in edges.items():
        assert abs(700.0 * (10.0 ** (j * mel_hi / 81.0 / 2595.0) - 1.0) - f_hz) < 1e-8
    # band b spans edges b, b+1 (peak), b+2; bin k is at k * 46.875 Hz.  (bin, band) -> weight:
    #   bin 72 = 3375.000 Hz on the falling side of band 39: (3550.6033 - 3375) / (3550.6033 - 3367.6579) = 0.95986709
    #   bin 75 = 3515.625 Hz on the rising side of band 40:  (3515.625 - 3367.6579) / (3550.6033 - 3367.6579) = 0.80880468
    #   bin 500 = 23437.5 Hz, falling side of the last band: (24000 - 23437.5) / (24000 - 22936.9150) = 0.52912045
    #   bin 1 = 46.875 Hz, falling side of band 0:           (64.3818369 - 46.875) / (64.3818369 - 31.4829361) = 0.53214048
    #   bin 9 = 421.875 Hz, rising side of band 10:          (421.875 - 386.8250360) / (435.7056691 - 386.8250360) = 0.71705217
    #   bin 221 lies outside band 60 (9105.45 .. 10007.30 Hz): 0
    hand = {(72, 39): 0.9598670853116857, (75, 40): 0.808804683015161, (500, 79): 0.5291204477415157, (1, 0): 0.5321404805975217,
            (9, 10): 0.7170521687589072, (221, 60): 0.0}
    for (k, b), v in hand.items():
        assert abs(w[k, b] - v) < 1e-12, (k, b, w[k, b], v)
    # the closed form over the whole bank, written out independently of oracle/mel_oracle.py (scalar loops)
    for nfft2, fs2, nm2, lo2, hi2 in [(1024, 48000.0, 80, 0.0, 24000.0), (512, 16000.0, 40, 50.0, 7600.0)]:
        w2 = np.empty((nfft2 // 2 + 1, nm2))
        _capi.check(_capi.lib().sg_mel_weights(nfft2, fs2, nm2, lo2, hi2, w2.ctypes.data_as(C.POINTER(C.c_double))))
        m_lo, m_hi = 2595.0 * math.log10(1.0 + lo2 / 700.0), 2595.0 * math.log10(1.0 + hi2 / 700.0)
        e = [700.0 * (10.0 ** ((m_lo + (m_hi - m_lo) * j / (nm2 + 1)) / 2595.0) - 1.0) for j in range(nm2 + 2)]
        for b in range(nm2):
            for k in range(nfft2 // 2 + 1):
                f = k * fs2 / nfft2
                ref = max(0.0, min((f - e[b]) / (e[b + 1] - e[b]), (e[b + 2] - f) / (e[b + 2] - e[b + 1])))
                assert abs(w2[k, b] - ref) < 1e-11, (k, b)
        assert abs(w2.max() - 1.0) < 0.2 and w2.min() == 0.0       # unit-peak triangles (no area normalisation)


def test_jet_lut_matches_matplotlib():
    import ctypes as C
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import colormaps
    from spectro import _capi
    lut = np.empty((256, 4), np.uint8)
    _capi.check(_capi.lib().sg_jet_lut(lut.ctypes.data_as(C.POINTER(C.c_uint8))))
    ref = (colormaps["jet"](np.arange(256)) * 255).astype(np.uint8)
    np.testing.assert_array_equal(lut, ref)          # same arithmetic as matplotlib's lookup table builder


def test_c_client_builds_and_fails_loudly_without_a_gpu():
    """The ABI from plain C (examples/c_client.c, gcc -std=c99 -Werror): compiles against include/spectro.h, links
    libspectro.so without Python, and on a machine with no GPU exits with the library's own message (no fallback)."""
    import importlib.util
    import subprocess
    from conftest import PKG
    spec = importlib.util.spec_from_file_location("spectro_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    exe = mod.build_c_client()
    r = subprocess.run([exe, "--probe"], capture_output=True, text=True, timeout=120)
    assert "libspectro ABI version" in r.stdout
    assert r.returncode in (0, 3)
    if r.returncode == 3:
        assert "sg_init" in r.stderr and "no HIP device" in r.stderr


def test_bench_spawns_its_own_ranks_and_refuses_a_wrong_world():
    """bench.py --gpus N without a launcher starts N ranks itself (as a child process, before any GPU call); under a launcher whose
    WORLD_SIZE differs from --gpus it exits non-zero instead of reporting another n_gpus.  No GPU needed for either check."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "7", "--warmup", "2", "--dry-run-spawn"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["spawn"]
    assert d["n_ranks"] == 4 and cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], bench)
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]            # the ranks get the caller's arguments
    # a launcher started 2 ranks but the line was asked for 8 GPUs
    r = subprocess.run([sys.executable, bench, "--gpus", "8"], capture_output=True, text=True, timeout=120,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_hip_runtime_preload_checks_the_soname(tmp_path):
    """torch's bundled HIP runtime is preloaded only when its SONAME is the name libspectro.so needs (both read from
    the ELF files); a torch built for another ROCm major is left alone, with a warning.  Child processes: the decision is made
    once per process, at the first lib() call."""
    sys.path.insert(0, PKG)
    from spectro import _capi
    soname, needed = _capi._elf_dynamic(_capi.LIB_PATH)
    hip = [n for n in needed if n.startswith("libamdhip64.so")]
    assert len(hip) == 1 and "libstdc++.so.6" in needed
    # a stand-in "torch" whose bundled runtime carries another SONAME (a shared object built here with gcc)
    fake = tmp_path / "torch"
    (fake / "lib").mkdir(parents=True)
    (fake / "__init__.py").write_text("raise ImportError('the stand-in torch must never be imported')\n")
    src = tmp_path / "x.c"
    src.write_text("int sg_fake_runtime(void) { return 6; }\n")
    subprocess.run(["gcc", "-shared", "-fPIC", "-Wl,-soname,libamdhip64.so.6", str(src), "-o", str(fake / "lib" / "libamdhip64.so")], check=True)
    assert _capi._elf_dynamic(str(fake / "lib" / "libamdhip64.so"))[0] == "libamdhip64.so.6"
    code = textwrap.dedent('''
        import sys, warnings
        sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])      # the stand-in torch shadows the real one
        from spectro import _capi
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            try:
                _capi.lib()
                err = ""
            except ImportError as e:
                err = str(e)
        print("choice=%s warned=%d err=%s" % (_capi.hip_runtime["choice"], sum("cannot be shared" in str(x.message) for x in w), err[:60]))
        with open("/proc/self/maps") as fh:
            print("mapped_fake=%d" % sum("libamdhip64.so" in l and sys.argv[1] in l for l in fh))
    ''')
    env = {k: v for k, v in os.environ.items() if k != "SPECTRO_HIP_RUNTIME"}
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path), PKG], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-800:]
    assert "choice=system warned=1 err=" in r.stdout and "mapped_fake=0" in r.stdout, r.stdout
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path), PKG], capture_output=True, text=True, env=dict(env, SPECTRO_HIP_RUNTIME="torch"), timeout=120)
    assert r.returncode == 0 and "err=SPECTRO_HIP_RUNTIME=torch" in r.stdout, r.stdout + r.stderr[-400:]
    # the real torch of this image: its SONAME matches, so it is the one mapped (and said so)
    real = textwrap.dedent('''
        import sys
        sys.path.insert(0, sys.argv[1])
        from spectro import _capi
        _capi.lib()
        print("choice=%s path=%s" % (_capi.hip_runtime["choice"], _capi.hip_runtime["path"]))
    ''')
    r = subprocess.run([sys.executable, "-c", real, PKG], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "choice=torch path=" in r.stdout and "torch/lib/libamdhip64.so" in r.stdout, r.stdout + r.stderr[-400:]
    r = subprocess.run([sys.executable, "-c", real, PKG], capture_output=True, text=True, env=dict(env, SPECTRO_HIP_RUNTIME="system"), timeout=120)
    assert r.returncode == 0 and "choice=system path=None" in r.stdout, r.stdout + r.stderr[-400:]


@pytest.mark.parametrize("tool", ["sim_rtiny.py", "sim_rbluew.py", "sim_rsmall.py"])
def test_lane_models_of_the_register_kernels(tool):
    """The numpy models the register kernels' index maps were written against (tools/sim_*.py: radix-8 + quad-DPP transform of stft_rtiny.hip,
    the decimation-in-time / chirp-z / split algebra of stft_rbluew*.hip, rsmall's maps incl. R = 1) still reproduce numpy's rfft: the CPU-side
    check of what the GPU tests check against the oracle."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", tool)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    errs = [float(m) for m in re.findall(r"max rel err ([0-9.eE+-]+)", out.stdout)]
    assert len(errs) >= 2 and max(errs) < 1e-12, out.stdout

