"""GPU parity of the post-processing rows A8-A13 and of the PlotEngine drop-in, against the goldens that
were produced by the reference's own PlotEngine methods (g1_reference_engine.npz) and against the oracle."""
import warnings

import numpy as np
import pytest

from conftest import assert_spec_close, cfg1_signal, eeg_like, load_golden
from oracle import stft_oracle as orc

pytestmark = pytest.mark.gpu

SIGNALS = {
    "cfg1_lin": lambda: cfg1_signal(), "cfg1_log": lambda: cfg1_signal(),
    "cfg1_band_log": lambda: cfg1_signal(), "cfg1_gmax": lambda: cfg1_signal(),
    "cfg1_f32": lambda: cfg1_signal().astype(np.float32),
    "eeg_default": eeg_like, "eeg_lin_256": eeg_like,
    "short_clamp": lambda: cfg1_signal()[:300],
    "zeros": lambda: np.zeros(4096), "const": lambda: np.full(4096, 2.5),
    "empty_mask": lambda: cfg1_signal(),
    "eeg_np2_1000": eeg_like,
}


def _settings(g, tag):
    fs, nper, fmin, fmax, log, gmax = g[f"{tag}__args"]
    return fs, {"nperseg": int(nper), "fmin": fmin, "fmax": fmax, "log_scale": bool(log), "mode_raw": "Spectrogram",
                "mode_proc": "None", "draw_raw": False, "draw_proc": False}, (None if gmax < 0 else gmax)


@pytest.mark.parametrize("tag", sorted(SIGNALS))
def test_plotengine_matches_reference_engine(tag):
    from PlotEngine import PlotEngine
    g = load_golden("g1_reference_engine.npz")
    fs, settings, gmax = _settings(g, tag)
    x = SIGNALS[tag]()
    eng = PlotEngine()
    captured = {}
    real = eng.ax_spec.pcolormesh

    def spy(t, f, img, **kw):
        captured["img"] = np.array(img, copy=True)
        return real(t, f, img, **kw)
    eng.ax_spec.pcolormesh = spy
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        if gmax is None:
            eng.plot_extra(x, None, fs, settings)
        else:
            eng.last_fs, eng.last_settings, eng.spec_data_source = fs, settings, x
            eng._plot_spectrogram(x, fs, settings, gmax)
        tf, feats = eng._calculate_features(x, fs, settings)
    assert (sum("nperseg" in str(w.message) for w in wl) > 0) == (int(g[f"{tag}__n_warnings"]) > 0)
    np.testing.assert_array_equal(eng.last_f, g[f"{tag}__last_f"])            # bit-exact
    np.testing.assert_array_equal(eng.last_t, g[f"{tag}__last_t"])            # bit-exact
    ref = g[f"{tag}__last_Sxx"]
    assert eng.last_Sxx.shape == ref.shape and eng.last_Sxx.dtype == ref.dtype
    f64 = ref.dtype == np.float64
    if ref.size and tag not in ("zeros", "const"):
        if f64:
            assert np.abs(eng.last_Sxx - ref).max() <= 1e-11 * np.abs(ref).max()
        else:
            assert_spec_close(eng.last_Sxx, ref, time_axis=-1)
    elif ref.size:
        assert np.abs(eng.last_Sxx - ref).max() <= 1e-12
    ref_img = g[f"{tag}__image"]
    if ref_img.size == 0:
        assert "img" not in captured
    else:
        assert captured["img"].shape == ref_img.shape
        assert np.abs(captured["img"] - ref_img).max() <= (1e-9 if f64 else 2e-4)
    ref_feats = g[f"{tag}__feats"]
    if ref_feats.shape[0] == 0:
        assert feats is None
    else:
        np.testing.assert_array_equal(tf, g[f"{tag}__feat_t"])
        assert feats.shape == ref_feats.shape and feats.dtype == ref_feats.dtype
        if tag in ("zeros", "const"):
            # an all-zero / constant (detrended to zero) signal: band power 0 -> log10(0 + 1e-20) = -20, difference 0
            expect = np.column_stack([np.full(len(feats), -20.0), np.zeros(len(feats))])
            assert np.allclose(feats, expect, rtol=0, atol=1e-12)
            assert np.allclose(ref_feats, expect, rtol=0, atol=1e-9)
        else:
            assert np.allclose(feats, ref_feats, rtol=0, atol=1e-9 if f64 else 2e-5)
    ap = eng.calculate_absolute_power()
    assert np.isclose(ap, g[f"{tag}__abs_power"], rtol=1e-10 if f64 else 1e-5, atol=1e-12)
    bp = eng.calculate_band_powers()
    assert list(bp.keys()) == [str(s) for s in g[f"{tag}__band_names"]]
    assert np.allclose([float(v) for v in bp.values()], g[f"{tag}__band_values"], rtol=1e-9 if f64 else 1e-5, atol=1e-12)


def test_plot_sweeps_combine_and_export_surface():
    """plot_sweeps with combine=True: segment_map, last_raw_t, combined_* and the spectrogram of the concatenation."""
    from PlotEngine import PlotEngine
    rng = np.random.default_rng(4)
    a, b = rng.standard_normal(3000), rng.standard_normal(2000)
    items = ["itemA", "itemB"]
    infos = [{"item": items[0], "signal_raw": a, "signal_proc": None, "fs": 500.0},
             {"item": items[1], "signal_raw": b, "signal_proc": None, "fs": 500.0}]
    settings = {"combine": True, "draw_raw": True, "draw_proc": False, "mode_raw": "Both", "mode_proc": "None",
                "nperseg": 256, "fmin": 0.0, "fmax": 100.0, "log_scale": True}
    eng = PlotEngine()
    eng.plot_sweeps(infos, settings)
    assert eng.currently_plotted_items == items
    assert [m["source_item"] for m in eng.segment_map] == items
    assert eng.segment_map[0]["end_time_combined"] == 6.0 and eng.segment_map[1]["end_time_combined"] == 10.0
    joined = np.concatenate([a, b])
    np.testing.assert_array_equal(eng.combined_raw, joined)
    np.testing.assert_array_equal(eng.last_raw_t, np.arange(5000) / 500.0)
    f, t, s = orc.spectrogram(joined, fs=500.0, nperseg=256)
    lf, lt, ls, img = orc.plot_image(f, t, s, 0.0, 100.0, True)
    np.testing.assert_array_equal(eng.last_f, lf)
    np.testing.assert_array_equal(eng.last_t, lt)
    assert np.abs(eng.last_Sxx - ls).max() <= 1e-11 * ls.max()
    assert eng.spec_data_source is not None and eng.last_fs == 500.0


def test_device_epilogues_vs_oracle_f32_batch():
    from spectro import engine
    rng = np.random.default_rng(12)
    x = (rng.standard_normal((3, 20000)) * 0.2).astype(np.float32)
    dev = engine.stft(x, fs=8000.0, nperseg=1024)
    f, t, s = orc.spectrogram(x, fs=8000.0, nperseg=1024)
    assert_spec_close(dev.to_host(), s, time_axis=-1)
    k_lo, k_hi = engine.bin_range(f, 100.0, 2500.0)
    sl = dev.band_slice(k_lo, k_hi)
    assert_spec_close(sl, s[:, k_lo:k_hi + 1, :], time_axis=-1)
    lo, hi = dev.minmax(k_lo, k_hi)
    assert lo == sl.min() and hi == sl.max()
    # batch-global normalisation (one base for all clips)
    for log in (False, True):
        img = dev.image(k_lo, k_hi, log)
        _, _, _, ref = orc.plot_image(f, t, np.concatenate(list(s), axis=1), 100.0, 2500.0, log)
        got = np.concatenate(list(img), axis=1)
        assert np.abs(got - ref).max() <= 2e-4
    feats = dev.features(k_lo, k_hi)
    t2, fused = engine.band_features(x, 8000.0, 1024, 100.0, 2500.0)
    for c in range(3):
        _, ref = orc.hmm_features(x[c], 8000.0, 1024, 100.0, 2500.0)
        assert np.allclose(feats[c], ref, atol=2e-5) and np.allclose(fused[c], ref, atol=2e-5)
    tot = dev.band_totals([(0, 513), (10, 20), (20, 20), (500, 513)])
    sd = s.astype(np.float64)
    assert np.allclose(tot, [sd.sum(), sd[:, 10:20].sum(), 0.0, sd[:, 500:513].sum()], rtol=1e-5)
    dev.free()


def test_mel_epilogue_vs_own_oracle():
    """cfg3: 80-band mel of the PSD.  No reference behaviour exists (parity unpinned): checked against
    oracle/mel_oracle.py, the numpy restatement of this library's own definition.  Asymmetric data so that a
    transposed MFMA operand or accumulator map cannot pass."""
    from oracle import mel_oracle
    from spectro import engine
    from spectro.mel import MelBank
    rng = np.random.default_rng(21)
    x = (rng.standard_normal((2, 30000)) * np.linspace(0.05, 1.0, 30000)).astype(np.float32)
    dev = engine.stft(x, fs=48000.0, nperseg=1024, window="hann", noverlap=768)
    bank = MelBank(1024, 48000.0, 80, 20.0, 20000.0)
    w = mel_oracle.mel_weights(1024, 48000.0, 80, 20.0, 20000.0)
    assert np.abs(bank.weights - w).max() < 1e-12
    assert all(lo % 4 == 0 and hi % 4 == 0 and lo < hi for lo, hi in bank.tile_ranges)
    s = np.moveaxis(dev.to_host(), -1, -2)                    # [clip, frame, bin]
    for log in (False, True):
        ref = np.moveaxis(mel_oracle.mel_spectrogram(s, w.astype(np.float32), log), -1, -2)
        for dense, kernel in ((False, None), (False, "mfma"), (True, None)):      # band-sparse gather (default), block-sparse MFMA, dense MFMA
            got = bank.apply(dev, log_scale=log, dense=dense, kernel=kernel)
            assert got.shape == ref.shape == (2, 80, dev.n_frames)
            if log:
                assert np.abs(got - ref).max() < 1e-3          # dB
            else:
                assert np.abs(got - ref).max() <= 2e-6 * ref.max() and np.allclose(got, ref, rtol=2e-5, atol=1e-6 * ref.max())
    # odd sizes: 37 mel bands on a 513-bin spectrum with 1 clip / frames not a multiple of 16
    dev2 = engine.stft(x[0, :9000], fs=48000.0, nperseg=1024, window="hann", noverlap=512)
    bank2 = MelBank(1024, 48000.0, 37)
    ref2 = (np.moveaxis(dev2.to_host(), -1, -2).astype(np.float64) @ bank2.weights.astype(np.float32)).T
    for kernel in (None, "mfma"):
        got2 = bank2.apply(dev2, kernel=kernel)
        assert got2.shape == ref2.shape and np.allclose(got2, ref2, rtol=2e-5, atol=1e-6 * ref2.max())
    dev.free(); dev2.free(); bank.close(); bank2.close()


@pytest.mark.parametrize("transport", ["auto", "host", "device"])
@pytest.mark.parametrize("nperseg,hop,n_ch,fs", [(4096, 1024, 8, 96000.0), (1024, 256, 3, 48000.0), (256, 224, 2, 500.0)])
def test_streaming_equals_offline(nperseg, hop, n_ch, fs, transport):
    """cfg5: feeding chunks (4096 samples/channel, plus ragged sizes) yields exactly the offline frames -- with the staging rows in
    HBM (one H2D, one launch, one D2H per chunk) and in pinned host memory (small chunks: the kernel crosses PCIe itself)."""
    import spectro
    from spectro.stream import StreamingSTFT
    rng = np.random.default_rng(nperseg)
    total = nperseg * 6 + 777
    x = (rng.standard_normal((n_ch, total)) * 0.3).astype(np.float32)
    st = StreamingSTFT(n_ch, fs, nperseg, hop, window="hann", transport=transport)
    chunks = [4096, 1, 4096, 313, 0, 4096, 5000, 9000]
    pos, ts, outs = 0, [], []
    while pos < total:
        n = min(chunks[len(ts) % len(chunks)], total - pos)
        t, s = st.feed(x[:, pos:pos + n])
        ts.append(t); outs.append(s)
        pos += n
    t_all, s_all = np.concatenate(ts), np.concatenate(outs, axis=-1)
    f, t_ref, s_ref = spectro.spectrogram(x, fs=fs, nperseg=nperseg, window="hann", noverlap=nperseg - hop)
    assert s_all.shape == s_ref.shape
    np.testing.assert_array_equal(t_all, t_ref)
    assert_spec_close(s_all, s_ref, time_axis=-1)             # same samples; the kernel variant may differ (alignment)
    assert np.abs(s_all - s_ref).max() <= 2e-6 * s_ref.max()
    # parity proper: the streamed frames against the ORACLE's offline spectrogram of the same samples
    f_o, t_o, s_o = orc.spectrogram(x, fs=fs, nperseg=nperseg, window="hann", noverlap=nperseg - hop)
    np.testing.assert_array_equal(t_all, t_o)
    assert_spec_close(s_all, s_o, time_axis=-1)
    assert st.transport == ("host" if transport == "auto" else transport)          # these chunk sizes are zero-copy sized
    st.close()


@pytest.mark.parametrize("transport", ["host", "device"])
@pytest.mark.parametrize("nperseg,hop,max_chunk", [(256, 64, 300), (256, 37, 256), (1024, 256, 1000), (100, 100, 64)])
def test_streaming_many_small_chunks(nperseg, hop, max_chunk, transport):
    """The staging rows only advance an offset per chunk and move the tail back to the front when they run out of room
    (every ~32 chunks): hundreds of small ragged chunks, even and odd hops, still equal the offline call."""
    import spectro
    from spectro.stream import StreamingSTFT
    rng = np.random.default_rng(hop)
    total = 60000
    x = (rng.standard_normal((2, total)) * 0.3 + 0.1).astype(np.float32)
    st = StreamingSTFT(2, 8000.0, nperseg, hop, window="hann", max_chunk=max_chunk, transport=transport)
    pos, ts, outs = 0, [], []
    while pos < total:
        n = min(int(rng.integers(0, max_chunk + 1)), total - pos)
        t, s = st.feed(x[:, pos:pos + n])
        ts.append(t); outs.append(s)
        pos += n
    t_all, s_all = np.concatenate(ts), np.concatenate(outs, axis=-1)
    f, t_ref, s_ref = spectro.spectrogram(x, fs=8000.0, nperseg=nperseg, window="hann", noverlap=nperseg - hop)
    assert s_all.shape == s_ref.shape and len(ts) > 100
    np.testing.assert_array_equal(t_all, t_ref)
    assert_spec_close(s_all, s_ref, time_axis=-1)
    f_o, t_o, s_o = orc.spectrogram(x, fs=8000.0, nperseg=nperseg, window="hann", noverlap=nperseg - hop)     # vs the oracle
    np.testing.assert_array_equal(t_all, t_o)
    assert_spec_close(s_all, s_o, time_axis=-1)
    st.close()


@pytest.mark.parametrize("variant", ["default", "mfma", "ws", "ws_cons4"])
@pytest.mark.parametrize("hop,n_mels,detrend", [(256, 80, "constant"), (896, 40, "constant"), (130, 128, False)])
def test_fused_stft_mel(hop, n_mels, detrend, variant, monkeypatch):
    """cfg3 fused kernel (sg_stft_mel): equals mel(oracle PSD) and the unfused sg_stft + sg_mel path, incl. a frame
    count that is not a multiple of the 16-frame tile and several clips; every kernel form the library carries."""
    # default = the band-sparse epilogue of the register kernel; the three MFMA tile kernels stay selectable
    for k, v in {"default": {}, "mfma": {"SPECTRO_FUSED_MFMA": "1"}, "ws": {"SPECTRO_FUSED_MFMA": "1", "SPECTRO_FUSED_WS": "1"},
                 "ws_cons4": {"SPECTRO_FUSED_MFMA": "1", "SPECTRO_FUSED_WS": "1", "SPECTRO_FUSED_CONS": "4"}}[variant].items():
        monkeypatch.setenv(k, v)
    from oracle import mel_oracle
    from spectro import _capi, engine
    from spectro.mel import MelBank
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(hop)
    N = 1024 + hop * 53 + 11
    x = (rng.standard_normal((3, N)) * np.linspace(0.1, 1.0, N) + 0.3).astype(np.float32)
    if N % 2:
        x = np.ascontiguousarray(x[:, :-1])
    bank = MelBank(1024, 48000.0, n_mels, 30.0, 22000.0)
    assert bank._sparse is not None and 1 <= bank._sparse[0] <= 4                               # work items per lane
    plan = plan_for(get_window("hann", 1024), 1024, 1024, hop, _capi.DETREND[detrend], 48000.0, 0, 0, _capi.F32)
    assert plan.kernel == "r8x3"
    _, _, s = orc.spectrogram(x, fs=48000.0, nperseg=1024, window="hann", noverlap=1024 - hop, detrend=detrend)
    w32 = bank.weights.astype(np.float32)
    for log in (False, True):
        got = bank.stft_mel(x, plan, log_scale=log)
        ref = np.moveaxis(mel_oracle.mel_spectrogram(np.moveaxis(s, -1, -2), w32, log), -1, -2)
        assert got.shape == ref.shape == (3, n_mels, s.shape[-1])
        if log:
            assert np.abs(got - ref).max() < 2e-3
        else:
            assert np.abs(got - ref).max() <= 1e-4 * ref.max(axis=1, keepdims=True).max()
            assert np.allclose(got, ref, rtol=1e-4, atol=1e-5 * ref.max())
    dev = engine.stft(x, fs=48000.0, nperseg=1024, window="hann", noverlap=1024 - hop, detrend=detrend)
    unfused = bank.apply(dev)
    fused = bank.stft_mel(x, plan)
    assert np.allclose(fused, unfused, rtol=2e-5, atol=1e-6 * unfused.max())
    dev.free(); bank.close()


def test_device_colormap_matches_matplotlib_jet():
    """N3 display epilogue: RGBA bytes of the normalised image equal matplotlib's own jet mapping of the same image."""
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import colormaps
    from spectro import engine
    rng = np.random.default_rng(31)
    x = (rng.standard_normal(30000) * 0.2).astype(np.float32)
    dev = engine.stft(x, fs=8000.0, nperseg=512)
    k_lo, k_hi = engine.bin_range(dev.f, 50.0, 3000.0)
    for log in (False, True):
        img = dev.image(k_lo, k_hi, log)
        rgba = dev.image_rgba(k_lo, k_hi, log)
        ref = colormaps["jet"](img, bytes=True)
        assert rgba.shape == ref.shape == img.shape + (4,)
        diff = np.abs(rgba.astype(int) - ref.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3        # LUT identical up to float rounding at bin edges
    dev.free()


def test_sharded_sweep_single_rank_on_device():
    """cfg4 API on one rank: every (clip, n_fft, hop) item equals the offline band log-power of the same samples."""
    from spectro import sweep
    rng = np.random.default_rng(17)
    clips = (rng.standard_normal((2, 30000)) * 0.2).astype(np.float32)
    res = sweep.sharded_sweep(clips, 48000.0, [256, 512, 1024, 2048, 4096], [64, 256], fmin=200.0, fmax=8000.0)
    assert len(res) == 2 * 5 * 2
    for (clip, n, h), v in res.items():
        f, t, s = orc.spectrogram(clips[clip], fs=48000.0, nperseg=n, window="hann", noverlap=n - h)
        m = (f >= 200.0) & (f <= 8000.0)
        ref = np.log10(s[m].sum(axis=0) + 1e-20)
        assert v.shape == ref.shape and np.allclose(v, ref, atol=2e-5), (clip, n, h)
    # the batched deal (one upload, one call per pair) and the per-item deal give the same numbers
    per_item = sweep.sharded_sweep(clips, 48000.0, [256, 1024, 4096], [64, 256], fmin=200.0, fmax=8000.0, batched=False)
    for k, v in per_item.items():
        assert np.array_equal(v, res[k]), k
    # hop sharing (64 and 256 of one n_fft come from ONE hop-64 transform) changes no bit
    separate = sweep.sharded_sweep(clips, 48000.0, [256, 512, 1024, 2048, 4096], [64, 256], fmin=200.0, fmax=8000.0, share_hops=False)
    assert separate.keys() == res.keys()
    for k, v in separate.items():
        assert np.array_equal(v, res[k]), k


def test_sharded_sweep_products_left_on_the_device():
    """What the RCCL gather sends: the reduced products stay in HBM as torch tensors over the library's blocks (zero-copy through
    __cuda_array_interface__) until the gather has moved them.  One rank: same numbers, bit for bit, as the host-array path -- and
    torch is imported AFTER the engine here, which needs the two to share one HIP runtime."""
    from spectro import sweep
    rng = np.random.default_rng(23)
    clips = (rng.standard_normal((3, 20000)) * 0.2).astype(np.float32)
    args = (clips, 8000.0, [256, 1024, 2048], [64, 128, 256])
    host = sweep.sharded_sweep(*args, fmin=100.0, fmax=3000.0)
    for kw in (dict(), dict(share_hops=False)):
        dev = sweep.sharded_sweep(*args, fmin=100.0, fmax=3000.0, device_products=True, **kw)
        assert dev.keys() == host.keys() and len(dev) == 3 * 9
        for k, v in dev.items():
            assert isinstance(v, np.ndarray) and np.array_equal(v, host[k]), k
    # a clip too short for the largest n_fft (nperseg clamps, one frame) and an empty band, still on the device
    short = (rng.standard_normal((2, 700)) * 0.2).astype(np.float32)
    a = sweep.sharded_sweep(short, 8000.0, [256, 1024], [64], fmin=100.0, fmax=3000.0)
    b = sweep.sharded_sweep(short, 8000.0, [256, 1024], [64], fmin=100.0, fmax=3000.0, device_products=True)
    assert all(np.array_equal(a[k], b[k]) for k in a) and {v.shape for k, v in b.items() if k[1] == 1024} == {(1,)}
    e = sweep.sharded_sweep(short, 8000.0, [256], [64], fmin=3999.0, fmax=3999.5, device_products=True)
    assert all(np.all(v == np.log10(np.float32(1e-20))) for v in e.values())


def test_device_clips_products_vs_oracle():
    """DeviceClips: clips uploaded once; band features (whole batch and a clip range), log display through the fused kernel
    (nperseg 1024) and through the composed path (nperseg 512), int16 clips."""
    from spectro import engine
    rng = np.random.default_rng(33)
    x = (rng.standard_normal((5, 20000)) * 0.2).astype(np.float32)
    dc = engine.DeviceClips(x)
    try:
        for n, h in ((1024, 256), (512, 128), (1000, 250)):
            t, feats = dc.band_log_power(16000.0, n, h, 100.0, 3000.0, window="hann")
            for c in range(5):
                fo, to, so = orc.spectrogram(x[c], fs=16000.0, nperseg=n, window="hann", noverlap=n - h)
                m = (fo >= 100.0) & (fo <= 3000.0)
                lp = np.log10(so[m].sum(axis=0) + 1e-20)
                np.testing.assert_array_equal(t, to)
                assert np.allclose(feats[c, :, 0], lp, atol=2e-5)
                assert np.allclose(feats[c, :, 1], np.diff(lp, prepend=lp[0]), atol=4e-5)
            t2, part = dc.band_log_power(16000.0, n, h, 100.0, 3000.0, window="hann", clip_range=(1, 4))
            assert np.array_equal(part, feats[1:4])
        for n, h in ((1024, 256), (512, 128)):
            fo, to, so = orc.spectrogram(x, fs=16000.0, nperseg=n, window="hann", noverlap=n - h)
            gmax = float(so.max()) * 1e8                       # well-conditioned display minimum (tests/test_gpu_db.py)
            fb, tt, img = dc.log_image(16000.0, n, h, 100.0, 3000.0, gmax, window="hann")
            m = (fo >= 100.0) & (fo <= 3000.0)
            band = so[:, m, :]
            db = 10.0 * np.log10(np.clip(band / (np.float32(gmax) + np.float32(1e-20)), 0, 1) + np.float32(1e-12))
            ref = (db - db.min()) / (db.max() - db.min())      # min-max over the WHOLE batch (one global_max, one image)
            np.testing.assert_array_equal(fb, fo[m])
            assert img.shape == ref.shape and np.abs(img - ref).max() <= 2e-3
            strong = band >= 1e-3 * so.max(axis=1, keepdims=True)
            assert np.abs(img[strong] - ref[strong]).max() <= 1e-4
    finally:
        dc.free()
    xi = (x * 20000).astype(np.int16)
    di = engine.DeviceClips(xi)
    try:
        dev = di.stft(fs=16000.0, window="hann", nperseg=1024, hop=256)
        fo, to, so = orc.spectrogram(xi, fs=16000.0, nperseg=1024, window="hann", noverlap=768)
        assert_spec_close(dev.to_host(), so, time_axis=-1)
        dev.free()
        t, feats = di.band_log_power(16000.0, 1024, 256, 0.0, 8000.0, window="hann")
        assert np.allclose(feats[..., 0], np.log10(so.sum(axis=1) + 1e-20), atol=2e-5)
    finally:
        di.free()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_epilogue_abi_random_shapes(dtype):
    """A8-A13 through the C ABI on ragged shapes: every bin (the contiguous 16 B/lane walk, also from a pointer
    that is not 16 B aligned), sub-bands, one-bin bands, overlapping / empty / clipped band tables."""
    import ctypes as C
    from spectro import _capi
    _capi.ensure_device()
    L = _capi.lib()
    code = 0 if dtype == np.float32 else 1
    rng = np.random.default_rng(77)
    isz = np.dtype(dtype).itemsize
    for rows, nb, off in [(1, 1, 0), (7, 65, 0), (1000, 129, 0), (333, 513, 0), (333, 513, 1), (40, 2049, 0), (5, 5000, 1)]:
        host = (rng.random((rows, nb)) ** 8).astype(dtype) + dtype(1e-6)
        host[rng.integers(rows), rng.integers(nb)] = 3.0
        buf = _capi.DeviceBuffer((rows * nb + 4) * isz)
        flat = np.zeros(rows * nb + 4, dtype)
        flat[off:off + rows * nb] = host.ravel()
        buf.upload(flat)
        spec = C.c_void_p(buf.ptr + off * isz)
        img = _capi.DeviceBuffer(rows * nb * isz)
        mm = _capi.DeviceBuffer(64)
        sums = _capi.DeviceBuffer(16 * 8)
        bands = [(0, nb - 1), (nb // 3, nb // 3), (nb // 4, (3 * nb) // 4), (nb - 1, nb - 1)]
        for k_lo, k_hi in bands:
            w = k_hi - k_lo + 1
            ref = host[:, k_lo:k_hi + 1]
            _capi.check(L.sg_minmax(spec, code, rows, nb, k_lo, k_hi, C.c_void_p(mm.ptr), None))
            got = mm.download(np.zeros(2, dtype)); _capi.stream_sync()
            assert got[0] == ref.min() and got[1] == ref.max()
            bsum = np.zeros(rows, dtype)
            _capi.check(L.sg_band_sum(spec, code, rows, nb, k_lo, k_hi, C.c_void_p(img.ptr), None))
            img.download(bsum); _capi.stream_sync()
            assert np.allclose(bsum, ref.astype(np.float64).sum(axis=1), rtol=2e-6 if dtype == np.float32 else 1e-13, atol=0)
            out = np.zeros((rows, w), dtype)
            _capi.check(L.sg_slice_bins(spec, code, rows, nb, k_lo, k_hi, C.c_void_p(img.ptr), None))
            img.download(out); _capi.stream_sync()
            np.testing.assert_array_equal(out, ref)
            for log, gmax in [(0, 0.0), (1, 0.0), (1, 7.5)]:
                _capi.check(L.sg_normalise_image(spec, code, rows, nb, k_lo, k_hi, log, gmax, C.c_void_p(img.ptr),
                                                 C.c_void_p(mm.ptr), None))
                img.download(out); _capi.stream_sync()
                want = orc.plot_image(np.arange(w, dtype=float), np.arange(rows, dtype=float), ref.T.astype(np.float64),
                                      -1.0, w + 1.0, bool(log), gmax if gmax > 0 else None)[3].T
                assert np.abs(out - want).max() <= (2e-4 if dtype == np.float32 else 1e-9)
        table = [(0, nb), (0, 1), (nb // 3, nb // 2), (nb // 4, nb), (5, 5), (-3, 2), (nb - 1, nb + 9), (nb // 2, nb // 2 + 1)]
        lo = (C.c_int * len(table))(*[a for a, _ in table]); hi = (C.c_int * len(table))(*[b for _, b in table])
        _capi.check(L.sg_band_totals(spec, code, rows, nb, len(table), lo, hi, C.c_void_p(sums.ptr), None))
        got = sums.download(np.zeros(len(table))); _capi.stream_sync()
        h64 = host.astype(np.float64)
        want = [h64[:, max(a, 0):min(b, nb)].sum() if min(b, nb) > max(a, 0) else 0.0 for a, b in table]
        assert np.allclose(got, want, rtol=1e-6 if dtype == np.float32 else 1e-13, atol=0)
        for b_ in (buf, img, mm, sums):
            b_.free()


@pytest.mark.parametrize("nfft,n_mels,rows", [(1024, 80, 70003), (1024, 128, 1000), (256, 8, 333), (512, 40, 4099),
                                              (4096, 64, 257), (1000, 23, 100)])
def test_mel_abi_shapes(nfft, n_mels, rows):
    """sg_mel on synthetic spectra: more 16-frame tiles than persistent waves (70 003 rows), 1 and 8 mel tiles, banks whose
    weights fit LDS and banks that do not (nfft 4096), ragged row counts, odd nfft -- against a float64 product."""
    from spectro import _capi
    from spectro.mel import MelBank
    _capi.ensure_device()
    rng = np.random.default_rng(nfft + n_mels)
    nb = nfft // 2 + 1
    spec = (rng.random((rows, nb), dtype=np.float32) ** 4) * np.linspace(2.0, 0.01, nb, dtype=np.float32)
    bank = MelBank(nfft, 48000.0, n_mels)
    d_in, d_out = _capi.DeviceBuffer(spec.nbytes), _capi.DeviceBuffer(rows * n_mels * 4)
    d_in.upload(spec)
    ref = spec.astype(np.float64) @ bank.weights.astype(np.float32).astype(np.float64)
    variants = [(False, "mfma"), (True, None)] + ([(False, "sparse")] if bank._sparse is not None else [])
    for dense, kernel in variants:
        got = np.zeros((rows, n_mels), np.float32)
        bank.apply_ptr(d_in.ptr, rows, d_out.ptr, False, dense, kernel=kernel)
        d_out.download(got); _capi.stream_sync()
        assert np.allclose(got, ref, rtol=2e-5, atol=2e-6 * ref.max()), (dense, kernel)
    d_in.free(); d_out.free(); bank.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_plotengine_fast_image_equals_pcolormesh_colours(dtype):
    """settings['fast_image']: one imshow of the device-made RGBA image instead of a quad mesh -- same colours as
    matplotlib's jet on the reference's normalised image, cell edges where pcolormesh(shading='auto') puts them."""
    import time
    from matplotlib import cm
    from PlotEngine import PlotEngine
    x = cfg1_signal().astype(dtype)
    settings = {"nperseg": 256, "fmin": 0.0, "fmax": 4000.0, "log_scale": True, "mode_raw": "Spectrogram",
                "mode_proc": "None", "draw_raw": False, "draw_proc": False}
    slow, fast = PlotEngine(), PlotEngine()
    t0 = time.perf_counter(); slow.plot_extra(x, None, 16000.0, settings); t_slow = time.perf_counter() - t0
    t0 = time.perf_counter(); fast.plot_extra(x, None, 16000.0, dict(settings, fast_image=True)); t_fast = time.perf_counter() - t0
    np.testing.assert_array_equal(fast.last_Sxx, slow.last_Sxx)
    assert len(fast.ax_spec.images) == 1 and not fast.ax_spec.collections
    im = fast.ax_spec.images[0]
    rgba = np.asarray(im.get_array())
    f64 = slow.last_Sxx.dtype == np.float64
    _, _, _, ref_img = orc.plot_image(slow.last_f, slow.last_t, slow.last_Sxx.astype(np.float64), 0.0, 4000.0, True)
    want = cm.jet(ref_img, bytes=True)
    assert rgba.shape == want.shape
    # an image value that sits on a colour-table boundary may fall either side in f32: allow a step of the 256-entry table
    assert (np.abs(rgba.astype(int) - want.astype(int)).max(axis=-1) <= (0 if f64 else 6)).mean() > 0.999
    t, f = slow.last_t, slow.last_f
    ext = im.get_extent()
    assert np.allclose(ext, (t[0] - (t[1] - t[0]) / 2, t[-1] + (t[1] - t[0]) / 2, f[0] - (f[1] - f[0]) / 2, f[-1] + (f[1] - f[0]) / 2))
    assert fast.ax_spec.get_xlim() == slow.ax_spec.get_xlim() and fast.ax_spec.get_ylim() == slow.ax_spec.get_ylim()
    print(f"plot_extra: pcolormesh {t_slow*1e3:.1f} ms, imshow {t_fast*1e3:.1f} ms")


@pytest.mark.parametrize("dtype,mode", [(np.float32, "psd"), (np.int16, "psd"), (np.float64, "psd"), (np.float32, "complex")])
def test_pipelined_ingest_equals_unpipelined_and_oracle(dtype, mode):
    """N2: chunked double-buffered transfers into pinned memory give the unpipelined result bit for bit (several chunk
    sizes incl. a ragged last chunk and the single-chunk case) and match the oracle on sampled frames."""
    import spectro
    from spectro.pipeline import stft_pipelined, pinned_pool_clear
    rng = np.random.default_rng(21)
    x = rng.standard_normal((7, 30000)) * 0.3
    x = (x * 8000).astype(np.int16) if dtype == np.int16 else x.astype(dtype)
    kw = dict(fs=16000.0, nperseg=1024 if dtype != np.float64 else 512, window="hann", noverlap=768 if dtype != np.float64 else 384, mode=mode)
    f0, t0, s0 = spectro.spectrogram(x, **kw)
    for chunk in (2 * x[0].nbytes, 3 * x[0].nbytes + 5, 1 << 30, 1):
        f1, t1, s1 = stft_pipelined(x, chunk_bytes=chunk, **kw)
        np.testing.assert_array_equal(f1, f0)
        np.testing.assert_array_equal(t1, t0)
        assert s1.dtype == s0.dtype and s1.shape == s0.shape
        np.testing.assert_array_equal(s1, s0)
    fo, to, so = orc.spectrogram(x, **kw)
    assert_spec_close(s1, so, time_axis=-1) if dtype != np.float64 else np.testing.assert_allclose(s1, so, rtol=0, atol=1e-11 * so.max())
    # the result is an ordinary numpy array that outlives the call and can be written to
    keep = s1.copy()
    del s1, f1, t1
    stft_pipelined(x, **kw)
    np.testing.assert_array_equal(keep, s0)
    pinned_pool_clear()
    from spectro.pipeline import workspace_release
    workspace_release()


def test_sweepmanager_batch_pipeline_flag():
    from SweepManager import SweepManager
    sm = SweepManager()
    rng = np.random.default_rng(2)
    for i in range(5):
        sm.add_signal(f"rec_sweep{i}", (rng.standard_normal(20000) * 0.1).astype(np.float32), 8000.0)
    names = list(sm.data)
    a = sm.spectrogram_batch(names, 256, window="hann", noverlap=192)
    b = sm.spectrogram_batch(names, 256, window="hann", noverlap=192, pipeline=True)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    sweep = sm.parameter_sweep(names, [256, 1024], [64, 256])
    for (n, h), (f, t, s) in sweep.items():
        fo, to, so = orc.spectrogram(np.stack([sm.data[k]["raw"] for k in names]), fs=8000.0, nperseg=n, window="hann", noverlap=n - h)
        np.testing.assert_array_equal(t, to)
        assert_spec_close(s, so, time_axis=-1)
    # hop 256 came back as a strided view of the hop-64 download; the unshared sweep gives the same bits and owns its arrays
    assert sweep[(256, 256)][2].base is not None and np.shares_memory(sweep[(256, 256)][2], sweep[(256, 64)][2])
    plain = sm.parameter_sweep(names, [256, 1024], [64, 256, 96], share_hops=False)
    shared = sm.parameter_sweep(names, [256, 1024], [64, 256, 96])
    assert list(shared) == list(plain)
    for k in plain:
        for u, v in zip(plain[k], shared[k]):
            np.testing.assert_array_equal(u, v)
    assert not np.shares_memory(plain[(256, 256)][2], plain[(256, 64)][2])
    red = sm.parameter_sweep(names, [256], [64, 128], reduce=lambda dev: dev.minmax(0, dev.n_bins - 1))
    assert set(red) == {(256, 64), (256, 128)}


def test_long_clip_cut_at_frame_boundaries_is_the_whole_call():
    """one long recording sharded over ranks by frames (halo read, no exchange): the ranks' shares concatenated
    along time are the single-process result bit for bit, t included."""
    import spectro
    from spectro.dist import long_clip_spectrogram
    rng = np.random.default_rng(99)
    for dt, kw in [(np.float32, dict(nperseg=1024, window="hann", noverlap=768)), (np.float64, dict(nperseg=1024)),
                   (np.float32, dict(nperseg=2048, window="hann", noverlap=1984)), (np.float64, dict(nperseg=500, noverlap=123)),
                   (np.float32, dict(nperseg=512, mode="magnitude", detrend=False))]:
        x = (rng.standard_normal(700001) * 0.2 + 0.1).astype(dt)
        f, t, s = spectro.spectrogram(x, fs=48000.0, **kw)
        for world in (3, 8):
            parts = [long_clip_spectrogram(x, fs=48000.0, world=world, rank=r, **kw) for r in range(world)]
            assert [p[3] for p in parts][0][0] == 0 and parts[-1][3][1] == t.size
            np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), t)
            np.testing.assert_array_equal(np.concatenate([p[2] for p in parts], axis=-1), s)
            for p in parts:
                np.testing.assert_array_equal(p[0], f)
    # more ranks than frames: the extra ranks return empty shares
    x = rng.standard_normal(1024 + 3 * 896).astype(np.float32)
    parts = [long_clip_spectrogram(x, fs=1000.0, nperseg=1024, world=6, rank=r) for r in range(6)]
    assert [p[2].shape[-1] for p in parts] == [1, 1, 1, 1, 0, 0]
    # 'phase' unwraps along FREQUENCY (scipy:1003), frame by frame: frame shards of it are the whole call's columns as well
    x = (rng.standard_normal(40000) * 0.2).astype(np.float64)
    f, t, s = spectro.spectrogram(x, fs=8000.0, nperseg=256, noverlap=128, mode="phase")
    parts = [long_clip_spectrogram(x, fs=8000.0, nperseg=256, noverlap=128, mode="phase", world=3, rank=r) for r in range(3)]
    np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), t)
    np.testing.assert_array_equal(np.concatenate([p[2] for p in parts], axis=-1), s)
