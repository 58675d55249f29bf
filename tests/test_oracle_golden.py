"""Pin the CPU oracle (oracle/stft_oracle.py) against the golden vectors.

The goldens come from scipy 1.15.3 called with the reference's argument set and
from the reference's own PlotEngine methods (tests/golden/make_golden.py).
f64 cases must agree to ~1e-13 relative (same algorithm, pocketfft both sides);
integer framing and the f/t vectors bit-exactly.
"""
import json
import warnings

import numpy as np
import pytest

from conftest import cfg1_signal, cfg2_clips, eeg_like, load_golden, sweep_clip, assert_spec_close
from oracle import stft_oracle as orc


def _rel(a, b):
    wide = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    a, b = np.asarray(a, wide), np.asarray(b, wide)
    if b.size == 0:
        return 0.0
    den = np.abs(b).max()
    return 0.0 if den == 0 else float(np.abs(a - b).max() / den)


G1_SIGNALS = {
    "cfg1_lin": lambda: cfg1_signal(), "cfg1_log": lambda: cfg1_signal(),
    "cfg1_band_log": lambda: cfg1_signal(), "cfg1_gmax": lambda: cfg1_signal(),
    "cfg1_f32": lambda: cfg1_signal().astype(np.float32),
    "eeg_default": eeg_like, "eeg_lin_256": eeg_like, "eeg_np2_1000": eeg_like,
    "short_clamp": lambda: cfg1_signal()[:300],
    "zeros": lambda: np.zeros(4096), "const": lambda: np.full(4096, 2.5),
    "empty_mask": lambda: cfg1_signal(),
}


@pytest.mark.parametrize("tag", sorted(G1_SIGNALS))
def test_g1_reference_engine(tag):
    g = load_golden("g1_reference_engine.npz")
    fs, nper, fmin, fmax, log, gmax = g[f"{tag}__args"]
    x = G1_SIGNALS[tag]()
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        f, t, sxx = orc.spectrogram(x, fs=fs, nperseg=int(nper), scaling="density", mode="psd")
    n_warn = sum("nperseg" in str(w.message) for w in wl)
    assert (n_warn > 0) == (int(g[f"{tag}__n_warnings"]) > 0)
    lf, lt, ls, img = orc.plot_image(f, t, sxx, fmin, fmax, bool(log), None if gmax < 0 else gmax)
    np.testing.assert_array_equal(lf, g[f"{tag}__last_f"])
    np.testing.assert_array_equal(lt, g[f"{tag}__last_t"])
    assert ls.shape == g[f"{tag}__last_Sxx"].shape
    assert ls.dtype == g[f"{tag}__last_Sxx"].dtype
    tol = 1e-12 if ls.dtype == np.float64 else 2e-6
    assert _rel(ls, g[f"{tag}__last_Sxx"]) <= tol
    ref_img = g[f"{tag}__image"]
    if img is None:
        assert ref_img.size == 0
    else:
        assert img.shape == ref_img.shape
        assert np.abs(img - ref_img).max() <= (1e-9 if ls.dtype == np.float64 else 1e-4)
    # A11 features
    tf, feats = orc.hmm_features(x, fs, int(nper), fmin, fmax)
    ref_feats = g[f"{tag}__feats"]
    if ls.size == 0 and feats is not None:
        # features do not depend on the mask being empty unless Sxx itself is empty
        pass
    if feats is None:
        assert ref_feats.shape[0] == 0
    else:
        np.testing.assert_array_equal(tf, g[f"{tag}__feat_t"])
        assert np.allclose(feats, ref_feats, rtol=1e-9 if ls.dtype == np.float64 else 1e-5, atol=1e-9 if ls.dtype == np.float64 else 2e-5)
    # A12 / A13
    assert np.isclose(orc.absolute_power(ls), g[f"{tag}__abs_power"], rtol=tol * 10, atol=0)
    bp = orc.band_powers(lf, ls)
    assert list(bp.keys()) == [str(s) for s in g[f"{tag}__band_names"]]
    assert np.allclose([float(v) for v in bp.values()], g[f"{tag}__band_values"], rtol=1e-6, atol=1e-12)


def test_g1_cfg1_survey_numbers():
    """SURVEY §8c observed values: 257x35, sum 1.1132337761310838, first feature row."""
    g = load_golden("g1_reference_engine.npz")
    s = g["cfg1_lin__last_Sxx"]
    assert s.shape == (257, 35) and s.dtype == np.float64
    assert abs(s.sum() - 1.1132337761310838) < 1e-12
    assert np.allclose(g["cfg1_lin__last_t"][:2], [0.016, 0.044])
    assert np.allclose(g["cfg1_lin__feats"][0], [-1.47674084, 0.0], atol=1e-7)


G2_KW = {
    "ref": dict(nperseg=512),
    "hann256": dict(nperseg=512, window="hann", noverlap=256),
    "hann256_nodetrend": dict(nperseg=512, window="hann", noverlap=256, detrend=False),
    "hann256_linear": dict(nperseg=512, window="hann", noverlap=256, detrend="linear"),
    "hann256_mag": dict(nperseg=512, window="hann", noverlap=256, mode="magnitude"),
    "hann256_spectrum": dict(nperseg=512, window="hann", noverlap=256, scaling="spectrum"),
    "hann256_complex": dict(nperseg=512, window="hann", noverlap=256, mode="complex"),
    "hann256_nfft1024": dict(nperseg=512, window="hann", noverlap=256, nfft=1024),
}


@pytest.mark.parametrize("tag", sorted(G2_KW))
@pytest.mark.parametrize("dt", ["float64", "float32"])
def test_g2_cfg1_extended(tag, dt):
    g = load_golden("g2_cfg1_extended.npz")
    x = cfg1_signal().astype(dt)
    f, t, s = orc.spectrogram(x, fs=16000.0, **{"scaling": "density", "mode": "psd", **G2_KW[tag]})
    key = f"{tag}_{dt}"
    np.testing.assert_array_equal(f, g[key + "__f"])
    np.testing.assert_array_equal(t, g[key + "__t"])
    ref = g[key + "__Sxx"]
    assert s.shape == ref.shape and s.dtype == ref.dtype
    tol = 1e-12 if dt == "float64" else 3e-6
    if tag == "hann256_linear":
        tol = max(tol, 1e-9)      # lstsq vs scipy's own solve
    assert _rel(s, ref) <= tol


@pytest.mark.parametrize("mode", ["angle", "phase"])
def test_g2_angle_and_phase_modes(mode):
    """scipy's 'angle' and 'phase' (= angle unwrapped ALONG FREQUENCY, scipy:990-992) on the f64 signal: the oracle must
    agree with scipy's stored result modulo 2*pi wherever the bin carries energy, and exactly on the frames whose unwrap has
    no close calls (no neighbouring wrapped angles within 1e-6 of a +-pi step)."""
    g = load_golden("g2_cfg1_extended.npz")
    x = cfg1_signal()
    kw = dict(nperseg=512, window="hann", noverlap=256)
    f, t, s = orc.spectrogram(x, fs=16000.0, mode=mode, **kw)
    ref = g[f"hann256_{mode}_float64__Sxx"]
    mag = g["hann256_mag_float64__Sxx"]
    ang = g["hann256_angle_float64__Sxx"]
    assert s.shape == ref.shape
    strong = mag > 1e-6 * mag.max()
    d = np.angle(np.exp(1j * (s - ref)))
    assert np.abs(d[strong]).max() < 1e-8
    if mode == "phase":
        solid = strong.all(axis=0) & (np.abs(np.abs(np.diff(ang, axis=0)) - np.pi) > 1e-6).all(axis=0)
        assert solid.sum() >= 3
        assert np.abs(s[:, solid] - ref[:, solid]).max() < 1e-8
        assert np.abs(np.diff(s, axis=0)).max() <= np.pi + 1e-9


@pytest.mark.parametrize("tag", ["ext", "ref"])
def test_g3_cfg2_sampled(tag):
    g = load_golden("g3_cfg2_sampled.npz")
    clips = cfg2_clips(2)
    kw = dict(window="hann", noverlap=768) if tag == "ext" else {}
    f, t, s = orc.spectrogram(clips, fs=48000.0, nperseg=1024, scaling="density", mode="psd", **kw)
    np.testing.assert_array_equal(f, g[f"{tag}__f"])
    np.testing.assert_array_equal(t, g[f"{tag}__t"])
    assert str(s.dtype) == str(g[f"{tag}__dtype"]) == "float32"
    assert s.shape == (2, 513, 1872 if tag == "ext" else 535)
    idx = g[f"{tag}__frame_idx"]
    got = np.moveaxis(s[:, :, idx], -1, 1)
    assert_spec_close(got, g[f"{tag}__frames"], time_axis=1)
    assert np.allclose(s.astype(np.float64).sum(axis=1), g[f"{tag}__frame_sums"], rtol=1e-5)


def test_g4_sweep():
    g = load_golden("g4_sweep.npz")
    x = sweep_clip()
    for n in (256, 512, 1024, 2048, 4096):
        for hop in (64, 128, 256):
            k = f"n{n}_h{hop}"
            f, t, s = orc.spectrogram(x, fs=48000.0, nperseg=n, window="hann", noverlap=n - hop)
            assert s.shape[-1] == int(g[k + "__nframes"]) == orc.frame_count(len(x), n, hop)
            np.testing.assert_array_equal([t[0], t[-1]], g[k + "__t_ends"])
            assert_spec_close(s[:, g[k + "__frame_idx"]].T, g[k + "__frames"], time_axis=0)


G5_TAGS = ["short", "exact", "exact_plus", "two", "odd33", "np2_1000", "np2_96", "int16", "zeros", "const",
           "dc_large", "n8192", "n32", "hop1", "default_nperseg", "batch2d"]


@pytest.mark.parametrize("tag", G5_TAGS)
def test_g5_edges(tag):
    g = load_golden("g5_edges.npz")
    kw = json.loads(str(g[tag + "__kw"]))
    fs = kw.pop("fs")
    x = g[tag + "__x"]
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        f, t, s = orc.spectrogram(x, fs=fs, **kw)
    assert int(any("nperseg" in str(w.message) for w in wl)) == int(g[tag + "__warned"])
    np.testing.assert_array_equal(f, g[tag + "__f"])
    np.testing.assert_array_equal(t, g[tag + "__t"])
    ref = g[tag + "__Sxx"]
    assert s.shape == ref.shape and s.dtype == ref.dtype
    if ref.dtype == np.float64:
        assert _rel(s, ref) <= 1e-12
    elif tag in ("const", "dc_large"):
        # pure rounding noise (const) / heavy cancellation (dc_large): absolute bound vs signal scale
        assert np.abs(s - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-12) + 1e-10
    else:
        assert_spec_close(s, ref, time_axis=-1)


def test_window_tables_match_scipy():
    import scipy.signal as ss
    for n in (32, 33, 96, 512, 1000, 1024, 4096):
        np.testing.assert_array_equal(orc.tukey_periodic(n, 0.25), ss.get_window(("tukey", 0.25), n))
        np.testing.assert_allclose(orc.hann_periodic(n), ss.get_window("hann", n), rtol=0, atol=1e-16)
    assert orc.tukey_periodic(1024)[0] == 0.0
    assert float((orc.tukey_periodic(1024) ** 2).sum()) == pytest.approx(864.0, abs=1e-9)


def test_frame_count_and_vectors_property():
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = int(rng.integers(2, 300))
        N = int(rng.integers(n, 3000))
        step = int(rng.integers(1, n + 1))
        fs = float(rng.choice([1.0, 100.0, 500.0, 16000.0, 44100.0, 48000.0]))
        view = np.lib.stride_tricks.sliding_window_view(np.empty(N), n)[::step]
        assert orc.frame_count(N, n, step) == view.shape[0] == len(orc.time_vector(N, n, step, fs))
        assert np.array_equal(orc.freq_vector(n, fs), np.fft.rfftfreq(n, 1 / fs))


def test_get_signal_rules():
    """A15 hand-written cases for SweepManager.get_signal (SweepManager.py:151-185)."""
    raw, proc = np.arange(3.0), np.arange(4.0)
    data = {
        "abf": {"fs_raw": 10.0, "fs": 10.0, "raw": raw, "processed": None},
        "h5": {"fs_raw": 20.0, "fs": 5.0, "raw": raw, "processed": proc},
        "nofsraw": {"fs": 7.0, "raw": raw, "processed": proc},
        "noraw": {"fs": 7.0, "raw": None, "processed": None},
        "nofs": {"raw": raw, "processed": proc},
    }
    assert orc.get_signal(data, "abf")[1] == 10.0
    s, fs = orc.get_signal(data, "abf", processed=True)        # falls back to raw + fs_raw
    assert s is raw and fs == 10.0
    s, fs = orc.get_signal(data, "h5", processed=True)
    assert s is proc and fs == 5.0
    s, fs = orc.get_signal(data, "h5", processed=False)
    assert s is raw and fs == 20.0
    assert orc.get_signal(data, "nofsraw")[1] == 7.0
    for name, p in (("missing", False), ("noraw", False), ("noraw", True), ("nofs", False), ("nofs", True)):
        with pytest.raises(KeyError):
            orc.get_signal(data, name, processed=p)
