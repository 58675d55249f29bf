#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the BUILD container only).

Two sources, both recorded inside every .npz (``meta`` json string):

(a) ``scipy.signal.spectrogram`` (scipy 1.15.3 / numpy 2.2.6) called with the
    reference's exact argument set (PlotEngine.py:113 / :232) and with the
    BASELINE "extended" argument set (Hann, explicit hop);
(b) the reference's own ``PlotEngine`` class imported from /root/reference
    with inert stand-ins for the GUI toolkits it needs at import time (PyQt5,
    the Qt matplotlib backend, hmmlearn -- none of which take part in the
    numerics); its ``plot_extra`` / ``_plot_spectrogram`` /
    ``_calculate_features`` / ``calculate_absolute_power`` /
    ``calculate_band_powers`` are then *called* and their outputs stored.

Only inputs' seeds and the produced numbers are stored -- no reference source.
The GPU box has no /root/reference: nothing in tests/, bench.py or smoke()
runs this script; they only read the .npz files it wrote.

Usage:  python tests/golden/make_golden.py [--no-reference]
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types
import warnings

import numpy as np
import scipy
import scipy.signal as ss

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"


def _meta(**kw):
    d = {"scipy": scipy.__version__, "numpy": np.__version__}
    d.update(kw)
    return json.dumps(d)


def _save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ---------------------------------------------------------------------------
# inputs (regenerated from seeds in the tests; sha256 stored for safety)
# ---------------------------------------------------------------------------
def cfg1_signal():
    """cfg1: 1 s @ 16 kHz, SURVEY §8c: default_rng(0).standard_normal(16000)."""
    return np.random.default_rng(0).standard_normal(16000)


def cfg2_clips(n_clips=2):
    """cfg2-shaped: default_rng(1234).standard_normal((64, 480000)) f32 * 0.1, first clips."""
    x = np.random.default_rng(1234).standard_normal((64, 480000)).astype(np.float32) * np.float32(0.1)
    return np.ascontiguousarray(x[:n_clips])


def sweep_clip():
    """1 s @ 48 kHz f32 clip for the (n_fft, hop) sweep."""
    return (np.random.default_rng(77).standard_normal(48000) * 0.1).astype(np.float32)


def eeg_like():
    """20 s @ 500 Hz 'recording': 1/f-ish noise + 10 Hz alpha burst, f64 (ephys-like)."""
    rng = np.random.default_rng(5)
    n = 10000
    t = np.arange(n) / 500.0
    x = np.cumsum(rng.standard_normal(n)) * 0.01 + 0.2 * rng.standard_normal(n)
    x += np.where((t > 5) & (t < 8), 1.5 * np.sin(2 * np.pi * 10 * t), 0.0)
    return x + 3.0   # DC offset exercises the detrend


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------------------------------------------------------------------
# (b) import the reference PlotEngine with inert GUI stand-ins
# ---------------------------------------------------------------------------
def import_reference_plotengine():
    sys.dont_write_bytecode = True
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib.backends.backend_agg import FigureCanvasAgg

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    qtw = mod("PyQt5.QtWidgets", QFileDialog=object)
    qtc = mod("PyQt5.QtCore", Qt=types.SimpleNamespace(UserRole=256))
    qtg = mod("PyQt5.QtGui", QCursor=object)
    mod("PyQt5", QtWidgets=qtw, QtCore=qtc, QtGui=qtg)
    mod("matplotlib.backends.backend_qt5agg", FigureCanvasQTAgg=FigureCanvasAgg)
    # hmmlearn is absent here: the deterministic stand-in of tests/hmm_standin.py takes its place on BOTH sides (the reference's
    # engine below, this repository's engine in the GPU tests), so the logic around the model is what gets compared
    sys.path.insert(0, os.path.dirname(HERE))
    import hmm_standin
    hmm_standin.install(sys.modules)
    sys.path.insert(0, REFERENCE)
    try:
        import PlotEngine as ref_mod          # noqa: N813
        import ExportManager as ref_export    # noqa: N813
    finally:
        sys.path.remove(REFERENCE)
    import_reference_plotengine.export_manager = ref_export.ExportManager
    return ref_mod.PlotEngine


class _Recorder:
    """Wraps ax.pcolormesh so the normalised image handed to it can be stored."""

    def __init__(self, ax):
        self.ax, self.orig, self.image = ax, ax.pcolormesh, None
        ax.pcolormesh = self

    def __call__(self, t, f, img, **kw):
        self.image = np.array(img, copy=True)
        return self.orig(t, f, img, **kw)


def reference_run(PlotEngine, x, fs, nperseg, fmin, fmax, log_scale, global_max=None):
    eng = PlotEngine()
    settings = {"nperseg": nperseg, "fmin": fmin, "fmax": fmax, "log_scale": log_scale,
                "mode_raw": "Spectrogram", "mode_proc": "None", "draw_raw": False, "draw_proc": False}
    rec = _Recorder(eng.ax_spec)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        if global_max is None:
            eng.plot_extra(x, None, fs, settings)
        else:
            eng.last_fs, eng.last_settings, eng.spec_data_source = fs, settings, x
            eng._plot_spectrogram(x, fs, settings, global_max)
        t_feat, feats = eng._calculate_features(x, fs, settings)
    out = {
        "last_f": eng.last_f, "last_t": np.asarray(eng.last_t), "last_Sxx": eng.last_Sxx,
        "image": rec.image if rec.image is not None else np.zeros((0, 0)),
        "feat_t": np.zeros(0) if t_feat is None else t_feat,
        "feats": np.zeros((0, 2)) if feats is None else feats,
        "abs_power": np.float64(eng.calculate_absolute_power()),
        "n_warnings": np.int64(sum(issubclass(w.category, UserWarning) and "nperseg" in str(w.message)
                                   for w in wlist)),
    }
    bp = eng.calculate_band_powers()
    out["band_names"] = np.array(list(bp.keys()))
    out["band_values"] = np.array([float(v) for v in bp.values()])
    return out


def g1_reference_engine(PlotEngine):
    x = cfg1_signal()
    cases = {}
    specs = [
        # tag, signal, fs, nperseg, fmin, fmax, log, global_max
        ("cfg1_lin", x, 16000.0, 512, 0.0, 8000.0, False, None),
        ("cfg1_log", x, 16000.0, 512, 0.0, 8000.0, True, None),
        ("cfg1_band_log", x, 16000.0, 512, 100.0, 3000.0, True, None),
        ("cfg1_gmax", x, 16000.0, 512, 0.0, 8000.0, True, 2.5e-4),
        ("cfg1_f32", x.astype(np.float32), 16000.0, 512, 0.0, 8000.0, True, None),
        ("eeg_default", eeg_like(), 500.0, 1024, 0.0, 30.0, True, None),     # GUI defaults GUI.py:212-214
        ("eeg_lin_256", eeg_like(), 500.0, 256, 0.0, 250.0, False, None),
        ("eeg_np2_1000", eeg_like(), 500.0, 1000, 0.0, 100.0, True, None),   # non-pow2 nperseg reachable in GUI
        ("short_clamp", x[:300], 16000.0, 512, 0.0, 8000.0, True, None),     # N < nperseg -> warning + clamp
        ("zeros", np.zeros(4096), 1000.0, 256, 0.0, 500.0, True, None),      # all-zero -> degenerate dB range
        ("const", np.full(4096, 2.5), 1000.0, 256, 0.0, 500.0, True, None),  # constant -> detrend gives 0
        ("empty_mask", x, 16000.0, 512, 9000.0, 9500.0, False, None),        # Sxx.size == 0 path
    ]
    for tag, sig, fs, nper, fmin, fmax, log, gmax in specs:
        r = reference_run(PlotEngine, sig, fs, nper, fmin, fmax, log, gmax)
        for k, v in r.items():
            cases[f"{tag}__{k}"] = v
        cases[f"{tag}__args"] = np.array([fs, nper, fmin, fmax, float(log), -1.0 if gmax is None else gmax])
    cases["meta"] = np.array(_meta(source="reference PlotEngine via stand-in import (PlotEngine.py:78-145,229-242,686-719)",
                                   cfg1_sha=sha(x), eeg_sha=sha(eeg_like())))
    _save("g1_reference_engine.npz", **cases)


def g2_cfg1_extended():
    x = cfg1_signal()
    out = {"meta": np.array(_meta(source="scipy.signal.spectrogram", cfg1_sha=sha(x)))}
    for tag, kw in {
        "ref": dict(nperseg=512),
        "hann256": dict(nperseg=512, window="hann", noverlap=256),
        "hann256_nodetrend": dict(nperseg=512, window="hann", noverlap=256, detrend=False),
        "hann256_linear": dict(nperseg=512, window="hann", noverlap=256, detrend="linear"),
        "hann256_mag": dict(nperseg=512, window="hann", noverlap=256, mode="magnitude"),
        "hann256_spectrum": dict(nperseg=512, window="hann", noverlap=256, scaling="spectrum"),
        "hann256_complex": dict(nperseg=512, window="hann", noverlap=256, mode="complex"),
        "hann256_nfft1024": dict(nperseg=512, window="hann", noverlap=256, nfft=1024),
        "hann256_angle": dict(nperseg=512, window="hann", noverlap=256, mode="angle"),
        "hann256_phase": dict(nperseg=512, window="hann", noverlap=256, mode="phase"),       # unwrap along frequency (scipy:990-992)
    }.items():
        for dt in (np.float64, np.float32):
            f, t, s = ss.spectrogram(x.astype(dt), fs=16000.0, **{"scaling": "density", "mode": "psd", **kw})
            key = f"{tag}_{np.dtype(dt).name}"
            out[key + "__f"], out[key + "__t"], out[key + "__Sxx"] = f, t, s
    _save("g2_cfg1_extended.npz", **out)


def g3_cfg2_sampled():
    clips = cfg2_clips(2)
    out = {"meta": np.array(_meta(source="scipy.signal.spectrogram", clips_sha=sha(clips)))}
    rng = np.random.default_rng(99)
    for tag, kw in {"ext": dict(window="hann", noverlap=768), "ref": {}}.items():
        f, t, s = ss.spectrogram(clips, fs=48000.0, nperseg=1024, scaling="density", mode="psd", **kw)
        # s: [clip, freq, time] f32
        nfr = s.shape[-1]
        idx = np.sort(np.concatenate([[0, 1, nfr - 2, nfr - 1], rng.choice(np.arange(2, nfr - 2), 12, replace=False)]))
        out[f"{tag}__f"], out[f"{tag}__t"] = f, t
        out[f"{tag}__frame_idx"] = idx
        out[f"{tag}__frames"] = np.ascontiguousarray(np.moveaxis(s[:, :, idx], -1, 1))   # [clip, 16, 513]
        out[f"{tag}__frame_sums"] = s.astype(np.float64).sum(axis=1)                      # [clip, nfr]
        out[f"{tag}__frame_max"] = s.max(axis=1)
        out[f"{tag}__sha256"] = np.array(sha(np.moveaxis(s, -1, 1)))                      # frame-major f32
        out[f"{tag}__dtype"] = np.array(str(s.dtype))
    _save("g3_cfg2_sampled.npz", **out)


def g4_sweep():
    x = sweep_clip()
    out = {"meta": np.array(_meta(source="scipy.signal.spectrogram", clip_sha=sha(x)))}
    for n in (256, 512, 1024, 2048, 4096):
        for hop in (64, 128, 256):
            f, t, s = ss.spectrogram(x, fs=48000.0, nperseg=n, window="hann", noverlap=n - hop,
                                     scaling="density", mode="psd")
            nfr = s.shape[-1]
            idx = np.array([0, nfr // 3, (2 * nfr) // 3, nfr - 1])
            k = f"n{n}_h{hop}"
            out[k + "__nframes"] = np.int64(nfr)
            out[k + "__t_ends"] = np.array([t[0], t[-1]])
            out[k + "__frame_idx"] = idx
            out[k + "__frames"] = np.ascontiguousarray(s[:, idx].T)
            out[k + "__frame_sums"] = s.astype(np.float64).sum(axis=0)
    _save("g4_sweep.npz", **out)


def g5_edges():
    rng = np.random.default_rng(2024)
    out = {"meta": np.array(_meta(source="scipy.signal.spectrogram, reference argument set"))}

    def run(tag, x, fs, **kw):
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            f, t, s = ss.spectrogram(x, fs=fs, **{"scaling": "density", "mode": "psd", **kw})
        out[tag + "__x"] = x
        out[tag + "__f"], out[tag + "__t"], out[tag + "__Sxx"] = f, t, s
        out[tag + "__warned"] = np.int64(any("nperseg" in str(w.message) for w in wl))
        out[tag + "__kw"] = np.array(json.dumps({"fs": fs, **{k: (v if not isinstance(v, tuple) else list(v)) for k, v in kw.items()}}))

    run("short", rng.standard_normal(100), 1000.0, nperseg=256)                 # N < nperseg
    run("exact", rng.standard_normal(256), 1000.0, nperseg=256)                 # N == nperseg
    run("exact_plus", rng.standard_normal(256 + 223), 1000.0, nperseg=256)      # 1 frame, tail dropped
    run("two", rng.standard_normal(256 + 224), 1000.0, nperseg=256)             # exactly 2 frames
    run("odd33", rng.standard_normal(500), 1000.0, nperseg=33)                  # odd nperseg: Nyquist doubled
    run("np2_1000", rng.standard_normal(5000), 2000.0, nperseg=1000)            # non-pow2 even
    run("np2_96", rng.standard_normal(1000), 2000.0, nperseg=96)                # GUI step 32
    run("int16", (rng.standard_normal(4000) * 3000).astype(np.int16), 8000.0, nperseg=512)
    run("zeros", np.zeros(2048, np.float32), 1000.0, nperseg=256)
    run("const", np.full(2048, 7.25, np.float32), 1000.0, nperseg=256)
    run("dc_large", (rng.standard_normal(4096) * 1e-3 + 100.0).astype(np.float32), 1000.0, nperseg=512)
    run("n8192", rng.standard_normal(20000).astype(np.float32), 44100.0, nperseg=8192)
    run("n32", rng.standard_normal(700).astype(np.float32), 100.0, nperseg=32)
    run("hop1", rng.standard_normal(300).astype(np.float32), 100.0, nperseg=64, noverlap=63)
    run("default_nperseg", rng.standard_normal(3000), 1000.0)                   # nperseg None -> 256
    run("batch2d", rng.standard_normal((3, 2000)).astype(np.float32), 1000.0, nperseg=128)
    _save("g5_edges.npz", **out)


# ---------------------------------------------------------------------------
# g6: the reference's CONSUMERS driven against the reference engine: GUI.plot_selected's call sequence
# (GUI.py:374-453), Auto-Detect / Learn (GUI.py:455-476, 286-312) and ExportManager.export_to_csv
# (ExportManager.py:13-90).  Stored: engine state the consumers read and the CSV text they write.
# ---------------------------------------------------------------------------
class FakeItem:
    """What GUI.py hands around as tree items: only ``.data(0, Qt.UserRole)`` (the display name) is ever read."""

    def __init__(self, name):
        self.name = name

    def data(self, column, role):
        return self.name


def consumer_sweeps():
    """three "sweeps" of one recording + one of another, 500 Hz, eeg-like with bursts at known places"""
    rng = np.random.default_rng(11)
    out = []
    for i, (name, n) in enumerate((("/data/recA_sweep0", 6000), ("/data/recA_sweep1", 5000), ("/data/recB_sweep3", 7000))):
        t = np.arange(n) / 500.0
        x = 0.2 * rng.standard_normal(n) + 1.0
        for b0 in (2.0 + i, 7.0 + 0.5 * i):
            x += np.where((t > b0) & (t < b0 + 1.5), 2.0 * np.sin(2 * np.pi * 12 * t), 0.0)
        out.append((name, x))
    return out


CONSUMER_SETTINGS = {"combine": False, "draw_raw": True, "draw_proc": False, "mode_raw": "Both", "mode_proc": "None",
                     "nperseg": 256, "fmin": 5.0, "fmax": 30.0, "log_scale": False}


def g6_consumers(PlotEngine):
    import tempfile
    ExportManager = import_reference_plotengine.export_manager
    sweeps = consumer_sweeps()
    out = {"meta": np.array(_meta(source="reference PlotEngine + ExportManager via stand-in import; HMM = tests/hmm_standin.py",
                                  sweeps_sha=sha(np.concatenate([x for _, x in sweeps]))))}

    def export(eng):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "bursts.csv")
            msg = ExportManager().export_to_csv(path, eng)
            text = open(path).read() if os.path.exists(path) else ""
        return np.array(msg), np.array(text)

    for tag, combine in (("single", False), ("combined", True)):
        eng = PlotEngine()
        settings = dict(CONSUMER_SETTINGS, combine=combine)
        infos = [{"item": FakeItem(n), "signal_raw": x, "signal_proc": None, "fs": 500.0} for n, x in (sweeps if combine else sweeps[:1])]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            eng.plot_sweeps(infos, settings)                       # GUI.py:437
            eng.draw()                                             # GUI.py:448
        out[f"{tag}__abs_power"] = np.float64(eng.calculate_absolute_power())     # GUI.py:451
        out[f"{tag}__last_t"] = np.asarray(eng.last_t)
        out[f"{tag}__last_f"] = np.asarray(eng.last_f)
        out[f"{tag}__segments"] = np.array([[s["start_time_combined"], s["end_time_combined"]] for s in eng.segment_map]).reshape(-1, 2)
        out[f"{tag}__segment_names"] = np.array([s["source_item"].name for s in eng.segment_map] or [""])
        out[f"{tag}__last_raw_t_end"] = np.float64(eng.last_raw_t[-1]) if len(eng.last_raw_t) else np.float64(-1)
        out[f"{tag}__combined_raw_len"] = np.int64(-1 if eng.combined_raw is None else len(eng.combined_raw))
        msg0, _ = export(eng)                                      # nothing detected yet
        out[f"{tag}__export_empty_msg"] = msg0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev = eng.unsupervised_detect()                         # GUI.py:464
        out[f"{tag}__auto_events"] = np.array(ev, dtype=np.float64).reshape(-1, 2)
        out[f"{tag}__auto_transmat"] = np.asarray(eng.model.transmat_)
        eng.plot_detection_lines(ev)                               # GUI.py:471
        out[f"{tag}__auto_n_patches"] = np.int64(len(eng.burst_patches))
        msg, text = export(eng)                                    # GUI.py:496-527 -> ExportManager.py:13
        out[f"{tag}__auto_csv_msg"], out[f"{tag}__auto_csv"] = msg, text
        # Learn: two hand-drawn regions around known bursts (GUI.py:286-312); plot_detection_lines stands in for the mouse
        rois = [(1.6, 4.0), (6.6, 9.0)] if not combine else [(1.6, 4.0), (14.6, 17.2), (24.5, 27.0)]
        eng.plot_detection_lines(rois)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev2 = eng.learn_and_detect()                           # GUI.py:300
        out[f"{tag}__learn_rois"] = np.array(rois)
        out[f"{tag}__learn_events"] = np.array(ev2, dtype=np.float64).reshape(-1, 2)
        out[f"{tag}__learn_transmat"] = np.asarray(eng.model.transmat_)
        out[f"{tag}__learn_means"] = np.asarray(eng.model.means_)
        eng.plot_detection_lines(ev2)                              # GUI.py:307
        msg, text = export(eng)
        out[f"{tag}__learn_csv_msg"], out[f"{tag}__learn_csv"] = msg, text
        # refined model: Auto-Detect now skips the fit (PlotEngine.py:417) and decodes with the supervised parameters
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev3 = eng.unsupervised_detect()
        out[f"{tag}__refined_events"] = np.array(ev3, dtype=np.float64).reshape(-1, 2)
    _save("g6_consumers.npz", **out)


def main():
    os.makedirs(HERE, exist_ok=True)
    if "--only-consumers" in sys.argv:
        g6_consumers(import_reference_plotengine())
        return
    g2_cfg1_extended()
    g3_cfg2_sampled()
    g4_sweep()
    g5_edges()
    if "--no-reference" not in sys.argv:
        engine = import_reference_plotengine()
        g1_reference_engine(engine)
        g6_consumers(engine)


if __name__ == "__main__":
    main()
