"""PlotEngine -- host-side mirror of the reference's plotting/analysis canvas, numerics on the MI355X.

Drop-in for /root/reference PlotEngine.py: same class name, constructor, public methods and the
attributes GUI.py / ExportManager.py read.  What changed is *where the numbers come from*:

  reference                                        here
  scipy.signal.spectrogram  (PlotEngine.py:113)    spectro.engine.stft        -> HIP kernels
  mask / max / clip / dB / min-max (:114-131)      DeviceSpectrogram.band_slice / .image
  band sum / log10 / diff        (:232-242)        spectro.engine.band_features (fused, no spectrum in HBM)
  np.sum / band ratios           (:686-719)        DeviceSpectrogram.band_totals

matplotlib only draws.  HMM burst detection (PlotEngine.py:244-478) is outside the accelerated path but on
the drop-in surface: ``unsupervised_detect`` / ``learn_and_detect`` keep the reference's behaviour around the model
(features from the device, post-fit transition surgery, supervised re-estimation, event extraction) and need
``hmmlearn`` at call time.  The mouse ROI editor (:480-667) keeps its state fields; handlers are wired by
``set_editing_enabled``.
"""
from __future__ import annotations

import ctypes as _C
import os

import numpy as np
from matplotlib.figure import Figure

from spectro import _capi
from spectro import engine as _engine

try:                                              # a QWidget when Qt is present (GUI.py:157-158) ...
    from matplotlib.backends.backend_qt5agg import FigureCanvasQTAgg as _Canvas
except Exception:                                 # ... a headless Agg canvas otherwise (tests, batch export)
    from matplotlib.backends.backend_agg import FigureCanvasAgg as _Canvas

_SPECTRO_MODES = ("Spectrogram", "Both")
_DEFAULT_BANDS = (                                # PlotEngine.py:699-706
    ("Delta (δ)", 0, 4), ("Theta (θ)", 4, 8), ("Alpha (α)", 8, 13),
    ("Beta (β)", 13, 30), ("Gamma (γ)", 30, 80), ("HFO (ripples)", 80, 250),
)


def _new_hmm():
    try:
        from hmmlearn import hmm
    except Exception:
        return None
    return hmm.GaussianHMM(n_components=4, covariance_type="diag", n_iter=100, random_state=42)


class PlotEngine(_Canvas):
    ROI_COLOR = "blue"
    HOVER_COLOR = "red"

    def __init__(self, *args, **kwargs):
        self.fig = Figure(constrained_layout=True)
        super().__init__(self.fig)
        self.ax_signal = self.ax_spec = None
        self._make_axes()
        # analysis state (names are part of the surface: GUI.py:279-292,456,498,530; ExportManager.py:17-38)
        self.model = _new_hmm()
        self.is_model_refined = False
        self.spec_data_source = None
        self.last_fs = None
        self.last_settings = None
        self.last_t = np.array([])
        self.last_f = None
        self.last_Sxx = None
        self.segment_map = []
        self.currently_plotted_items = []
        self.burst_patches = []
        # ROI editor state
        self.editing_enabled = False
        self.hovered_patch = None
        self.is_adding = False
        self.adding_patch = None
        self.press_x = None
        self.press_cid = self.release_cid = self.motion_cid = None
        # device-side shadow of last_Sxx
        self._dev = None
        self._dev_band = None
        self._dev_shadow_of = None

    # ------------------------------------------------------------------ canvas
    def _make_axes(self):
        grid = self.fig.add_gridspec(nrows=2, ncols=1, height_ratios=[1, 1])
        self.ax_signal = self.fig.add_subplot(grid[0, 0])
        self.ax_spec = self.fig.add_subplot(grid[1, 0], sharex=self.ax_signal)

    _create_axes = _make_axes

    def clear(self):
        """Reset the figure; ``last_detected_events`` / ``last_raw_t`` come into existence here,
        not in ``__init__`` (callers test them with ``hasattr``)."""
        self.burst_patches.clear()
        self.segment_map.clear()
        self.currently_plotted_items.clear()
        self.fig.clf()
        self._make_axes()
        self.last_detected_events = []
        self.last_raw_t = np.array([])
        self.last_fs = None

    def _redraw(self):
        self.fig.canvas.draw()

    # ------------------------------------------------------------------ plotting
    def plot_sweeps(self, sweeps_info, settings):
        """Assemble the signal(s) to show (optionally concatenating sweeps) and plot them.

        ``sweeps_info``: list of ``{'item', 'signal_raw', 'signal_proc', 'fs'}`` (GUI.py:413)."""
        self.clear()
        self.currently_plotted_items = [s["item"] for s in sweeps_info]
        fs0 = sweeps_info[0]["fs"] if sweeps_info else 0
        raw_parts, proc_parts = [], []
        raw_plot = proc_plot = None

        if settings.get("combine", False):
            prefer_proc = settings.get("draw_proc", True)
            chosen, cursor = [], 0.0
            for s in sweeps_info:
                raw = s["signal_raw"]
                proc = raw if s["signal_proc"] is None else s["signal_proc"]
                raw_parts.append(raw)
                proc_parts.append(proc)
                sig = proc if prefer_proc else raw
                if sig is None:
                    continue
                span = len(sig) / s["fs"]
                self.segment_map.append({"start_time_combined": cursor, "end_time_combined": cursor + span,
                                         "source_item": s["item"]})
                chosen.append(sig)
                cursor += span
            if chosen:
                joined = np.concatenate(chosen)
                if prefer_proc and any(s["signal_proc"] is not None for s in sweeps_info):
                    proc_plot = joined
                else:
                    raw_plot = joined
                self.last_raw_t = np.arange(len(joined)) / fs0
                self.last_fs = fs0
        else:
            first = sweeps_info[0]
            raw_plot = first["signal_raw"] if settings.get("draw_raw") else None
            proc_plot = first["signal_proc"] if settings.get("draw_proc") else None

        self.plot_extra(signal_raw=raw_plot, signal_proc=proc_plot, fs=fs0, settings=settings)
        self.combined_raw = np.concatenate(raw_parts) if raw_parts else None
        self.combined_proc = np.concatenate(proc_parts) if proc_parts else None

    def plot_extra(self, signal_raw, signal_proc, fs, settings, global_max=None):
        self.last_fs = fs
        if self.ax_signal is None:
            self._make_axes()
        for wanted, sig, colour, label in (("draw_raw", signal_raw, "blue", "Raw"),
                                           ("draw_proc", signal_proc, "black", "Processed")):
            if settings.get(wanted) and sig is not None:
                self.ax_signal.plot(np.arange(len(sig)) / fs, sig, color=colour, label=label)
        if self.ax_signal.has_data():
            self.ax_signal.set_ylabel("Amplitude")
            legend = self.ax_signal.legend(loc="upper right", frameon=True)
            if hasattr(self, "last_raw_t") and len(self.last_raw_t) > 1:
                self.ax_signal.set_xlim(0, self.last_raw_t[-1])
            legend.set_zorder(100)

        source = None                                            # processed wins over raw
        if settings["mode_proc"] in _SPECTRO_MODES and signal_proc is not None:
            source = signal_proc
        elif settings["mode_raw"] in _SPECTRO_MODES and signal_raw is not None:
            source = signal_raw
        if source is not None:
            self.spec_data_source = source
            self.last_fs = fs
            self.last_settings = settings
            self._plot_spectrogram(source, fs, settings, global_max)
        self._redraw()
        self._redraw()

    def _plot_spectrogram(self, data, fs, settings, global_max=None):
        """STFT on the device, then mask/store (A8), normalise (A9), optional dB + min-max (A10), draw."""
        fmin, fmax = settings["fmin"], settings["fmax"]
        self._drop_device()
        dev = _engine.stft(np.asarray(data), fs=fs, nperseg=settings["nperseg"], scaling="density", mode="psd")
        k_lo, k_hi = _engine.bin_range(dev.f, fmin, fmax)
        f = dev.f[k_lo:k_hi + 1].copy()
        t = dev.t.copy()
        self.last_f, self.last_t = f, t
        self.last_Sxx = np.ascontiguousarray(dev.band_slice(k_lo, k_hi))
        self._dev, self._dev_band, self._dev_shadow_of = dev, (k_lo, k_hi), self.last_Sxx
        if self.last_Sxx.size == 0:
            self.last_t = np.array([])
            return
        if settings.get("fast_image", os.environ.get("SPECTRO_FAST_IMAGE") == "1") and len(t) > 1 and len(f) > 1:
            mesh = self._draw_fast_image(dev, k_lo, k_hi, t, f, settings["log_scale"], global_max)
        else:
            image = dev.image(k_lo, k_hi, settings["log_scale"], global_max)
            mesh = self.ax_spec.pcolormesh(t, f, image, shading="auto", cmap="jet", vmin=0.0, vmax=1.0, zorder=0)
        self.ax_spec.set_ylabel("Frequency (Hz)")
        self.ax_spec.set_xlabel("Time (s)")
        self.fig.colorbar(mesh, ax=self.ax_spec, orientation="vertical", label="Normalized Power")
        t_end = t[-1]
        if hasattr(self, "last_raw_t") and len(self.last_raw_t) > 1:
            t_end = max(t_end, self.last_raw_t[-1])
        self.ax_spec.set_xlim(0, t_end)
        self.ax_spec.set_ylim(fmin, f[-1])

    def _draw_fast_image(self, dev, k_lo, k_hi, t, f, log_scale, global_max):
        """Display epilogue, opt-in through ``settings['fast_image']`` or ``SPECTRO_FAST_IMAGE=1``: the
        ``pcolormesh(..., cmap='jet', vmin=0, vmax=1)`` of PlotEngine.py:134 draws one quad per bin and is 83 % of the
        reference's plot latency; the same picture is one ``imshow`` of the RGBA image that the device (f32 spectra) or a
        table lookup (f64) produces with matplotlib's own jet arithmetic.  Cell edges sit half a step around t and f, as
        pcolormesh's shading='auto' puts them."""
        from matplotlib.cm import ScalarMappable
        from matplotlib.colors import Normalize
        if dev.dtype_code == _capi.F32:
            rgba = dev.image_rgba(k_lo, k_hi, log_scale, global_max)
        else:
            image = dev.image(k_lo, k_hi, log_scale, global_max)
            lut = np.empty((256, 4), np.uint8)
            _capi.check(_capi.lib().sg_jet_lut(lut.ctypes.data_as(_C.POINTER(_C.c_uint8))))
            rgba = lut[np.clip((image * 256.0).astype(np.int64), 0, 255)]
        dt, df = t[1] - t[0], f[1] - f[0]
        self.ax_spec.imshow(rgba, origin="lower", aspect="auto", interpolation="nearest", zorder=0,
                            extent=(t[0] - dt / 2, t[-1] + dt / 2, f[0] - df / 2, f[-1] + df / 2))
        return ScalarMappable(norm=Normalize(0.0, 1.0), cmap="jet")

    def plot_single_signal(self, name, signal, fs, use_log=False):
        """One labelled time-domain trace (batch export helper)."""
        self.clear()
        ax = self.fig.add_subplot(111)
        ax.plot(np.arange(len(signal)) / fs, signal)
        ax.set_xlabel("Time (s)")
        ax.set_ylabel("Amplitude")
        if use_log:
            ax.set_yscale("log")
        self.draw()

    # ------------------------------------------------------------------ features / powers
    def _calculate_features(self, signal, fs=None, settings=None):
        """A11: ``(t, [log10 band power, its first difference])`` or ``(None, None)`` -- fused on the device."""
        fs = fs or self.last_fs
        settings = settings or self.last_settings
        t, feats = _engine.band_features(np.asarray(signal), fs, settings["nperseg"], settings["fmin"], settings["fmax"])
        return t, feats

    def _device_shadow(self):
        """Device copy of whatever ``last_Sxx`` currently is (re-uploaded if a caller replaced the attribute)."""
        if self._dev is not None and self._dev_shadow_of is self.last_Sxx:
            k_lo, k_hi = self._dev_band
            return self._dev, k_lo, k_hi - k_lo + 1
        from spectro import _capi
        s = np.asarray(self.last_Sxx)
        dt = np.float32 if s.dtype == np.float32 else np.float64
        rows = np.ascontiguousarray(s.T, dtype=dt)                 # frame-major [n_frames][n_mask]
        buf = _capi.DeviceBuffer(max(rows.nbytes, 8))
        buf.upload(rows)
        _capi.stream_sync()
        self._drop_device()
        self._dev = _engine.DeviceSpectrogram(buf, _capi.F32 if dt == np.float32 else _capi.F64, 1, rows.shape[0],
                                              max(rows.shape[1], 1), np.asarray(self.last_f), np.asarray(self.last_t), self.last_fs)
        self._dev_band, self._dev_shadow_of = (0, rows.shape[1] - 1), self.last_Sxx
        return self._dev, 0, rows.shape[1]

    def _drop_device(self):
        if self._dev is not None:
            self._dev.free()
        self._dev = self._dev_band = self._dev_shadow_of = None

    def calculate_absolute_power(self):
        """A12: total of the stored (masked, linear) PSD, or None before the first plot."""
        if self.last_Sxx is None:
            return None
        s = np.asarray(self.last_Sxx)
        if s.size == 0:
            return s.dtype.type(0)
        dev, k0, width = self._device_shadow()
        return s.dtype.type(dev.band_totals([(k0, k0 + width)])[0])

    def calculate_band_powers(self, bands=None):
        """A13: power of each band relative to the total; band edges are half-open ``[low, high)``."""
        if self.last_Sxx is None or self.last_f is None:
            return None
        if bands is None:
            bands = {name: (lo, hi) for name, lo, hi in _DEFAULT_BANDS}
        f = np.asarray(self.last_f)
        if np.asarray(self.last_Sxx).size == 0:
            return {name: 0.0 for name in bands}
        dev, k0, width = self._device_shadow()
        ranges = [(k0, k0 + width)]
        for lo, hi in bands.values():
            a = int(np.searchsorted(f, lo, side="left"))
            b = int(np.searchsorted(f, hi, side="left"))
            ranges.append((k0 + a, k0 + max(a, b)))
        sums = dev.band_totals(ranges)
        total = sums[0]
        if total < 1e-18:
            return {name: 0.0 for name in bands}
        kind = np.asarray(self.last_Sxx).dtype.type
        return {name: kind(max(sums[i + 1] / total, 0.0)) for i, name in enumerate(bands)}

    # ------------------------------------------------------------------ detection (hmmlearn, not accelerated)
    def _need_model(self):
        if self.model is None:
            self.model = _new_hmm()
        if self.model is None:
            raise RuntimeError("hmmlearn is not installed: HMM burst detection is outside the accelerated path")
        return self.model

    def reset_model(self):
        self.model = _new_hmm()
        self.is_model_refined = False

    @staticmethod
    def _merge_overlapping_events(events, tolerance=1e-6):
        merged = []
        for start, end in sorted(events, key=lambda e: e[0]):
            if merged and start <= merged[-1][1] + tolerance:
                merged[-1] = (merged[-1][0], max(merged[-1][1], end))
            else:
                merged.append((start, end))
        return merged

    @staticmethod
    def _open_escape_routes(model):
        """After an unsupervised fit (PlotEngine.py:422-438): a non-baseline state that cannot reach the baseline state
        (transition probability < 1e-5) donates 5 % of its self-transition, at most 0.05, to that transition -- provided
        its self-transition exceeds 0.1 -- so a burst state can always end.  Baseline = lowest mean log-power."""
        quiet = int(np.argmin(np.asarray(model.means_)[:, 0]))
        trans = np.array(model.transmat_, dtype=float, copy=True)
        for s in range(model.n_components):
            if s != quiet and trans[s, quiet] < 1e-5 and trans[s, s] > 0.1:
                gift = min(0.05 * trans[s, s], 0.05)
                trans[s, s] -= gift
                trans[s, quiet] += gift
        model.transmat_ = trans

    def unsupervised_detect(self):
        if self.spec_data_source is None:
            raise ValueError("Please plot a spectrogram before detecting.")
        t, feats = self._calculate_features(self.spec_data_source, self.last_fs, self.last_settings)
        if t is None or len(t) == 0:
            return []
        model = self._need_model()
        if not self.is_model_refined:
            if len(feats) < model.n_components:
                raise ValueError("Not enough data to train the model. Signal may be too short.")
            model.fit(feats)
            self._open_escape_routes(model)
        states = np.asarray(model.predict(feats))
        means = np.asarray(model.means_)
        if means.ndim != 2:
            raise TypeError(f"HMM means has unexpected dimension: {means.ndim}. Expected 2.")
        quiet = int(np.argmin(means[:, 0]))
        active = states != quiet
        # an event opens at the last baseline frame before activity and closes at the last active frame
        # (PlotEngine.py:447-470); one still open at the end of the signal closes at t[-1]
        events, start = [], None
        for i in range(1, len(states)):
            if start is None and not active[i - 1] and active[i]:
                start = t[i - 1]
            elif start is not None and active[i - 1] and not active[i]:
                if t[i - 1] > start:
                    events.append((start, t[i - 1]))
                start = None
        if start is not None:
            events.append((start, t[-1]))
        self.last_detected_events = self._merge_overlapping_events(events)
        return self.last_detected_events

    def _find_burst_in_roi(self, roi_feats, roi_t):
        """Inside one hand-drawn region: a fresh 2-state model, the state with the higher mean log-power is the burst,
        its first and last frame are the burst's extent (PlotEngine.py:389-409)."""
        model = self._need_model()
        if len(roi_feats) < model.n_components:
            return None
        from hmmlearn import hmm
        import warnings
        local = hmm.GaussianHMM(n_components=2, covariance_type="diag", n_iter=50, random_state=42)
        try:
            with warnings.catch_warnings():
                warnings.filterwarnings("ignore", category=UserWarning, module="hmmlearn")
                local.fit(roi_feats)
        except ValueError:
            return None
        means = np.asarray(local.means_)
        if means.ndim != 2:
            raise TypeError(f"HMM means has unexpected dimension: {means.ndim}. Expected 2.")
        hits = np.where(np.asarray(local.predict(roi_feats)) == int(np.argmax(means[:, 0])))[0]
        if len(hits) == 0:
            return None
        return roi_t[hits[0]], roi_t[hits[-1]]

    def _train_supervised(self, feats, labels):
        """Model parameters from labelled frames (PlotEngine.py:329-387): per state mean / variance (+1e-6; a single frame
        gives variance 1e-6, no frame gives mean 0), transition counts normalised per row (a row without counts
        becomes a pure self-transition), state 3 ("falling edge") always returns to state 0, start in state 0."""
        model = self._need_model()
        k, dim = model.n_components, feats.shape[1]
        means, variances = np.zeros((k, dim)), np.full((k, dim), 1e-6)
        for s in range(k):
            rows = feats[labels == s]
            if len(rows) > 1:
                means[s], variances[s] = rows.mean(axis=0), rows.var(axis=0) + 1e-6
            elif len(rows) == 1:
                means[s] = rows[0]
        model.means_, model.covars_ = means, variances
        counts = np.zeros((k, k))
        np.add.at(counts, (labels[:-1], labels[1:]), 1.0)
        totals = counts.sum(axis=1, keepdims=True)
        trans = np.divide(counts, totals, out=np.zeros_like(counts), where=totals != 0)
        for s in np.where(totals.ravel() == 0)[0]:
            trans[s, s] = 1.0
        if k > 3:
            trans[3, :] = 0.0
            trans[3, 0] = 1.0
        model.transmat_ = trans
        model.startprob_ = np.array([1.0] + [0.0] * (k - 1)) if k == 4 else np.eye(k)[0]
        self.is_model_refined = True

    def learn_and_detect(self):
        """Supervised refinement from the hand-drawn regions (PlotEngine.py:244-327): each region is narrowed to its burst
        by a local 2-state model, frames are labelled 0 baseline / 1 rising edge / 2 burst / 3 falling edge, the 4-state
        model is re-estimated from the labels and decoded over the whole signal; an event runs from the first frame in
        state 1 or 2 to the next frame in state 0."""
        if self.spec_data_source is None:
            raise ValueError("Please plot a spectrogram before learning.")
        if not self.burst_patches:
            raise ValueError("No manual regions provided to learn from.")
        t, feats = self._calculate_features(self.spec_data_source, self.last_fs, self.last_settings)
        if t is None:
            return []
        feats = np.asarray(feats)
        bursts = []
        for pair in self.burst_patches:
            try:
                lo, hi = pair[0].event_data
            except AttributeError:                                  # spans drawn by older code carry no exact times
                box = pair[0].get_extents()
                lo, hi = box.x0, box.x1
            inside = np.where((t >= lo) & (t <= hi))[0]
            if len(inside) < 2:
                continue
            found = self._find_burst_in_roi(feats[inside, :], t[inside])
            if found:
                bursts.append(found)
        if not bursts:
            raise ValueError("Could not identify a clear burst in any of the provided regions.")
        labels = np.zeros(len(t), dtype=int)
        for b0, b1 in bursts:
            i0, i1 = np.searchsorted(t, b0), np.searchsorted(t, b1)
            if i0 >= i1:
                continue
            labels[i0] = 1
            if i1 > i0 + 1:
                labels[i0 + 1:i1] = 2
            if i1 < len(labels):
                labels[i1] = 3
        self._train_supervised(feats, labels)
        states = np.asarray(self.model.predict(feats))
        events, start = [], None
        for i, st in enumerate(states):
            if start is None and st in (1, 2):
                start = t[i]
            elif start is not None and st == 0:
                if t[i] > start:
                    events.append((start, t[i]))
                start = None
        if start is not None:
            events.append((start, t[-1]))
        self.last_detected_events = self._merge_overlapping_events(events)
        return self.last_detected_events

    # ------------------------------------------------------------------ ROI patches (mouse editor, PlotEngine.py:480-667)
    def set_editing_enabled(self, enabled):
        """(Dis)connect the three mouse callbacks; a fresh start either way (GUI.py:315,434,447)."""
        for attr in ("press_cid", "release_cid", "motion_cid"):
            cid = getattr(self, attr)
            if cid:
                self.fig.canvas.mpl_disconnect(cid)
            setattr(self, attr, None)
        self.is_adding = self.adding_patch = self.press_x = self.hovered_patch = None
        self.editing_enabled = enabled
        if enabled:
            connect = self.fig.canvas.mpl_connect
            self.press_cid = connect("button_press_event", self.on_press)
            self.release_cid = connect("button_release_event", self.on_release)
            self.motion_cid = connect("motion_notify_event", self.on_motion)

    @staticmethod
    def _get_correct_xdata(event):
        """Data-space x of a mouse event; falls back to the inverse axes transform of the pixel position."""
        ax = event.inaxes
        if ax is None:
            return None
        if event.xdata is not None:
            return event.xdata
        try:
            return ax.transData.inverted().transform((event.x, event.y))[0]
        except Exception:
            return None

    def _on_axes(self, event):
        return self.editing_enabled and event.inaxes in (self.ax_signal, self.ax_spec) and event.xdata is not None

    def _paint(self, pair, colour):
        for patch in pair:
            patch.set_color(colour)

    def _drop_rubber_band(self):
        if self.adding_patch:
            for patch in self.adding_patch:
                patch.remove()
        self.adding_patch = None

    def _span_pair(self, x0, x1, **style):
        return tuple(ax.axvspan(x0, x1, **style) for ax in (self.ax_signal, self.ax_spec))

    def on_motion(self, event):
        if not self._on_axes(event):
            if self.hovered_patch:                      # the pointer left the axes: un-highlight
                self._paint(self.hovered_patch, self.ROI_COLOR)
                self.hovered_patch = None
                self._redraw()
            return
        x = self._get_correct_xdata(event)
        if x is None:
            return
        if self.is_adding and self.press_x is not None:  # rubber band of the region being drawn
            self._drop_rubber_band()
            self.adding_patch = self._span_pair(self.press_x, x, color="green", alpha=0.3, zorder=5)
            self._redraw()
            return
        which = 0 if event.inaxes is self.ax_signal else 1
        under = next((pair for pair in self.burst_patches if pair[which].contains(event)[0]), None)
        if under is not self.hovered_patch:
            if self.hovered_patch:
                self._paint(self.hovered_patch, self.ROI_COLOR)
            if under:
                self._paint(under, self.HOVER_COLOR)
            self.hovered_patch = under
            self._redraw()

    def delete_hovered(self):
        """Context-menu "Delete" on the highlighted region."""
        if self.hovered_patch:
            self.remove_patch(self.hovered_patch)
            self.hovered_patch = None

    def merge_into_hovered(self):
        """Context-menu "Merge": the regions lying inside the highlighted one are replaced, together with it, by one event
        from the earliest start to the latest end of the contained ones (exact times come from ``event_data``)."""
        box = self.hovered_patch
        if not box:
            return
        outer = box[0].get_extents()
        inner = [pair for pair in self.burst_patches if pair is not box
                 and pair[0].get_extents().x0 >= outer.x0 and pair[0].get_extents().x1 <= outer.x1]
        if not inner:
            return
        gone = {pair[0].event_data for pair in inner + [box]}
        merged = (min(pair[0].event_data[0] for pair in inner), max(pair[0].event_data[1] for pair in inner))
        self.last_detected_events = sorted([ev for ev in self.last_detected_events if ev not in gone] + [merged])
        self.plot_detection_lines(self.last_detected_events)
        self.hovered_patch = None
        self._redraw()

    def on_press(self, event):
        if not self._on_axes(event):
            return
        x = self._get_correct_xdata(event)
        if x is None:
            return
        if event.button == 3 and self.hovered_patch:
            choice = self._context_menu()
            if choice == "Delete":
                self.delete_hovered()
            elif choice == "Merge":
                self.merge_into_hovered()
                return
        if event.button == 1:
            self.is_adding, self.press_x = True, x

    def _context_menu(self):
        """Qt pop-up with "Delete" / "Merge" at the cursor; None without Qt (headless canvas) or when dismissed."""
        try:
            from PyQt5 import QtWidgets
            from PyQt5.QtGui import QCursor
            menu = QtWidgets.QMenu(self.parent())
        except Exception:
            return None
        delete, merge = menu.addAction("Delete"), menu.addAction("Merge")
        chosen = menu.exec_(QCursor.pos())
        return "Delete" if chosen == delete else "Merge" if chosen == merge else None

    def on_release(self, event):
        x = self._get_correct_xdata(event)
        if not self.editing_enabled or not self.is_adding or x is None:
            self._drop_rubber_band()
            self.is_adding = self.press_x = None
            return
        self._drop_rubber_band()
        if hasattr(self, "last_raw_t") and len(self.last_raw_t) > 1:
            narrowest = self.last_raw_t[1] - self.last_raw_t[0]        # one sample
        else:
            narrowest = 1.0 / self.last_fs if self.last_fs else 0.01
        if abs(self.press_x - x) >= narrowest:
            span = (min(self.press_x, x), max(self.press_x, x))
            pair = self._span_pair(span[0], span[1], color=self.ROI_COLOR, alpha=0.5, zorder=10)
            for patch in pair:
                patch.event_data = span
            self.burst_patches.append(pair)
            self.last_detected_events.append(span)
        self.is_adding = self.press_x = None
        self._redraw()

    def remove_patch(self, patch_pair):
        for p in patch_pair:
            p.remove()
        if patch_pair in self.burst_patches:
            self.burst_patches.remove(patch_pair)
        self._redraw()

    def plot_detection_lines(self, event_pairs):
        """Replace the ROI spans by one (signal, spectrogram) span pair per ``(t_start, t_end)``."""
        for pair in list(self.burst_patches):
            self.remove_patch(pair)
        for t0, t1 in event_pairs:
            pair = tuple(ax.axvspan(t0, t1, color=self.ROI_COLOR, alpha=0.5, zorder=10)
                         for ax in (self.ax_signal, self.ax_spec))
            for p in pair:
                p.event_data = (t0, t1)
            self.burst_patches.append(pair)
        self._redraw()
