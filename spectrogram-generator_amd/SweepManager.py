"""SweepManager -- recording loader with the reference's surface, plus the batch entry points.

Mirror of /root/reference SweepManager.py: ``SweepManager()``, ``.data`` (name -> entry dict,
mutated directly by GUI.py:237,256-267), ``load_file(path) -> [names]``, ``get_signal(name,
processed) -> (signal, fs)``.  Entry schema (SweepManager.py:48-55,139-146):
``{filepath, sweep_idx, fs_raw, fs, raw, processed}``; display names ``"{base}_sweep{i}"``
(ExportManager.py:60-61 and GUI.py:509 match on ``_sweep\\d+$``).

File decoding is not the accelerated path: ``.abf`` needs pyabf and ``.h5`` needs neo (both imported
lazily, so the module loads without them); ``.wav`` (new here, PCM 8/16/24/32 and IEEE float) is
parsed with numpy.  New on top of the reference: ``add_signal`` and the device batch calls
``spectrogram_batch`` / ``parameter_sweep`` (BASELINE cfg2 / cfg4) built on ``spectro.engine``.
"""
from __future__ import annotations

import os
import struct

import numpy as np


class SweepManager:
    def __init__(self):
        self.data = {}

    # ------------------------------------------------------------------ loading
    def load_file(self, filepath: str):
        ext = os.path.splitext(filepath)[1].lower()
        loader = {".abf": self._load_abf, ".h5": self._load_h5, ".wav": self._load_wav}.get(ext)
        if loader is None:
            raise ValueError(f"Unsupported file type: {ext}")
        return loader(filepath)

    def _register(self, name, filepath, idx, fs, fs_raw, raw, processed):
        self.data[name] = {"filepath": filepath, "sweep_idx": idx, "fs_raw": fs_raw, "fs": fs,
                           "raw": raw, "processed": processed}
        return name

    def add_signal(self, name, raw, fs, processed=None, filepath=None, sweep_idx=0):
        """Register an in-memory array under the reference's entry schema."""
        return self._register(name, filepath, sweep_idx, fs, fs, None if raw is None else np.asarray(raw),
                              None if processed is None else np.asarray(processed))

    def _load_abf(self, filepath):
        try:
            import pyabf
        except ImportError as e:                                   # pragma: no cover - optional dependency
            raise ValueError(f"Cannot read ABF files: pyabf is not installed ({e})")
        abf = pyabf.ABF(filepath)
        if abf.channelCount < 1:
            raise ValueError("Expected at least 1 channel in ABF file.")
        base = os.path.splitext(os.path.basename(filepath))[0]
        names = []
        for i in range(abf.sweepCount):
            chans = []
            for ch in range(min(abf.channelCount, 2)):
                abf.setSweep(i, channel=ch)
                chans.append(abf.sweepY.copy())
            # channel 0 unless it is all zero and channel 1 is not (SweepManager.py:41-46)
            live = [c for c in chans if np.any(np.abs(c) > 0)]
            raw = live[0] if live else chans[0]
            names.append(self._register(f"{base}_sweep{i}", filepath, i, abf.dataRate, abf.dataRate, raw, None))
        return names

    def _load_h5(self, filepath):
        try:
            from neo.io import NixIO
        except ImportError as e:                                   # pragma: no cover - optional dependency
            raise ValueError(f"Failed to open H5 via NixIO: neo is not installed ({e})")
        try:
            block = NixIO(filename=filepath, mode="ro").read_block(lazy=False)
        except Exception as e:
            raise ValueError(f"Failed to open H5 via NixIO: {e}")
        names = []
        segments = getattr(block, "segments", None) or []
        base = os.path.splitext(os.path.basename(filepath))[0]

        def label(sig):
            n = sig.name
            return (n.decode("utf-8", "ignore") if isinstance(n, bytes) else str(n)).lower()

        def rate(sig):
            try:
                return float(sig.sampling_rate.rescale("Hz").magnitude)
            except Exception:
                return None

        for i, seg in enumerate(segments):
            sigs = list(seg.analogsignals)
            proc = next((s for s in sigs if "proc" in label(s)), None)
            raw = next((s for s in sigs if "raw" in label(s)), None)
            if proc is None and raw is None and sigs:
                proc = sigs[0]
            proc = raw if proc is None else proc
            raw = proc if raw is None else raw
            if proc is None:
                continue
            fs_proc = rate(proc)
            fs_raw = fs_proc if raw is proc else rate(raw)
            fs = fs_proc if fs_proc is not None else fs_raw
            if fs is None:
                continue
            names.append(self._register(f"{base}_sweep{i}", filepath, i, fs, fs_raw,
                                        raw.magnitude.copy().reshape(-1), proc.magnitude.copy().reshape(-1)))
        return names

    def _load_wav(self, filepath):
        """RIFF/WAVE reader: one entry per channel ("sweep"); int16 data stays int16 (the device converts)."""
        fs, channels = read_wav(filepath)
        base = os.path.splitext(os.path.basename(filepath))[0]
        return [self._register(f"{base}_sweep{i}", filepath, i, fs, fs, ch, None) for i, ch in enumerate(channels)]

    # ------------------------------------------------------------------ lookup
    def get_signal(self, display_name: str, processed: bool = False):
        """``(signal, fs)``; processed falls back to raw (+ ``fs_raw``); KeyError messages as the reference."""
        entry = self.data.get(display_name)
        if entry is None:
            raise KeyError(f"{display_name} not found in SweepManager.data")
        raw_fs = entry.get("fs_raw", entry.get("fs"))
        if processed and entry.get("processed") is not None:
            sig, fs, what = entry["processed"], entry.get("fs"), "processed"
        else:
            sig, fs, what = entry.get("raw"), raw_fs, ("processed" if processed else "raw")
            if sig is None:
                raise KeyError(f"No 'processed' or 'raw' signal for {display_name}" if processed
                               else f"No 'raw' signal for {display_name}")
        if fs is None:
            raise KeyError(f"No sampling rate for {what} signal of {display_name}")
        return sig, fs

    # ------------------------------------------------------------------ device batch calls (new)
    def _stack(self, names, processed):
        sigs, rates = zip(*(self.get_signal(n, processed) for n in names))
        if len(set(rates)) != 1 or len({len(s) for s in sigs}) != 1:
            raise ValueError("batch calls need clips of one length and one sampling rate")
        return np.stack([np.asarray(s) for s in sigs]), rates[0]

    def spectrogram_batch(self, names, nperseg, processed=False, pipeline=False, **kw):
        """All named clips through the device; returns ``(f, t, Sxx[clip, freq, time])``.

        ``pipeline=True``: chunked, double-buffered transfers into pinned host memory (``spectro.pipeline``): the upload of
        one chunk overlaps the download of the previous one, int16 PCM travels as int16; same values bit for bit."""
        x, fs = self._stack(names, processed)
        if pipeline:
            from spectro.pipeline import stft_pipelined
            return stft_pipelined(x, fs=fs, nperseg=nperseg, **kw)
        from spectro import spectrogram
        return spectrogram(x, fs=fs, nperseg=nperseg, **kw)

    def parameter_sweep(self, names, n_ffts, hops, processed=False, window="hann", reduce=None, share_hops=True):
        """BASELINE cfg4: every (n_fft, hop) pair over the named clips.

        Returns ``{(n_fft, hop): (f, t, result)}``; ``reduce`` maps a ``DeviceSpectrogram`` to what should be
        copied back (default: the full ``[clip, freq, time]`` PSD).

        ``share_hops`` (default on, full-PSD form only): hops of one n_fft that divide each other are served by ONE transform
        and ONE download at their gcd (``spectro.sweep.hop_families``); the coarser hops come back as strided views
        ``[..., ::h//g]`` of that array -- the same numbers bit for bit (frame i at hop h is frame i*h/g at hop g), 1 unit of
        device work and PCIe traffic per n_fft instead of 1 + 1/2 + 1/4 for cfg4's hops.  Copy a result before writing to it."""
        from spectro import _capi, engine
        from spectro.dist import n_frames
        from spectro.sweep import hop_families
        x, fs = self._stack(names, processed)
        clips = engine.DeviceClips(x)                  # the clips cross PCIe once for all pairs
        out = {}
        try:
            for n in n_ffts:
                families = hop_families(hops) if (share_hops and reduce is None) else [(int(h), [int(h)]) for h in hops]
                for g, fam in families:
                    dev = clips.stft(fs=fs, window=window, nperseg=n, hop=g)
                    nps = min(int(n), x.shape[-1])              # resolve_segments clamps nperseg to the clip length
                    try:
                        if reduce is not None:
                            out[(n, g)] = (dev.f, dev.t, reduce(dev))
                            continue
                        full = dev.to_host()
                        for h in fam:
                            nfr = n_frames(x.shape[-1], nps, h)
                            t = dev.t if h == g else _capi.times(x.shape[-1], nps, h, fs)
                            out[(n, h)] = (dev.f, t, full if h == g else full[..., ::h // g][..., :nfr])
                    finally:
                        dev.free()
        finally:
            clips.free()
        return {(n, h): out[(n, int(h))] for n in n_ffts for h in hops}


def read_wav(path):
    """Minimal RIFF/WAVE decoder -> ``(fs, [channel arrays])``.  PCM 8/16/24/32-bit and IEEE float 32/64."""
    with open(path, "rb") as fh:
        blob = fh.read()
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(blob):
        tag, size = blob[pos:pos + 4], struct.unpack_from("<I", blob, pos + 4)[0]
        body = blob[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", body)
            if fmt[0] == 0xFFFE and len(body) >= 26:               # WAVE_FORMAT_EXTENSIBLE: real tag in the GUID
                fmt = (struct.unpack_from("<H", body, 24)[0],) + fmt[1:]
        elif tag == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    kind, n_ch, fs, _, _, bits = fmt
    if kind == 1 and bits == 8:
        flat = np.frombuffer(payload, np.uint8).astype(np.int16) - 128
    elif kind == 1 and bits == 16:
        flat = np.frombuffer(payload, "<i2")
    elif kind == 1 and bits == 24:
        b = np.frombuffer(payload[:len(payload) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        flat = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
        flat = np.where(flat & 0x800000, flat - (1 << 24), flat).astype(np.int32)
    elif kind == 1 and bits == 32:
        flat = np.frombuffer(payload, "<i4")
    elif kind == 3 and bits in (32, 64):
        flat = np.frombuffer(payload, "<f4" if bits == 32 else "<f8")
    else:
        raise ValueError(f"{path}: unsupported WAV encoding (format {kind}, {bits} bit)")
    flat = flat[:len(flat) // n_ch * n_ch].reshape(-1, n_ch)
    return float(fs), [np.ascontiguousarray(flat[:, c]) for c in range(n_ch)]
