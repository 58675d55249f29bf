"""GPU parity of the fused log display (sg_stft_db): PlotEngine.py:126-131 with a caller-supplied global_max, computed
inside the STFT kernel, against the oracle's plot_image on the same samples -- small shapes exhaustively, the full cfg2
batch (64 x 480 000 samples, 119 808 x 513 image) on sampled frames plus exact min/max bookkeeping."""
import ctypes as C

import numpy as np
import pytest

from conftest import cfg2_clips
from oracle import stft_oracle as orc

pytestmark = pytest.mark.gpu

FS = 48000.0


def _plan(hop, detrend="constant", window="hann"):
    from spectro import _capi
    from spectro.windows import get_window
    _capi.ensure_device()
    return _capi.Plan(1024, 1024, hop, get_window(window, 1024), _capi.DETREND[detrend], FS, _capi.SCALING["density"],
                      _capi.MODE["psd"], _capi.F32)


def _run_db(plan, x, k_lo, k_hi, gmax):
    """-> (db image [clips, frames, nb] f32, mm (2,) f32) through the C ABI"""
    from spectro import _capi
    x = np.ascontiguousarray(x, np.float32)
    n_clips, n = x.shape
    nfr = plan.n_frames(n)
    nb = k_hi - k_lo + 1
    xin = _capi.DeviceBuffer(max(x.nbytes, 4))
    xin.upload(x)
    out = _capi.DeviceBuffer(max(n_clips * nfr * nb * 4, 4))
    mm = _capi.DeviceBuffer(8)
    plan.stft_db(xin.ptr, n, n, n_clips, k_lo, k_hi, gmax, out.ptr, nfr * nb, mm.ptr)
    img = np.empty((n_clips, nfr, nb), np.float32)
    mmh = np.empty(2, np.float32)
    out.download(img)
    mm.download(mmh)
    _capi.stream_sync()
    return img, mmh, (xin, out, mm)


def _ref_db(x, hop, detrend, k_lo, k_hi, gmax, window="hann"):
    """oracle: dB image before the min-max rescale, [clips, frames, nb] f32, and the linear band it came from"""
    f, t, s = orc.spectrogram(x, fs=FS, window=window, nperseg=1024, noverlap=1024 - hop, detrend=detrend)
    band = np.moveaxis(s, -1, -2)[..., k_lo:k_hi + 1]                      # [clips, frames, nb]
    norm = np.clip(band / (np.float32(gmax) + np.float32(1e-20)), 0.0, 1.0)    # PlotEngine.py:126-127
    fmax = np.moveaxis(s, -1, -2).max(axis=-1, keepdims=True)                  # frame maximum over the whole spectrum
    return (10.0 * np.log10(norm + np.float32(1e-12))).astype(np.float32), band, fmax    # :129


def _check_db(img, ref_db, band, fmax, gmax):
    """strong bins: dB within 5e-3 dB; every bin: back in linear terms within 1e-4 of the frame maximum (the fp32 parity
    criterion of the spectrum itself -- a near-zero bin's dB amplifies the relative error every fp32 FFT has there)"""
    assert img.shape == ref_db.shape and img.dtype == np.float32
    strong = band >= 1e-3 * fmax
    assert np.max(np.abs(img[strong] - ref_db[strong])) <= 5e-3
    lin_got = 10.0 ** (img.astype(np.float64) / 10.0)
    lin_ref = 10.0 ** (ref_db.astype(np.float64) / 10.0)
    tol = 1e-4 * fmax / gmax + 5e-6 * lin_ref + 1e-13
    assert np.all(np.abs(lin_got - lin_ref) <= tol), float(np.max(np.abs(lin_got - lin_ref) / tol))


@pytest.mark.parametrize("hop,detrend,band", [(256, "constant", (0, 512)), (256, "constant", (3, 200)), (896, "constant", (0, 512)),
                                              (128, False, (0, 512)), (512, "constant", (256, 256)), (100, "constant", (0, 512)),
                                              (255, False, (500, 512)), (256, "constant", (0, 0))])
def test_stft_db_matches_oracle(hop, detrend, band):
    x = cfg2_clips(2)[:, :60000]
    x[1, 1000:3000] += np.sin(2 * np.pi * 1000.0 * np.arange(2000) / FS).astype(np.float32)   # a strong line
    plan = _plan(hop, detrend if detrend else "none")
    k_lo, k_hi = band
    f, t, s = orc.spectrogram(x, fs=FS, window="hann", nperseg=1024, noverlap=1024 - hop, detrend=detrend)
    gmax = float(0.5 * s[:, k_lo:k_hi + 1].max())            # below the maximum, so the clip at 1 acts
    img, mm, _keep = _run_db(plan, x, k_lo, k_hi, gmax)
    ref_db, bandv, fmax = _ref_db(x, hop, detrend, k_lo, k_hi, gmax)
    _check_db(img, ref_db, bandv, fmax, gmax)
    # the folded partials ARE the extrema of what the kernel wrote
    assert mm[0] == img.min() and mm[1] == img.max()
    if bandv.max() >= 1e-3 * fmax.max():
        assert abs(mm[1] - ref_db.max()) <= 5e-3 and mm[1] <= 1e-4      # clipped bins sit at 10*log10(1 + 1e-12)


def test_stft_db_rescale_and_colormap_equal_reference_display():
    """db -> (db - min)/(max - min) in place == oracle.plot_image(log_scale=True, global_max=g); the colour map with the
    rescale folded in equals sg_colormap of the rescaled image.

    The display's minimum is the dB value of the single weakest bin.  With global_max near the signal's own level that bin
    is a near-zero fp32 FFT output whose relative error is O(1) for ANY fp32 implementation (scipy's included), so the
    case pinned against plot_image end to end is the one where the reference's own +1e-12 floor conditions it
    (global_max from a much louder sweep, PlotEngine.py:110); for the ill-conditioned case the rescale is checked with the
    device's own extrema."""
    from spectro import _capi
    x = cfg2_clips(1)[0, :100000]
    hop = 256
    plan = _plan(hop)
    f, t, s = orc.spectrogram(x, fs=FS, window="hann", nperseg=1024, noverlap=768)
    lut = np.empty((256, 4), np.uint8)
    _capi.check(_capi.lib().sg_jet_lut(lut.ctypes.data_as(C.POINTER(C.c_uint8))))
    lutd = _capi.DeviceBuffer(1024)
    lutd.upload(lut)
    for gmax, conditioned in ((float(s.max()) * 1e8, True), (float(s.max()), False)):
        lf, lt, lsxx, ref_img = orc.plot_image(f, t, s, 0.0, FS, True, global_max=gmax)
        img, mm, (xin, out, mmd) = _run_db(plan, x[None], 0, 512, gmax)
        n = img.size
        rgba1, rgba2 = _capi.DeviceBuffer(n * 4), _capi.DeviceBuffer(n * 4)
        _capi.check(_capi.lib().sg_colormap_db(C.c_void_p(out.ptr), n, C.c_void_p(mmd.ptr), C.c_void_p(lutd.ptr), C.c_void_p(rgba1.ptr), None))
        _capi.check(_capi.lib().sg_db_rescale(C.c_void_p(out.ptr), n, C.c_void_p(mmd.ptr), None))
        _capi.check(_capi.lib().sg_colormap(C.c_void_p(out.ptr), n, C.c_void_p(lutd.ptr), C.c_void_p(rgba2.ptr), None))
        got = np.empty((img.shape[1], 513), np.float32)
        out.download(got)
        c1, c2 = np.empty((n, 4), np.uint8), np.empty((n, 4), np.uint8)
        rgba1.download(c1)
        rgba2.download(c2)
        _capi.stream_sync()
        assert got.min() == 0.0 and got.max() == 1.0
        np.testing.assert_array_equal(c1, c2)
        strong = lsxx.T >= 1e-3 * lsxx.T.max(axis=1, keepdims=True)
        ref_db, _band, _fm = _ref_db(x[None], hop, "constant", 0, 512, gmax)
        if conditioned:
            # against the reference display itself: [bins, frames] -> [frames, bins]
            assert abs(float(mm[0]) - float(ref_db.min())) <= 5e-3 and abs(float(mm[1]) - float(ref_db.max())) <= 5e-3
            assert np.max(np.abs(got[strong] - ref_img.T[strong])) <= 1e-4
            assert np.max(np.abs(got - ref_img.T)) <= 2e-3      # weak bins sit near the -120 dB floor
        else:
            expect = (ref_db[0].astype(np.float64) - float(mm[0])) / (float(mm[1]) - float(mm[0]))
            assert np.max(np.abs(got[strong] - expect[strong])) <= 1e-4
            assert abs(float(mm[1]) - float(ref_db.max())) <= 5e-3
            assert float(mm[0]) <= float(ref_db.max()) - 40.0          # a noise spectrum spans far more than 40 dB


def test_stft_db_full_cfg2_size():
    """BASELINE cfg2 (64 clips x 10 s x 48 kHz, 119 808 frames x 513 bins): sampled frames against the oracle, min/max
    partials against numpy over the whole downloaded image, clip independence of the batch."""
    x = cfg2_clips(64)
    plan = _plan(256)
    gmax = 2.5e-6          # ~ 6 x the batch's mean PSD level (0.01 / 24 kHz): a fraction of a percent of the bins clip
    img, mm, _keep = _run_db(plan, x, 0, 512, gmax)
    assert img.shape == (64, 1872, 513)
    assert mm[0] == img.min() and mm[1] == img.max()
    assert np.isfinite(img).all() and mm[0] >= -120.0 - 1e-3 and mm[1] <= 1e-4
    rng = np.random.default_rng(7)
    for c, fr in zip(rng.integers(0, 64, 24), rng.integers(0, 1872, 24)):
        seg = x[c, fr * 256: fr * 256 + 1024]
        ref_db, band, fm = _ref_db(seg[None], 256, "constant", 0, 512, gmax)
        _check_db(img[c, fr][None, None], ref_db, band, fm, gmax)
    # first / last frame of a clip and the clip boundary are where indexing errors would show
    for c, fr in ((0, 0), (0, 1871), (1, 0), (63, 1871), (31, 935)):
        seg = x[c, fr * 256: fr * 256 + 1024]
        ref_db, band, fm = _ref_db(seg[None], 256, "constant", 0, 512, gmax)
        _check_db(img[c, fr][None, None], ref_db, band, fm, gmax)


def test_stft_db_argument_errors():
    from spectro import _capi
    from spectro.windows import get_window
    plan = _plan(256)
    buf = _capi.DeviceBuffer(1 << 16)
    with pytest.raises(ValueError):
        plan.stft_db(buf.ptr, 4096, 4096, 1, 0, 512, 0.0, buf.ptr, 0, buf.ptr)          # global_max must be > 0
    with pytest.raises(ValueError):
        plan.stft_db(buf.ptr, 4096, 4096, 1, 5, 513, 1.0, buf.ptr, 0, buf.ptr)          # band outside the spectrum
    with pytest.raises(ValueError):
        plan.stft_db(buf.ptr, 4096, 4096, 1, 0, 512, 1.0, buf.ptr, 0, None)             # null min/max pointer
    other = _capi.Plan(512, 512, 128, get_window("hann", 512), 1, FS, 0, 0, _capi.F32)
    with pytest.raises(NotImplementedError):
        other.stft_db(buf.ptr, 4096, 4096, 1, 0, 256, 1.0, buf.ptr, 0, buf.ptr)         # only the nfft-1024 register kernel
    # shorter than one frame: nothing written, (min, max) = (+inf, -inf)
    mm = np.empty(2, np.float32)
    plan.stft_db(buf.ptr, 100, 100, 1, 0, 512, 1.0, buf.ptr, 0, buf.ptr)
    buf.download(mm, nbytes=8)
    _capi.stream_sync()
    assert mm[0] == np.inf and mm[1] == -np.inf
