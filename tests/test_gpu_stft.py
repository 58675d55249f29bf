"""GPU parity tests proper: HIP path (through the C ABI / ctypes) vs the CPU oracle and the
committed golden vectors.  Run on the MI355X box with ``pytest -m gpu``.

Tolerance (fp32 path, BASELINE.md §2 / SURVEY H2), written out in conftest.assert_spec_close:
  per frame |gpu - ref| <= 1e-4 * max_k ref[k];  normwise ||gpu-ref||/||ref|| <= 1e-5;
  per-bin rtol 1e-4 for bins >= 1e-3 * frame max.
f64 path: 1e-11 relative to the frame max.  Frame indexing, f and t: bit-exact.
"""
import ctypes as C
import json
import warnings

import numpy as np
import pytest

from conftest import assert_spec_close, cfg1_signal, cfg2_clips, eeg_like, load_golden, sweep_clip
from oracle import stft_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sp():
    import spectro
    from spectro import _capi
    _capi.ensure_device()
    return spectro


def _is_pow2(n):
    return n > 0 and (n & (n - 1)) == 0


def _check(got, ref, dtype):
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert got.dtype == ref.dtype, (got.dtype, ref.dtype)
    if dtype == np.float64:
        g, r = np.asarray(got), np.asarray(ref)
        if r.size:
            fmax = np.abs(r).max(axis=-2, keepdims=True) if r.ndim > 1 else np.abs(r).max()
            assert np.all(np.abs(g - r) <= 1e-11 * fmax + 1e-300), float(np.max(np.abs(g - r) / (fmax + 1e-300)))
    else:
        assert_spec_close(got, ref, time_axis=-1)


# ---------------------------------------------------------------- headline kernel (r8x3)
@pytest.mark.parametrize("tag", ["ext", "ref"])
def test_cfg2_golden_and_oracle(sp, tag):
    g = load_golden("g3_cfg2_sampled.npz")
    clips = cfg2_clips(2)
    kw = dict(window="hann", noverlap=768) if tag == "ext" else {}
    f, t, s = sp.spectrogram(clips, fs=48000.0, nperseg=1024, scaling="density", mode="psd", **kw)
    np.testing.assert_array_equal(f, g[f"{tag}__f"])          # bit-exact
    np.testing.assert_array_equal(t, g[f"{tag}__t"])          # bit-exact
    assert s.dtype == np.float32 and s.shape == (2, 513, 1872 if tag == "ext" else 535)
    idx = g[f"{tag}__frame_idx"]
    assert_spec_close(np.moveaxis(s[:, :, idx], -1, 1), g[f"{tag}__frames"], time_axis=1)
    assert np.allclose(s.astype(np.float64).sum(axis=1), g[f"{tag}__frame_sums"], rtol=2e-5)
    # every frame against the oracle
    _, _, so = orc.spectrogram(clips, fs=48000.0, nperseg=1024, **kw)
    assert_spec_close(s, so, time_axis=-1)


def test_r8x3_is_the_kernel_and_matches_stockham(sp):
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((3, 30000)) * 0.3 + 0.7).astype(np.float32)
    win = get_window("hann", 1024)
    plan = plan_for(win, 1024, 1024, 256, 1, 48000.0, 0, 0, _capi.F32)
    assert plan.kernel == "r8x3"
    nfr, nb = plan.n_frames(30000), 513
    d_in, d_a, d_b = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(3 * nfr * nb * 4), _capi.DeviceBuffer(3 * nfr * nb * 4)
    d_in.upload(x)
    plan.stft(d_in.ptr, 30000, 30000, 3, d_a.ptr, nfr * nb)
    plan.force_kernel("stockham")
    try:
        assert plan.kernel == "stockham"
        plan.stft(d_in.ptr, 30000, 30000, 3, d_b.ptr, nfr * nb)
    finally:
        plan.force_kernel("r8x3")
    a, b = np.empty((3, nfr, nb), np.float32), np.empty((3, nfr, nb), np.float32)
    d_a.download(a)
    d_b.download(b)
    _capi.stream_sync()
    assert_spec_close(a, b, time_axis=1)
    _, _, so = orc.spectrogram(x, fs=48000.0, nperseg=1024, window="hann", noverlap=768)
    assert_spec_close(np.moveaxis(a, 1, 2), so, time_axis=-1)


@pytest.mark.parametrize("hop,detrend,mode", [(255, "constant", "psd"), (1, "constant", "psd"), (1024, False, "psd"),
                                              (896, "constant", "magnitude"), (333, False, "magnitude")])
def test_r8x3_variants(sp, hop, detrend, mode):
    rng = np.random.default_rng(hop)
    n = 1024 + hop * 37 + 5
    x = (rng.standard_normal((2, n)) + 2.0).astype(np.float32)
    kw = dict(fs=1000.0, nperseg=1024, window=("tukey", 0.25), noverlap=1024 - hop, detrend=detrend, mode=mode)
    f, t, s = sp.spectrogram(x, **kw)
    fo, to, so = orc.spectrogram(x, **kw)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    assert_spec_close(s, so, time_axis=-1)


@pytest.mark.parametrize("nperseg,hop", [(32, 8), (64, 16), (96, 24), (224, 56), (128, 32), (256, 64), (512, 128), (512, 127), (1024, 256), (2048, 128), (4096, 1024), (2048, 333), (1000, 250), (4000, 1000)])
def test_int16_pcm_batches_every_family(sp, nperseg, hop):
    """int16 PCM (16-bit WAV): the result must equal the float call on the same values BIT FOR BIT in every kernel family -- r8x3
    and the LDS kernel load int16 themselves, rsmall / rbig batches convert once into a stream-ordered workspace
    (sg_stft_i16 -> sg_convert_i16 + the float kernel), small calls and odd hops stay on the LDS kernel -- and match the oracle."""
    from spectro import engine
    rng = np.random.default_rng(nperseg + hop)
    for n_clips, n in ((1, nperseg + hop * 20 + 3), (5, 70001)):          # a GUI-sized call and a batch above the conversion threshold
        xi = (rng.standard_normal((n_clips, n)) * 5000).astype(np.int16)
        kw = dict(fs=44100.0, nperseg=nperseg, window="hann", noverlap=nperseg - hop)
        f, t, s_i = sp.spectrogram(xi, **kw)
        _, _, s_f = sp.spectrogram(xi.astype(np.float32), **kw)
        assert s_i.dtype == np.float32
        same_kernel = nperseg == 1024 or (n_clips * n >= (1 << 18) and hop % 2 == 0)
        if same_kernel:
            np.testing.assert_array_equal(s_i, s_f)
        else:                                                             # small call / odd hop: int16 runs the LDS kernel
            assert_spec_close(s_i, s_f, time_axis=-1)
        _, _, so = orc.spectrogram(xi, **kw)
        assert_spec_close(s_i, so, time_axis=-1)
    dc = engine.DeviceClips(xi)                                           # int16 clips resident: the float copy is made on the device
    try:
        dev = dc.stft(fs=44100.0, window="hann", nperseg=nperseg, hop=hop)
        if hop % 2 == 0 or nperseg in (1024, 1000):                       # the float copy runs the float call's kernel
            np.testing.assert_array_equal(dev.to_host(), s_f)
        else:
            assert_spec_close(dev.to_host(), s_f, time_axis=-1)
        dev.free()
    finally:
        dc.free()


def test_int16_batches_many_calls_in_a_row(sp):
    """Back-to-back int16 batches of changing size on the plans that convert into the stream's workspace: each must equal the
    float call bit for bit.  (The first version of that path took its workspace from hipMallocAsync and returned foreign data in
    ~8 % of such calls on this ROCm build -- tools/repro_i16.py; only a many-call loop shows it.)"""
    rng = np.random.default_rng(5)
    for it in range(80):
        n_clips, n = int(rng.choice([33, 70, 16])), int(rng.integers(4000, 30000))
        nper = int(rng.choice([256, 512, 2048, 4096]))
        hop = int(rng.choice([64, 128, 256, nper - nper // 8]))
        n = max(n, nper + hop)
        kw = dict(fs=500.0, nperseg=nper, noverlap=nper - hop, window="hann", detrend=False)
        x = ((rng.standard_normal((n_clips, n)) * 1.3 + 0.6) * 3000).astype(np.int16)
        _, _, s = sp.spectrogram(x, **kw)
        _, _, sf = sp.spectrogram(x.astype(np.float32), **kw)
        if n_clips * n >= (1 << 18):
            np.testing.assert_array_equal(s, sf, err_msg=f"call {it}: {n_clips} x {n}, nperseg {nper}, hop {hop}")
        else:
            assert_spec_close(s, sf, time_axis=-1)


@pytest.mark.parametrize("nperseg,hop", [(1024, 64), (1024, 32), (1024, 16), (2048, 64), (2048, 128), (2048, 256), (2048, 32),
                                         (4096, 64), (4096, 128), (4096, 256), (4096, 16),
                                         (256, 128), (256, 64), (256, 16), (512, 128), (512, 64), (512, 32), (128, 32), (128, 16), (128, 128)])
def test_sliding_window_walks(sp, nperseg, hop):
    """Register sliding windows: hops 128 / 256 directly, hops 64 / 32 / 16 as 2 / 4 / 8 interleaved hop-128 sequences (r8x3, rbig
    and rsmall): every frame count from one frame up, several clips, spectrum + fused band power against the oracle / the written
    spectrum; at 1024 also the fused dB image."""
    from spectro import _capi, engine
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(hop + nperseg)
    for n_frames in (1, 2, 3, 4, 5, 7, 8, 9, 17, 38, 77, 301):
        n = nperseg + hop * (n_frames - 1) + 6
        x = (rng.standard_normal((3, n)) * 0.3 + 0.05).astype(np.float32)      # (a large DC offset is test_r8x3_variants' business)
        kw = dict(fs=48000.0, nperseg=nperseg, window="hann", noverlap=nperseg - hop)
        f, t, s = sp.spectrogram(x, **kw)
        fo, to, so = orc.spectrogram(x, **kw)
        assert s.shape[-1] == n_frames
        np.testing.assert_array_equal(t, to)
        assert_spec_close(s, so, time_axis=-1)
        plan = plan_for(get_window("hann", nperseg), nperseg, nperseg, hop, 1, 48000.0, 0, 0, _capi.F32)
        d_in, d_bp = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(3 * n_frames * 4)
        d_in.upload(x)
        _capi.check(_capi.lib().sg_memset(C.c_void_p(d_bp.ptr), 0xFF, 3 * n_frames * 4, None))
        k_hi = min(300, nperseg // 2)
        plan.band_power(d_in.ptr, n, n, 3, 5, k_hi, d_bp.ptr, n_frames)
        bp = np.empty((3, n_frames), np.float32)
        d_bp.download(bp)
        _capi.stream_sync()
        ref = s[:, 5:k_hi + 1, :].astype(np.float64).sum(axis=1)
        assert np.all(np.abs(bp - ref) <= 2e-6 * s.astype(np.float64).sum(axis=1) + 1e-30)
        d_in.free(); d_bp.free()
    if nperseg != 1024:
        return
    # fused dB image on the interleaved walk: the same image as the hop-128 rows of it
    x = (rng.standard_normal((2, 1024 + 128 * 40)) * 0.2).astype(np.float32)
    dc = engine.DeviceClips(x)
    try:
        fb, t_s, img_s = dc.log_image(48000.0, 1024, hop, 500.0, 9000.0, global_max=1e-4, window="hann", rescale=False)
        fb2, t_c, img_c = dc.log_image(48000.0, 1024, 128, 500.0, 9000.0, global_max=1e-4, window="hann", rescale=False)
        np.testing.assert_array_equal(img_s[..., ::128 // hop][..., :img_c.shape[-1]], img_c)
    finally:
        dc.free()


@pytest.mark.parametrize("hop", [64, 32, 16, 128, 256, 512])
def test_register_f64_small_and_block_hops(sp, hop):
    """The double-precision 1024 kernel at the hops where the f32 kernels slide their window: every frame count, band power."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(900 + hop)
    for n_frames in (1, 2, 3, 4, 5, 7, 8, 9, 17, 38, 77, 150):
        n = 1024 + hop * (n_frames - 1) + 6
        x = rng.standard_normal((3, n)) * 0.3 + 0.7
        kw = dict(fs=20000.0, nperseg=1024, window=("tukey", 0.25), noverlap=1024 - hop)
        f, t, s = sp.spectrogram(x, **kw)
        fo, to, so = orc.spectrogram(x, **kw)
        assert s.shape[-1] == n_frames
        np.testing.assert_array_equal(t, to)
        _check(s, so, np.float64)
        plan = plan_for(get_window(("tukey", 0.25), 1024), 1024, 1024, hop, 1, 20000.0, 0, 0, _capi.F64)
        d_in, d_bp = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(3 * n_frames * 8)
        d_in.upload(x)
        plan.band_power(d_in.ptr, n, n, 3, 5, 300, d_bp.ptr, n_frames)
        bp = np.empty((3, n_frames))
        d_bp.download(bp)
        _capi.stream_sync()
        ref = s[:, 5:301, :].sum(axis=1)
        assert np.all(np.abs(bp - ref) <= 1e-12 * 296 * s.max(axis=1) + 1e-300)
        d_in.free(); d_bp.free()


def test_r8x3_int16_input(sp):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((2, 20000)) * 4000).astype(np.int16)
    f, t, s = sp.spectrogram(x, fs=8000.0, nperseg=1024)
    _, _, so = orc.spectrogram(x, fs=8000.0, nperseg=1024)
    assert s.dtype == so.dtype == np.float32
    assert_spec_close(s, so, time_axis=-1)


@pytest.mark.parametrize("nperseg", [1024, 512, 256, 128])
@pytest.mark.parametrize("hop_of,window,detrend,mode", [
    (lambda n: n - n // 8, ("tukey", 0.25), "constant", "psd"),          # the reference's call
    (lambda n: n // 4, "hann", "constant", "psd"),
    (lambda n: 2, "hann", False, "psd"),
    (lambda n: n, "boxcar", False, "magnitude"),
    (lambda n: 2 * (n // 6), ("tukey", 0.25), "constant", "magnitude"),
    (lambda n: n // 4 - 1, "hann", "constant", "psd"),                     # odd hop: the call falls back to the LDS kernel
], ids=["ref", "quarter", "hop2", "nooverlap_mag", "third_mag", "odd"])
def test_register_f64_variants(sp, nperseg, hop_of, window, detrend, mode):
    """The reference's nperseg settings on its float64 recordings: the double-precision register kernels (stft_r8x3_f64.hip)."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    hop = hop_of(nperseg)
    rng = np.random.default_rng(1000 + hop + nperseg)
    n = nperseg + hop * 40 + 6                                # 41 frames: a partial last group at 256 / 512; even clip stride
    n += n % 2
    x = rng.standard_normal((3, n)) * 0.4 + 1.5
    kw = dict(fs=20000.0, nperseg=nperseg, window=window, noverlap=nperseg - hop, detrend=detrend, mode=mode)
    f, t, s = sp.spectrogram(x, **kw)
    fo, to, so = orc.spectrogram(x, **kw)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    _check(s, so, np.float64)
    plan = plan_for(get_window(window, nperseg), nperseg, nperseg, hop, _capi.DETREND[detrend], 20000.0, 0, _capi.MODE[mode], _capi.F64)
    assert plan.kernel == ("r8x3d" if nperseg == 1024 else "rsmalld")
    _, _, s1 = sp.spectrogram(x[1, 1:], **kw)
    _, _, so1 = orc.spectrogram(x[1, 1:], **kw)
    _check(s1, so1, np.float64)


@pytest.mark.parametrize("nperseg", [1024, 512, 256, 128])
def test_register_f64_matches_stockham_and_edges(sp, nperseg):
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(77)
    hop = nperseg - nperseg // 8
    ns = nperseg + hop * 10                                 # 11 frames per clip (a partial last group at 256 / 512), 7 clips
    x = rng.standard_normal((7, ns))
    x[2] = 0.0                                              # all-zero clip
    x[3] = 3.25                                             # constant clip: detrended to exact zeros
    family = "r8x3d" if nperseg == 1024 else "rsmalld"
    plan = plan_for(get_window(("tukey", 0.25), nperseg), nperseg, nperseg, hop, 1, 20000.0, 0, 0, _capi.F64)
    assert plan.kernel == family
    nfr, nb = plan.n_frames(ns), nperseg // 2 + 1
    d_in, d_a, d_b = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(7 * nfr * nb * 8), _capi.DeviceBuffer(7 * nfr * nb * 8)
    d_in.upload(x)
    plan.stft(d_in.ptr, ns, ns, 7, d_a.ptr, nfr * nb)
    plan.force_kernel("stockham")
    try:
        plan.stft(d_in.ptr, ns, ns, 7, d_b.ptr, nfr * nb)
    finally:
        plan.force_kernel(family)
    a, b = np.empty((7, nfr, nb)), np.empty((7, nfr, nb))
    d_a.download(a)
    d_b.download(b)
    _capi.stream_sync()
    _check(np.moveaxis(a, 1, 2), np.moveaxis(b, 1, 2), np.float64)
    assert np.all(a[2] == 0.0) and np.all(a[3] == 0.0)
    _, _, so = orc.spectrogram(x, fs=20000.0, nperseg=nperseg)
    _check(np.moveaxis(a, 1, 2), so, np.float64)
    # fused band power (A11) on the same kernel: per-frame sums of bins [k_lo, k_hi], the spectrum is never written
    d_bp = _capi.DeviceBuffer(7 * nfr * 8)
    h = nperseg // 2
    for k_lo, k_hi in [(0, h), (3, 40), (h // 2, h // 2), (h // 2 - 1, h // 2 + 1), (h // 2 + 1, h - 12), (0, 0), (h, h), (min(100, h - 2), h - 1)]:
        plan.band_power(d_in.ptr, ns, ns, 7, k_lo, k_hi, d_bp.ptr, nfr)
        bp = np.empty((7, nfr))
        d_bp.download(bp)
        _capi.stream_sync()
        ref_bp = a[:, :, k_lo:k_hi + 1].sum(-1)
        assert np.all(np.abs(bp - ref_bp) <= 1e-12 * np.abs(a).max(axis=-1) * (k_hi - k_lo + 1) + 1e-300), (k_lo, k_hi)
    # a clip that starts on an odd sample of the device buffer (8-byte aligned only): the call falls back to the LDS kernel
    plan.stft(d_in.ptr + 8, ns - 1, ns, 1, d_b.ptr, nfr * nb)
    b1 = np.empty((plan.n_frames(ns - 1), nb))
    d_b.download(b1)
    _capi.stream_sync()
    _, _, so1 = orc.spectrogram(x[0, 1:], fs=20000.0, nperseg=nperseg)
    _check(b1.T, so1, np.float64)
    with pytest.raises(NotImplementedError):
        plan_for(get_window("hann", 1024), 1024, 1024, 256, 1, 1.0, 0, 0, _capi.F32).force_kernel("r8x3d")
    with pytest.raises(NotImplementedError):
        plan.force_kernel("rsmalld" if nperseg == 1024 else "r8x3d")


@pytest.mark.parametrize("nperseg,hop,family", [(128, 32, "rsmall"), (256, 64, "rsmall"), (512, 448, "rsmall"), (1024, 256, "r8x3"), (2048, 128, "rbig"),
                                                (4096, 1024, "rbig"), (1024, 256, "stockham")])
def test_fused_band_power_every_f32_family(sp, nperseg, hop, family):
    """A11 fused into the transform (sg_stft_band_power): the per-frame band sum must equal the sum over the written spectrum's
    bins, for the register kernels (round 2: rsmall and rbig too -- the parameter sweep's reduced product no longer falls back
    to the LDS kernel) and for the LDS kernel itself."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(nperseg + hop)
    ns = nperseg + hop * 21 + 2                              # 22 frames: a partial last group for rsmall
    x = (rng.standard_normal((5, ns)) * 0.3 + 0.5).astype(np.float32)
    plan = plan_for(get_window("hann", nperseg), nperseg, nperseg, hop, 1, 48000.0, 0, 0, _capi.F32)
    restore = plan.kernel
    if family == "stockham":
        plan.force_kernel("stockham")
    try:
        assert plan.kernel == family
        nfr, nb = plan.n_frames(ns), nperseg // 2 + 1
        d_in, d_s, d_bp = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(5 * nfr * nb * 4), _capi.DeviceBuffer(5 * nfr * 4)
        d_in.upload(x)
        plan.stft(d_in.ptr, ns, ns, 5, d_s.ptr, nfr * nb)
        spec = np.empty((5, nfr, nb), np.float32)
        d_s.download(spec)
        _capi.stream_sync()
        h = nperseg // 2
        for k_lo, k_hi in [(0, h), (1, 7), (h // 2, h // 2), (h // 2 - 1, h // 2 + 1), (h // 2 + 1, h - 3), (0, 0), (h, h), (min(65, h - 2), h - 1)]:
            _capi.check(_capi.lib().sg_memset(C.c_void_p(d_bp.ptr), 0xFF, 5 * nfr * 4, None))
            plan.band_power(d_in.ptr, ns, ns, 5, k_lo, k_hi, d_bp.ptr, nfr)
            bp = np.empty((5, nfr), np.float32)
            d_bp.download(bp)
            _capi.stream_sync()
            ref = spec[:, :, k_lo:k_hi + 1].astype(np.float64).sum(-1)
            assert np.all(np.abs(bp - ref) <= 2e-6 * spec.astype(np.float64).sum(-1) + 1e-30), (k_lo, k_hi)
    finally:
        plan.force_kernel(restore)


@pytest.mark.parametrize("dt", ["float32", "float64"])
@pytest.mark.parametrize("nperseg", [256, 512, 1024, 2048, 4096, 1000])
def test_coarser_hops_are_row_subsets_of_the_finest(sp, dt, nperseg):
    """Frame i at hop h is frame i*h/g at hop g (same samples, same arithmetic): what sweep.hop_families relies on, for every
    kernel family -- including the r8x3 variants that slide their window in registers at some hops and reload at others."""
    rng = np.random.default_rng(nperseg)
    x = (rng.standard_normal((3, nperseg + 64 * 57 + 10)) * 0.3 + 0.2).astype(dt)
    kw = dict(fs=48000.0, nperseg=nperseg, window="hann")
    _, t64, s64 = sp.spectrogram(x, noverlap=nperseg - 64, **kw)
    for hop in (128, 256, 192):
        _, t, s = sp.spectrogram(x, noverlap=nperseg - hop, **kw)
        np.testing.assert_array_equal(s, s64[..., ::hop // 64][..., :s.shape[-1]])
        np.testing.assert_array_equal(t, t64[::hop // 64][:t.size])


def test_full_size_properties(sp):
    """BASELINE cfg2 at full size (64 x 480000): size-independent properties.

    (1) Parseval per frame: sum_k Sxx[k] * fs * sum(w^2) / n  ==  sum_n ((x-mean)*w)^2   (one-sided doubling
        makes the PSD bins sum to the full two-sided energy);
    (2) linearity in amplitude: Sxx(a*x) == a^2 * Sxx(x);
    (3) clip independence: clip c of the batch equals the same clip run alone.
    """
    from spectro.windows import get_window
    x = np.random.default_rng(1234).standard_normal((64, 480000)).astype(np.float32) * np.float32(0.1)
    f, t, s = sp.spectrogram(x, fs=48000.0, nperseg=1024, window="hann", noverlap=768)
    assert s.shape == (64, 513, 1872)
    w = get_window("hann", 1024)
    for c, fr in [(0, 0), (17, 911), (63, 1871)]:
        seg = x[c, fr * 256: fr * 256 + 1024].astype(np.float64)
        seg = (seg - seg.mean()) * w
        lhs = s[c, :, fr].astype(np.float64).sum() * 48000.0 * (w * w).sum() / 1024
        assert abs(lhs - (seg ** 2).sum()) <= 2e-5 * (seg ** 2).sum()
    _, _, s2 = sp.spectrogram(x[:4] * np.float32(4.0), fs=48000.0, nperseg=1024, window="hann", noverlap=768)
    assert_spec_close(s2, s[:4] * np.float32(16.0), time_axis=-1)
    _, _, s1 = sp.spectrogram(x[40], fs=48000.0, nperseg=1024, window="hann", noverlap=768)
    np.testing.assert_array_equal(s1, s[40])


# ---------------------------------------------------------------- general kernel
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
def test_cfg1_golden(sp, tag, dt):
    g = load_golden("g2_cfg1_extended.npz")
    x = cfg1_signal().astype(dt)
    f, t, s = sp.spectrogram(x, fs=16000.0, **{"scaling": "density", "mode": "psd", **G2_KW[tag]})
    key = f"{tag}_{dt}"
    np.testing.assert_array_equal(f, g[key + "__f"])
    np.testing.assert_array_equal(t, g[key + "__t"])
    ref = g[key + "__Sxx"]
    if tag == "hann256_linear" and dt == "float64":
        assert s.shape == ref.shape and np.abs(s - ref).max() <= 1e-9 * np.abs(ref).max()
    else:
        _check(s, ref, np.dtype(dt))


def test_angle_and_phase_modes(sp):
    x = cfg1_signal()[:6000]
    for mode in ("angle", "phase"):
        f, t, s = sp.spectrogram(x, fs=16000.0, nperseg=256, window="hann", noverlap=128, mode=mode)
        _, _, so = orc.spectrogram(x, fs=16000.0, nperseg=256, window="hann", noverlap=128, mode=mode)
        _, _, mag = orc.spectrogram(x, fs=16000.0, nperseg=256, window="hann", noverlap=128, mode="magnitude")
        assert s.shape == so.shape
        strong = mag > 1e-3 * mag.max()
        if mode == "angle":
            d = np.angle(np.exp(1j * (s - so)))
            assert np.abs(d[strong]).max() < 1e-9
        else:
            # 'phase' = unwrap(angle) along frequency (scipy:1003): a weak bin whose angle is numerically arbitrary can move
            # every later bin of that frame by a multiple of 2*pi, so compare modulo 2*pi on strong bins ...
            d = np.angle(np.exp(1j * (s - so)))
            assert np.abs(d[strong]).max() < 1e-9
            # ... the unwrap invariant on every bin (no jump above pi along frequency) ...
            assert np.abs(np.diff(s, axis=0)).max() <= np.pi + 1e-9
            # ... and equality of the VALUES on the frames where the unwrap has no close calls: every bin strong, and no
            # neighbouring pair of wrapped angles within 1e-6 of a +-pi step (where a 1e-12 error flips a 2*pi decision)
            _, _, ang = orc.spectrogram(x, fs=16000.0, nperseg=256, window="hann", noverlap=128, mode="angle")
            step = np.abs(np.diff(ang, axis=0))
            solid = (mag > 1e-6 * mag.max()).all(axis=0) & (np.abs(step - np.pi) > 1e-6).all(axis=0)
            assert solid.sum() >= 3
            assert np.abs(s[:, solid] - so[:, solid]).max() < 1e-8


def test_sweep_golden(sp):
    g = load_golden("g4_sweep.npz")
    x = sweep_clip()
    for n in (256, 512, 1024, 2048, 4096):
        for hop in (64, 128, 256):
            k = f"n{n}_h{hop}"
            f, t, s = sp.spectrogram(x, fs=48000.0, nperseg=n, window="hann", noverlap=n - hop)
            assert s.shape[-1] == int(g[k + "__nframes"])
            np.testing.assert_array_equal([t[0], t[-1]], g[k + "__t_ends"])
            assert_spec_close(s[:, g[k + "__frame_idx"]].T, g[k + "__frames"], time_axis=0)
            assert np.allclose(s.astype(np.float64).sum(axis=0), g[k + "__frame_sums"], rtol=2e-5)


G5_TAGS = ["short", "exact", "exact_plus", "two", "odd33", "np2_1000", "np2_96", "int16", "zeros", "const",
           "dc_large", "n8192", "n32", "hop1", "default_nperseg", "batch2d"]


@pytest.mark.parametrize("tag", G5_TAGS)
def test_edges_golden(sp, tag):
    g = load_golden("g5_edges.npz")
    kw = json.loads(str(g[tag + "__kw"]))
    fs = kw.pop("fs")
    x = g[tag + "__x"]
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        f, t, s = sp.spectrogram(x, fs=fs, **kw)
    assert int(any("nperseg" in str(w.message) for w in wl)) == int(g[tag + "__warned"])
    np.testing.assert_array_equal(f, g[tag + "__f"])
    np.testing.assert_array_equal(t, g[tag + "__t"])
    ref = g[tag + "__Sxx"]
    assert s.shape == ref.shape and s.dtype == ref.dtype
    if tag in ("const", "zeros"):
        assert np.abs(s - ref).max() <= 1e-6 * max(float(np.abs(x).max()) ** 2, 1e-12)
    elif tag == "dc_large":
        # mean 100, signal 1e-3 in fp32: the samples themselves carry ~1% quantisation and scipy's own fp32 mean is
        # off by a few ulp, so scipy-f32 is not the truth here.  Judge both against the f64 result on the same samples:
        # the device path must be at least as close to it as scipy's fp32 path (x2 slack) or within 1e-4 of frame max.
        _, _, truth = orc.spectrogram(x.astype(np.float64), fs=fs, **kw)
        err_ref = np.abs(ref.astype(np.float64) - truth).max()
        err_gpu = np.abs(s.astype(np.float64) - truth).max()
        assert err_gpu <= max(2.0 * err_ref, 1e-4 * truth.max()), (err_gpu, err_ref)
    else:
        _check(s, ref, ref.dtype)


@pytest.mark.parametrize("nperseg,dtype", [(4100, np.float64), (6000, np.float64), (8191, np.float64), (8192, np.float64),
                                           (16384, np.float64), (12000, np.float32), (8190, np.float32)])
def test_gui_range_large_nperseg(sp, nperseg, dtype):
    """The GUI's nperseg spin box goes to 8192 (GUI.py:87-89) and neo / H5 signals are float64: non-power-of-two f64 sizes
    above 4096 need a 16384-point chirp-z convolution (256 KiB) that no longer fits the LDS and runs in the HBM workspace;
    so do powers of two beyond the Stockham kernel.  Reference call (Tukey, hop = n - n//8) on an ephys-like trace."""
    rng = np.random.default_rng(nperseg)
    n_samples = 3 * nperseg + 777
    x = (np.cumsum(rng.standard_normal(n_samples)) * 0.01 + 0.2 * rng.standard_normal(n_samples) + 3.0).astype(dtype)
    f, t, s = sp.spectrogram(x, fs=1000.0, nperseg=nperseg, scaling="density", mode="psd")
    fo, to, so = orc.spectrogram(x, fs=1000.0, nperseg=nperseg, scaling="density", mode="psd")
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    assert s.shape == so.shape == (nperseg // 2 + 1, 3) and s.dtype == so.dtype
    if dtype == np.float64:
        assert np.abs(s - so).max() <= 1e-11 * np.abs(so).max()
    else:
        assert_spec_close(s, so, time_axis=-1)
    # and the two paths of the engine that sit on it
    from PlotEngine import PlotEngine
    eng = PlotEngine()
    settings = {"nperseg": nperseg, "fmin": 0.0, "fmax": 30.0, "log_scale": True, "mode_raw": "Spectrogram", "mode_proc": "None",
                "draw_raw": False, "draw_proc": False}
    eng.plot_extra(x, None, 1000.0, settings)
    m = (fo >= 0.0) & (fo <= 30.0)
    assert eng.last_Sxx.shape == so[m].shape
    tf, feats = eng._calculate_features(x, 1000.0, settings)
    lp = np.log10(so[m].sum(axis=0) + 1e-20)
    assert np.allclose(feats[:, 0], lp, atol=1e-9 if dtype == np.float64 else 2e-5)


def test_concurrent_callers_share_the_staging_and_the_pools(sp):
    """The library is re-entrant per (plan, stream) and the shim keeps process-wide caches (plans, windows, pinned staging for
    the zero-copy path, device and pinned pools): eight threads hammering small and medium calls must all get their own,
    correct results."""
    import threading
    rng = np.random.default_rng(99)
    jobs = []
    for i in range(8):
        n = (256, 512, 1024, 100)[i % 4]
        x = (rng.standard_normal(3000 + 997 * i) * (1 + i)).astype(np.float32 if i % 2 else np.float64)
        jobs.append((x, dict(fs=1000.0 + i, nperseg=n)))
    refs = [orc.spectrogram(x, **kw) for x, kw in jobs]
    errors = []

    def worker(idx):
        try:
            x, kw = jobs[idx]
            for _ in range(40):
                f, t, s = sp.spectrogram(x, **kw)
                fo, to, so = refs[idx]
                assert np.array_equal(f, fo) and np.array_equal(t, to) and s.shape == so.shape
                if x.dtype == np.float64:
                    assert np.abs(s - so).max() <= 1e-11 * np.abs(so).max()
                else:
                    assert_spec_close(s, so, time_axis=-1)
        except Exception as e:                      # noqa: BLE001 - reported below
            errors.append((idx, repr(e)))
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_concurrent_batches_share_the_stream_scratch(sp):
    """Host threads on the SAME (default) stream: int16 batches go through the stream's float workspace (convert, then transform)
    and the display epilogue through the stream's reduction scratch (partials, then fold) -- each such sequence must reach the
    stream back to back, whoever else is submitting (launch_sequence_mutex)."""
    import threading
    from spectro import engine
    rng = np.random.default_rng(123)
    jobs = []
    for i in range(6):
        nper = (256, 512, 2048, 4096, 512, 2048)[i]
        x = ((rng.standard_normal((20 + i, 15000 + 1111 * i)) * 0.7 + 0.2) * 3000).astype(np.int16)
        kw = dict(fs=8000.0, nperseg=nper, window="hann", noverlap=nper - nper // 4)
        jobs.append((x, kw, sp.spectrogram(x.astype(np.float32), **kw)[2]))
    xf = (rng.standard_normal((4, 60000)) * 0.3).astype(np.float32)
    dev = engine.stft(xf, fs=8000.0, nperseg=512)
    img_ref = dev.image(3, 200, True, None)
    errors = []

    def batch_worker(idx):
        try:
            x, kw, ref = jobs[idx]
            for _ in range(12):
                np.testing.assert_array_equal(sp.spectrogram(x, **kw)[2], ref)
        except Exception as e:                      # noqa: BLE001 - reported below
            errors.append((idx, repr(e)[:300]))

    def image_worker():
        try:
            for _ in range(60):
                np.testing.assert_array_equal(dev.image(3, 200, True, None), img_ref)
        except Exception as e:                      # noqa: BLE001
            errors.append(("image", repr(e)[:300]))
    threads = [threading.Thread(target=batch_worker, args=(i,)) for i in range(6)] + [threading.Thread(target=image_worker) for _ in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dev.free()
    assert not errors, errors


def test_axis_argument(sp):
    rng = np.random.default_rng(8)
    x = rng.standard_normal((700, 3)).astype(np.float32)
    import scipy.signal as ss
    f, t, s = sp.spectrogram(x, fs=100.0, nperseg=64, axis=0)
    fr, tr, sr = ss.spectrogram(x, fs=100.0, nperseg=64, axis=0)
    assert s.shape == sr.shape
    assert_spec_close(s, sr, time_axis=-1)


def test_argument_errors(sp):
    x = np.zeros(1000, np.float32)
    with pytest.raises(ValueError):
        sp.spectrogram(x, nperseg=64, noverlap=64)
    with pytest.raises(ValueError):
        sp.spectrogram(x, nperseg=64, nfft=32)
    with pytest.raises(ValueError):
        sp.spectrogram(x, nperseg=64, mode="nope")
    with pytest.raises(ValueError):
        sp.spectrogram(x, nperseg=64, scaling="nope")
    with pytest.raises(ValueError):
        sp.spectrogram(x, window=np.ones(2000))
    with pytest.raises(NotImplementedError):
        sp.spectrogram(x.astype(np.complex64), nperseg=64)


@pytest.mark.parametrize("n,hop,detrend,mode", [(n, h, d, m) for n in (256, 512) for h, d, m in
                                                [(64, "constant", "psd"), (128, "constant", "psd"), (256, False, "psd"), (2, "constant", "magnitude"), (None, "constant", "psd")]]
                         + [(128, 32, "constant", "psd"), (128, 64, "constant", "psd"), (128, 128, False, "psd"), (128, 2, "constant", "magnitude"),
                            (128, None, "constant", "psd")])
def test_rsmall_kernel(sp, n, hop, detrend, mode):
    """Register kernel for nfft 128 / 256 / 512 (G = 8 / 4 / 2 frames per wave; 128 since round 4): vs oracle and vs the Stockham kernel,
    incl. a frame count that is not a multiple of the group size and several clips."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    hop = n - n // 8 if hop is None else hop
    rng = np.random.default_rng(n + hop)
    N = n + hop * 41 + 3
    x = (rng.standard_normal((3, N)) * 0.4 + 1.5).astype(np.float32)
    kw = dict(fs=48000.0, nperseg=n, window="hann", noverlap=n - hop, detrend=detrend, mode=mode)
    f, t, s = sp.spectrogram(x, **kw)
    fo, to, so = orc.spectrogram(x, **kw)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    assert_spec_close(s, so, time_axis=-1, bin_floor=1e-3 if mode == "psd" else 1e-3 ** 0.5)
    plan = plan_for(get_window("hann", n), n, n, hop, _capi.DETREND[detrend], 48000.0, 0, _capi.MODE[mode], _capi.F32)
    assert plan.kernel == "rsmall"
    # odd hop falls back to the Stockham kernel inside the same plan
    y = (x[0] - np.float32(1.5))
    f2, t2, s2 = sp.spectrogram(y, fs=48000.0, nperseg=n, window="hann", noverlap=n - 33)
    _, _, so2 = orc.spectrogram(y, fs=48000.0, nperseg=n, window="hann", noverlap=n - 33)
    assert_spec_close(s2, so2, time_axis=-1)


_RTINY_POW2 = [(64, 16, "constant", "psd"), (64, 2, "constant", "magnitude"), (64, 64, False, "psd"), (64, None, "constant", "psd"),
               (32, 8, "constant", "psd"), (32, 2, False, "magnitude"), (32, 32, "constant", "psd"), (32, None, "constant", "psd")]
_RTINY_NP2 = [(96, 24, "constant", "psd"), (96, None, "constant", "psd"), (96, 2, False, "magnitude"), (160, 40, "constant", "psd"), (160, 160, False, "psd"),
              (192, 48, "constant", "magnitude"), (192, None, "constant", "psd"), (224, 56, "constant", "psd"), (224, None, "constant", "psd"), (224, 2, False, "psd")]


@pytest.mark.parametrize("dt,n,hop,detrend,mode", [(dt, *c) for dt in ("float32", "float64") for c in _RTINY_POW2 + _RTINY_NP2])
def test_rtiny_kernel(sp, dt, n, hop, detrend, mode):
    """The smallest sizes of the spin box (GUI.py:87-89: 32, 64): the quad-DPP register kernel (stft_rtiny.hip, round 4: 16 / 32 frames per
    wave, no LDS in the transform) in f32 and f64 -- and its form for nperseg 96 / 160 / 192 / 224 (6 ... 14 lanes per frame, the cross-lane
    DFT as a direct sum through LDS) -- vs the oracle and vs the Stockham / LDS chirp-z kernel of the same plan, frame counts that are no
    multiple of the group size, several clips incl. all-zero and constant ones, the fused band power, int16 and the odd-hop fallback."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    hop = n - n // 8 if hop is None else hop
    rng = np.random.default_rng(n + hop)
    frames = 83 if hop > 2 else 301
    N = n + hop * (frames - 1) + (2 if hop > 2 else 0)       # (even: clips at an even stride)
    # (DC offset 3.75 sigma at the powers of two, whose mean is exact; 0.5 sigma at the others: after the detrend bin 0 is a cancellation bin, and with a
    # large offset this kernel and the float32 oracle each sit ~9e-5 from the float64 truth there, on either side -- large offsets: golden dc_large)
    x = (rng.standard_normal((5, N)) * 0.4 + (1.5 if n & (n - 1) == 0 else 0.2)).astype(dt)
    x[2] = 0.0
    x[3] = -7.5
    kw = dict(fs=500.0, nperseg=n, window=("tukey", 0.25), noverlap=n - hop, detrend=detrend, mode=mode)
    code = _capi.F32 if dt == "float32" else _capi.F64
    family = "rtiny" if dt == "float32" else "rtinyd"
    plan = plan_for(get_window(("tukey", 0.25), n), n, n, hop, _capi.DETREND[detrend], 500.0, 0, _capi.MODE[mode], code)
    assert plan.kernel == family
    f, t, s = sp.spectrogram(x, **kw)
    fo, to, so = orc.spectrogram(x, **kw)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    assert s.shape == so.shape == (5, n // 2 + 1, frames) and s.dtype == so.dtype
    keep = [0, 1, 4] if detrend else [0, 1, 3, 4]
    tol = {}
    if dt == "float64":
        _check(s[keep], so[keep], np.float64)
    else:
        assert_spec_close(s[keep], so[keep], time_axis=-1, bin_floor=1e-3 if mode == "psd" else 1e-3 ** 0.5, **tol)
    assert np.all(s[2] == 0.0)
    if detrend:
        assert np.all(s[3] == 0.0)                           # the mean of a constant is exact (a power-of-two n, or a true division)
    plan.force_kernel("stockham" if n & (n - 1) == 0 else "bluestein")
    try:
        _, _, s_lds = sp.spectrogram(x, **kw)
    finally:
        plan.force_kernel(family)
    if dt == "float64":
        _check(s[keep], s_lds[keep], np.float64)
    else:
        assert_spec_close(s[keep], s_lds[keep], time_axis=-1, bin_floor=1e-3 if mode == "psd" else 1e-3 ** 0.5, **tol)
    if mode == "psd":                                        # fused band power == the sum over the written bins
        isz = np.dtype(dt).itemsize
        d_in, d_bp = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(5 * frames * isz)
        d_in.upload(np.ascontiguousarray(x))
        h = n // 2
        spec = np.moveaxis(s, -1, -2).astype(np.float64)
        for k_lo, k_hi in [(0, h), (1, 7), (h, h), (0, 0), (h // 2, h - 1), (7, 9)]:
            plan.band_power(d_in.ptr, N, N, 5, k_lo, k_hi, d_bp.ptr, frames)
            bp = np.empty((5, frames), dt)
            d_bp.download(bp)
            _capi.stream_sync()
            ref = spec[:, :, k_lo:k_hi + 1].sum(-1)
            assert np.all(np.abs(bp - ref) <= (2e-6 if dt == "float32" else 1e-12) * spec.sum(-1) + 1e-300), (k_lo, k_hi)
        d_in.free(); d_bp.free()
    # a clip that starts on an odd sample (the register kernel needs aligned pairs) and an odd hop: the Stockham kernel of the same plan
    y = np.ascontiguousarray(x[0, 1:])
    for kw2 in (kw, dict(kw, noverlap=n - 3)):
        _, _, s2 = sp.spectrogram(y, **kw2)
        _, _, so2 = orc.spectrogram(y, **kw2)
        if dt == "float64":
            _check(s2, so2, np.float64)
        else:
            assert_spec_close(s2, so2, time_axis=-1, bin_floor=1e-3 if mode == "psd" else 1e-3 ** 0.5, **tol)
    if dt == "float32":
        xi = np.round(x[:, :n + hop * 40] * 800).astype(np.int16)
        _, _, s_i = sp.spectrogram(xi, **kw)                 # (a GUI-sized int16 call: the Stockham kernel loads int16 itself; batches are
        _, _, so_i = orc.spectrogram(xi, **kw)               #  converted on the device and run this kernel: test_int16_pcm_batches_every_family)
        assert_spec_close(s_i[keep], so_i[keep], time_axis=-1, bin_floor=1e-3 if mode == "psd" else 1e-3 ** 0.5, **tol)


@pytest.mark.parametrize("n", [2048, 4096])
@pytest.mark.parametrize("hop,detrend,mode", [(64, "constant", "psd"), (256, "constant", "psd"), (1024, False, "magnitude"),
                                              (None, "constant", "psd")])
def test_rbig_kernel(sp, n, hop, detrend, mode):
    """Register kernel for nfft 2048 / 4096 (16 / 32 complex values per lane): vs oracle, several clips, ragged tail."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    hop = n - n // 8 if hop is None else hop
    rng = np.random.default_rng(n + hop)
    N = n + hop * 23 + 7
    x = (rng.standard_normal((3, N)) * 0.4 + 0.2).astype(np.float32)
    kw = dict(fs=48000.0, nperseg=n, window="hann", noverlap=n - hop, detrend=detrend, mode=mode)
    f, t, s = sp.spectrogram(x, **kw)
    fo, to, so = orc.spectrogram(x, **kw)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    assert_spec_close(s, so, time_axis=-1, bin_floor=1e-3 if mode == "psd" else 1e-3 ** 0.5)
    plan = plan_for(get_window("hann", n), n, n, hop, _capi.DETREND[detrend], 48000.0, 0, _capi.MODE[mode], _capi.F32)
    assert plan.kernel == "rbig"


@pytest.mark.parametrize("n", [2048, 4096])
@pytest.mark.parametrize("hop,detrend,mode,window", [(64, "constant", "psd", "hann"), (256, "constant", "psd", ("tukey", 0.25)),
                                                     (1024, False, "magnitude", "hann"), (None, "constant", "psd", ("tukey", 0.25))])
def test_rbig_f64_kernel(sp, n, hop, detrend, mode, window):
    """The reference's nperseg 2048 / 4096 on float64 recordings (GUI.py:87-89, SweepManager.py:135-136): the double-precision
    register kernel (stft_rbig_f64.hip) against the oracle, against the LDS kernel of the same plan, band power, edge clips, and the
    fall-back for a clip the 16-byte loads cannot take."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    hop = n - n // 8 if hop is None else hop
    rng = np.random.default_rng(n + hop)
    N = n + hop * 23 + 6
    N += N % 2
    x = rng.standard_normal((5, N)) * 0.4 + 0.2
    x[2] = 0.0
    x[3] = -7.5
    kw = dict(fs=48000.0, nperseg=n, window=window, noverlap=n - hop, detrend=detrend, mode=mode)
    f, t, s = sp.spectrogram(x, **kw)
    fo, to, so = orc.spectrogram(x, **kw)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    _check(s, so, np.float64)
    if detrend:
        assert np.all(s[2] == 0.0) and np.all(s[3] == 0.0)
    plan = plan_for(get_window(window, n), n, n, hop, _capi.DETREND[detrend], 48000.0, 0, _capi.MODE[mode], _capi.F64)
    assert plan.kernel == "rbigd"
    nfr, nb = plan.n_frames(N), n // 2 + 1
    d_in, d_a, d_b = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(5 * nfr * nb * 8), _capi.DeviceBuffer(5 * nfr * nb * 8)
    d_in.upload(x)
    plan.stft(d_in.ptr, N, N, 5, d_a.ptr, nfr * nb)
    plan.force_kernel("stockham")
    try:
        plan.stft(d_in.ptr, N, N, 5, d_b.ptr, nfr * nb)
    finally:
        plan.force_kernel("rbigd")
    a, b = np.empty((5, nfr, nb)), np.empty((5, nfr, nb))
    d_a.download(a)
    d_b.download(b)
    _capi.stream_sync()
    _check(np.moveaxis(a, 1, 2), np.moveaxis(b, 1, 2), np.float64)
    if mode == "psd":
        d_bp = _capi.DeviceBuffer(5 * nfr * 8)
        h = n // 2
        for k_lo, k_hi in [(0, h), (3, 40), (h // 2, h // 2), (h // 2 - 1, h // 2 + 1), (0, 0), (h, h), (100, h - 1)]:
            plan.band_power(d_in.ptr, N, N, 5, k_lo, k_hi, d_bp.ptr, nfr)
            bp = np.empty((5, nfr))
            d_bp.download(bp)
            _capi.stream_sync()
            ref_bp = a[:, :, k_lo:k_hi + 1].sum(-1)
            assert np.all(np.abs(bp - ref_bp) <= 1e-12 * np.abs(a).max(axis=-1) * (k_hi - k_lo + 1) + 1e-300), (k_lo, k_hi)
        d_bp.free()
    # a clip that starts on an odd sample of the buffer (8-byte aligned only): the call falls back to the LDS kernel
    plan.stft(d_in.ptr + 8, N - 1, N, 1, d_b.ptr, nfr * nb)
    b1 = np.empty((plan.n_frames(N - 1), nb))
    d_b.download(b1)
    _capi.stream_sync()
    _, _, so1 = orc.spectrogram(x[0, 1:], **kw)
    _check(b1.T, so1, np.float64)
    for buf in (d_in, d_a, d_b):
        buf.free()
    with pytest.raises(NotImplementedError):
        plan_for(get_window("hann", 2048), 2048, 2048, 256, 1, 1.0, 0, 0, _capi.F32).force_kernel("rbigd")


def test_random_shapes_property(sp):
    """Property test over random (N, nperseg, hop, dtype, detrend): shapes, bit-exact f/t and frame indexing, values vs
    the oracle.  Seeded (no hypothesis dependency on the GPU box); covers N < nperseg, odd and non power-of-two nperseg,
    hop 1 .. nperseg, every kernel family."""
    rng = np.random.default_rng(20260101)
    families = set()
    from spectro.signal import plan_for
    from spectro.windows import get_window
    from spectro import _capi
    for it in range(90):
        nper = int(rng.choice([8, 32, 33, 64, 96, 100, 128, 255, 256, 500, 512, 1000, 1024, 1536, 2048, 4096, 3008, 8192]))
        hop = int(rng.integers(1, nper + 1)) if rng.random() < 0.7 else nper - nper // 8
        hop = max(1, hop)
        n_frames_target = int(rng.integers(1, 40))
        N = nper + hop * (n_frames_target - 1) + int(rng.integers(0, hop))
        if rng.random() < 0.15:
            N = max(2, nper // 2)                                  # N < nperseg -> clamp + warning
        dt = np.float64 if rng.random() < 0.25 else np.float32
        detrend = rng.choice(["constant", "constant", False, "linear"])
        detrend = False if detrend == "False" else detrend
        window = rng.choice(["hann", "tukey", "boxcar"])
        window = ("tukey", 0.25) if window == "tukey" else str(window)
        x = (rng.standard_normal(N) * 0.5 + rng.uniform(-1, 1)).astype(dt)
        kw = dict(fs=float(rng.choice([100.0, 500.0, 16000.0, 48000.0])), nperseg=nper, window=window, detrend=detrend)
        if N >= nper:
            kw["noverlap"] = nper - hop
        with warnings.catch_warnings(record=True) as w1:
            warnings.simplefilter("always")
            f, t, s = sp.spectrogram(x, **kw)
        with warnings.catch_warnings(record=True) as w2:
            warnings.simplefilter("always")
            fo, to, so = orc.spectrogram(x, **kw)
        assert len([w for w in w1 if "nperseg" in str(w.message)]) == len([w for w in w2 if "nperseg" in str(w.message)])
        np.testing.assert_array_equal(f, fo)
        np.testing.assert_array_equal(t, to)
        assert s.shape == so.shape and s.dtype == so.dtype, (it, kw, s.shape, so.shape)
        if dt == np.float64:
            tol = 1e-9 if detrend == "linear" else 1e-11
            assert np.abs(s - so).max() <= tol * max(so.max(), 1e-300), (it, kw)
        else:
            fmax = so.max(axis=0, keepdims=True)
            assert np.all(np.abs(s - so) <= 1e-4 * fmax + 1e-30), (it, kw, float(np.max(np.abs(s - so) / (fmax + 1e-30))))
        n_eff = min(nper, N)
        if N >= nper:
            p = plan_for(get_window(window, n_eff), n_eff, n_eff, hop, _capi.DETREND[detrend], kw["fs"], 0, 0,
                         _capi.F32 if dt == np.float32 else _capi.F64)
            families.add(p.kernel)
    assert {"r8x3", "rsmall", "rbig", "rblue", "stockham", "bluestein"} <= families, families
    assert families & {"rtiny", "rtinyd"} and families & {"rbluew", "rbluewd"}, families      # (round 4: the edges of the spin box)


def test_abi_argument_errors(sp):
    """C ABI error paths: bad sizes / strides / nulls return SG_ERR_ARG (-> ValueError), never launch."""
    import ctypes as C
    from spectro import _capi
    from spectro.windows import get_window
    lib = _capi.lib()
    h = C.c_void_p()
    w = np.ascontiguousarray(get_window("hann", 64))
    wp = w.ctypes.data_as(C.POINTER(C.c_double))
    for args in [(0, 64, 16), (64, 32, 16), (64, 64, 0), (64, 64, 65)]:
        assert lib.sg_plan_create(C.byref(h), args[0], args[1], args[2], wp, 1, 1000.0, 0, 0, 0) == _capi.SG_ERR_ARG
        assert _capi.last_error()
    assert lib.sg_plan_create(C.byref(h), 64, 64, 16, wp, 7, 1000.0, 0, 0, 0) == _capi.SG_ERR_ARG
    assert lib.sg_plan_create(C.byref(h), 64, 64, 16, wp, 1, -1.0, 0, 0, 0) == _capi.SG_ERR_ARG
    assert lib.sg_plan_create(C.byref(h), 64, 64, 16, wp, 1, 1000.0, 0, 0, 5) == _capi.SG_ERR_ARG
    plan = _capi.Plan(64, 64, 16, w, 1, 1000.0, 0, 0, _capi.F32)
    buf = _capi.DeviceBuffer(4096 * 4)
    out = _capi.DeviceBuffer(1 << 20)
    with pytest.raises(ValueError):
        plan.stft(buf.ptr, 1000, 500, 2, out.ptr, 1 << 16)          # clip_stride < n_samples
    with pytest.raises(ValueError):
        plan.stft(buf.ptr, 1000, 1000, 2, out.ptr, 10)              # out_clip_stride too small
    with pytest.raises(ValueError):
        plan.stft(None, 1000, 1000, 1, out.ptr, 1 << 16)            # null input
    with pytest.raises(ValueError):
        plan.band_power(buf.ptr, 1000, 1000, 1, 5, 99, out.ptr, 100)   # band outside the bins
    plan.stft(buf.ptr, 10, 10, 1, out.ptr, 0)                      # fewer samples than nperseg: no frames, no launch
    with pytest.raises(ValueError):
        plan.force_kernel("nope")
    with pytest.raises(NotImplementedError):
        plan.force_kernel("r8x3")


def test_accuracy_against_f64_truth_not_worse_than_scipy_f32(sp):
    """How close is each fp32 path to the f64 result on the same samples?  The device kernels must be at least as
    accurate as the reference's own fp32 path (scipy/pocketfft in complex64), frame by frame, for every register kernel."""
    import scipy.signal as ss
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(200000) * 0.1).astype(np.float32)
    for n, hop in [(256, 64), (512, 128), (1024, 256), (2048, 256), (4096, 1024)]:
        kw = dict(fs=48000.0, nperseg=n, window="hann", noverlap=n - hop)
        _, _, truth = ss.spectrogram(x.astype(np.float64), **kw)
        _, _, ref32 = ss.spectrogram(x, **kw)
        _, _, gpu = sp.spectrogram(x, **kw)
        fmax = truth.max(axis=0, keepdims=True)
        e_ref = np.abs(ref32 - truth) / fmax
        e_gpu = np.abs(gpu - truth) / fmax
        n_ref = np.linalg.norm(ref32 - truth) / np.linalg.norm(truth)
        n_gpu = np.linalg.norm(gpu - truth) / np.linalg.norm(truth)
        print(f"n={n}: per-frame max rel err gpu {e_gpu.max():.2e} scipy-f32 {e_ref.max():.2e}; normwise gpu {n_gpu:.2e} scipy-f32 {n_ref:.2e}")
        assert e_gpu.max() <= max(2.0 * e_ref.max(), 2e-6)
        assert n_gpu <= max(2.0 * n_ref, 5e-7)


def test_long_single_clip_indexing(sp):
    """One 300 M-sample clip (1.2 GB in, 2.4 GB of PSD out, 1.17 M frames): 64-bit frame/sample indexing.  Checked by
    size-independent properties: Parseval on frames near the start, middle and the very end, and periodicity (the signal
    is a 2^20-sample block repeated, so frames 4096 hops apart are identical)."""
    import ctypes as C
    from spectro import _capi, engine
    from spectro.windows import get_window
    rng = np.random.default_rng(77)
    block = (rng.standard_normal(1 << 20) * 0.1).astype(np.float32)
    reps = 286
    x = np.tile(block, reps)                                   # 299.9 M samples
    dev = engine.stft(x, fs=48000.0, nperseg=1024, window="hann", noverlap=768)
    n_frames = (len(x) - 1024) // 256 + 1
    assert dev.n_frames == n_frames and len(dev.t) == n_frames
    assert dev.t[-1] == (512 + (n_frames - 1) * 256.0) / 48000.0
    w = get_window("hann", 1024)
    rows = np.empty((1, 513), np.float32)

    def frame(fr):
        _capi.check(_capi.lib().sg_memcpy_d2h(rows.ctypes.data_as(C.c_void_p), C.c_void_p(dev.buf.ptr + fr * 513 * 4), 513 * 4, None))
        _capi.stream_sync()
        return rows[0].astype(np.float64).copy()

    for fr in (0, 1, n_frames // 2 + 3, n_frames - 2, n_frames - 1):
        seg = x[fr * 256: fr * 256 + 1024].astype(np.float64)
        seg = (seg - seg.mean()) * w
        lhs = frame(fr).sum() * 48000.0 * (w * w).sum() / 1024
        assert abs(lhs - (seg ** 2).sum()) <= 2e-5 * (seg ** 2).sum(), fr
    period = (1 << 20) // 256
    a, b = frame(1234), frame(1234 + 200 * period)
    np.testing.assert_array_equal(a, b)
    dev.free()


def test_c_client_matches_python_path_and_oracle(tmp_path):
    """The plain-C caller (examples/c_client.c) of the ABI on the reference call's arguments: the Python shim's
    result (same plan, same kernel; the two window tables may differ by an ulp of libm's cos) and the oracle's."""
    import importlib.util
    import os
    import subprocess
    from conftest import PKG
    import spectro
    spec = importlib.util.spec_from_file_location("spectro_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    exe = mod.build_c_client()
    rng = np.random.default_rng(31)
    for n, nperseg, fs in [(48000, 1024, 48000.0), (16000, 512, 16000.0), (30000, 250, 500.0)]:
        x = (rng.standard_normal(n) * 0.3).astype(np.float32)
        fin, fout = tmp_path / "x.f32", tmp_path / "s.f32"
        x.tofile(fin)
        r = subprocess.run([exe, str(fin), str(n), repr(fs), str(nperseg), str(fout)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        f, t, s = spectro.spectrogram(x, fs=fs, nperseg=nperseg, scaling="density", mode="psd")
        got = np.fromfile(fout, dtype=np.float32).reshape(len(t), len(f)).T
        assert_spec_close(got, s, tol_frame=2e-6, tol_norm=1e-6, time_axis=-1)
        fo, to, so = orc.spectrogram(x, fs=fs, nperseg=nperseg)
        assert_spec_close(got, so, time_axis=-1)


@pytest.mark.parametrize("n,hop,detrend,mode,window", [
    (1000, 250, "constant", "psd", "hann"), (1000, 876, "constant", "psd", ("tukey", 0.25)), (960, 240, False, "magnitude", "hann"),
    (288, 72, "constant", "psd", ("tukey", 0.25)), (352, 308, "constant", "psd", "hann"), (6, 2, "constant", "psd", "boxcar"),
    (34, 30, False, "psd", "hann"), (1022, 2, "constant", "psd", "hann"), (1026, 256, "constant", "psd", "hann"),
    (1504, 1316, "constant", "psd", ("tukey", 0.25)), (2046, 512, "constant", "magnitude", "hann"), (2016, 64, False, "psd", "boxcar")])
def test_rblue_kernel(sp, n, hop, detrend, mode, window):
    """Even nperseg that is not a power of two (the GUI's spin box steps by 32, GUI.py:87-89): the register chirp-z kernel
    (stft_rblue.hip, round 4) against the oracle -- several clips incl. all-zero and constant ones, a ragged tail -- and against the
    LDS chirp-z kernel of the same plan."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(n + hop)
    N = n + hop * 23 + 6
    x = (rng.standard_normal((5, N)) * 0.4 + 0.2).astype(np.float32)
    x[2] = 0.0
    x[3] = -7.5
    kw = dict(fs=48000.0, nperseg=n, window=window, noverlap=n - hop, detrend=detrend, mode=mode)
    plan = plan_for(get_window(window, n), n, n, hop, _capi.DETREND[detrend], 48000.0, 0, _capi.MODE[mode], _capi.F32)
    assert plan.kernel == "rblue"
    f, t, s = sp.spectrogram(x, **kw)
    fo, to, so = orc.spectrogram(x, **kw)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    assert s.dtype == np.float32 and s.shape == so.shape
    floor = 1e-3 if mode == "psd" else 1e-3 ** 0.5
    keep = [0, 1, 4] if detrend else [0, 1, 3, 4]              # (a detrended constant clip is rounding noise in both: checked below)
    assert_spec_close(s[keep], so[keep], time_axis=-1, bin_floor=floor)
    assert np.all(s[2] == 0.0)
    if detrend:
        assert np.abs(s[3]).max() <= 1e-7 * np.abs(s[0]).max()
    plan.force_kernel("bluestein")
    try:
        _, _, s_lds = sp.spectrogram(x, **kw)
    finally:
        plan.force_kernel("rblue")
    assert_spec_close(s[keep], s_lds[keep], time_axis=-1, bin_floor=floor)


@pytest.mark.parametrize("n,hop", [(1000, 250), (288, 96), (1504, 188)])
def test_rblue_band_power_int16_and_fallbacks(sp, n, hop):
    """The register chirp-z kernel's other entry points: fused band power (A11) == the sum over the written bins; int16 batches
    (converted once on the device); odd hops and clips at an odd stride (the reference's own call at nperseg 1000 has hop 875) take
    4-byte loads in the same kernel, spectra and band power alike, with the same values."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(n * 7 + hop)
    ns = n + hop * 21 + 2
    x = (rng.standard_normal((4, ns)) * 0.3 + 0.5).astype(np.float32)
    plan = plan_for(get_window("hann", n), n, n, hop, 1, 48000.0, 0, 0, _capi.F32)
    assert plan.kernel == "rblue"
    nfr, nb = plan.n_frames(ns), n // 2 + 1
    d_in, d_s, d_bp = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(4 * nfr * nb * 4), _capi.DeviceBuffer(4 * nfr * 4)
    d_in.upload(x)
    plan.stft(d_in.ptr, ns, ns, 4, d_s.ptr, nfr * nb)
    spec = np.empty((4, nfr, nb), np.float32)
    d_s.download(spec)
    _capi.stream_sync()
    h = n // 2
    for k_lo, k_hi in [(0, h), (1, 7), (h // 2, h // 2), (0, 0), (h, h), (33, h - 1)]:
        _capi.check(_capi.lib().sg_memset(C.c_void_p(d_bp.ptr), 0xFF, 4 * nfr * 4, None))
        plan.band_power(d_in.ptr, ns, ns, 4, k_lo, k_hi, d_bp.ptr, nfr)
        bp = np.empty((4, nfr), np.float32)
        d_bp.download(bp)
        _capi.stream_sync()
        ref = spec[:, :, k_lo:k_hi + 1].astype(np.float64).sum(-1)
        assert np.all(np.abs(bp - ref) <= 2e-6 * spec.astype(np.float64).sum(-1) + 1e-30), (k_lo, k_hi)
    # int16 PCM: converted on the device, then the register kernel -- bit-identical to the float call on the same values
    xi = np.round(x * 8000).astype(np.int16)
    _, _, s_i = sp.spectrogram(xi, fs=48000.0, nperseg=n, window="hann", noverlap=n - hop)
    _, _, s_f = sp.spectrogram(xi.astype(np.float32), fs=48000.0, nperseg=n, window="hann", noverlap=n - hop)
    np.testing.assert_array_equal(s_i, s_f)
    # odd hop, and clips at an odd stride (ns - 1 samples of each row): the unaligned loads of the same kernel
    for kw, xs in ((dict(noverlap=n - hop - 1), x), (dict(noverlap=n - hop), x[:, :ns - 1] if (ns - 1) % 2 else x[:, :ns - 2])):
        _, _, s1 = sp.spectrogram(xs, fs=48000.0, nperseg=n, window="hann", **kw)
        _, _, so = orc.spectrogram(xs, fs=48000.0, nperseg=n, window="hann", **kw)
        assert_spec_close(s1, so, time_axis=-1)
    d2 = _capi.DeviceBuffer(4 * nfr * 4)
    plan_odd = plan_for(get_window("hann", n), n, n, hop + 1, 1, 48000.0, 0, 0, _capi.F32)
    nf2 = plan_odd.n_frames(ns)
    plan_odd.band_power(d_in.ptr, ns, ns, 4, 1, 7, d2.ptr, nf2)
    raw = np.empty(4 * nfr, np.float32)
    d2.download(raw)
    _capi.stream_sync()
    _, _, so = orc.spectrogram(x, fs=48000.0, nperseg=n, window="hann", noverlap=n - hop - 1)
    ref = so[:, 1:8, :].sum(axis=1)
    got = raw[:4 * nf2].reshape(4, nf2)
    assert np.all(np.abs(got - ref) <= 1e-4 * np.abs(ref) + 1e-12)
    for b in (d_in, d_s, d_bp, d2):
        b.free()


@pytest.mark.parametrize("n,hop,detrend,mode,window", [
    (3000, 750, "constant", "psd", "hann"), (2080, 1820, "constant", "psd", ("tukey", 0.25)), (2052, 513, False, "magnitude", "hann"),
    (4092, 1024, "constant", "psd", "boxcar"), (4064, 508, False, "psd", "hann"),
    (6000, 1500, "constant", "psd", "hann"), (8160, 7140, "constant", "psd", ("tukey", 0.25)), (4104, 1027, False, "magnitude", "hann"),
    (8184, 2046, "constant", "psd", "boxcar"), (5120, 640, "constant", "magnitude", "hann"), (8192, 2048, "constant", "psd", "hann"),
    (8192, 7168, "constant", "psd", ("tukey", 0.25))])
def test_rbluew_kernel(sp, n, hop, detrend, mode, window):
    """nperseg 2080 ... 8192 that is no power of two (the GUI's spin box runs to 8192 in steps of 32, GUI.py:87-89): the wide register
    chirp-z kernel (stft_rbluew.hip, round 4: two wavefronts per frame up to 4096, four above) against the oracle -- several clips incl.
    all-zero and constant ones, odd hops, the reference's own hop n - n // 8 -- and against the LDS chirp-z kernel of the same plan."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(n + hop)
    N = n + hop * 11 + 6
    x = (rng.standard_normal((5, N)) * 0.4 + 0.2).astype(np.float32)
    x[2] = 0.0
    x[3] = -7.5
    kw = dict(fs=48000.0, nperseg=n, window=window, noverlap=n - hop, detrend=detrend, mode=mode)
    plan = plan_for(get_window(window, n), n, n, hop, _capi.DETREND[detrend], 48000.0, 0, _capi.MODE[mode], _capi.F32)
    assert plan.kernel == "rbluew"
    f, t, s = sp.spectrogram(x, **kw)
    fo, to, so = orc.spectrogram(x, **kw)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    assert s.dtype == np.float32 and s.shape == so.shape
    floor = 1e-3 if mode == "psd" else 1e-3 ** 0.5
    keep = [0, 1, 4] if detrend else [0, 1, 3, 4]
    assert_spec_close(s[keep], so[keep], time_axis=-1, bin_floor=floor)
    assert np.all(s[2] == 0.0)
    if detrend:
        assert np.abs(s[3]).max() <= 1e-7 * np.abs(s[0]).max()
    plan.force_kernel("bluestein")
    try:
        _, _, s_lds = sp.spectrogram(x, **kw)
    finally:
        plan.force_kernel("rbluew")
    assert_spec_close(s[keep], s_lds[keep], time_axis=-1, bin_floor=floor)


@pytest.mark.parametrize("n,hop,clips,frames", [(2080, 64, 3, 701), (4128, 96, 2, 1031), (2400, 32, 1, 1025), (8000, 4000, 1, 1), (8192, 64, 2, 641)])
def test_rbluew_many_frames_band_power_and_int16(sp, n, hop, clips, frames):
    """More frames than the launch has frame groups (1 024 / 512: every group loops, the last pass is ragged) and a single frame; the fused
    band power (A11) == the sum over the written bins; int16 batches (converted once on the device) == the float call; clips at an
    odd stride take the 4-byte loads."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(n * 7 + hop)
    ns = n + hop * (frames - 1) + 3
    x = (rng.standard_normal((clips, ns)) * 0.3 + 0.5).astype(np.float32)
    plan = plan_for(get_window("hann", n), n, n, hop, 1, 48000.0, 0, 0, _capi.F32)
    assert plan.kernel == "rbluew"
    kw = dict(fs=48000.0, nperseg=n, window="hann", noverlap=n - hop)
    _, _, s = sp.spectrogram(x, **kw)
    _, _, so = orc.spectrogram(x, **kw)
    assert s.shape == so.shape == (clips, n // 2 + 1, frames)
    # (per-bin bound 3e-4 over these 1-2 M bins: the tail of float32 rounding at bins 1e-3 of the frame maximum -- scipy's own float32
    # path shows 1.2e-4 / 2.0e-4 on the same inputs, this kernel 1.3e-4 / 2.0e-4, tools/acc_np2.py; frame and norm bounds as everywhere)
    # (nperseg 8192, 5.2 M bins: 6e-4 -- the oracle computes in the input's precision as scipy does, so BOTH sides carry float32 rounding: against
    # float64 truth this kernel shows 1.9e-4, scipy's float32 path 2.0e-4, tools/acc_np2.py)
    assert_spec_close(s, so, time_axis=-1, bin_rtol=3e-4 if n < 8192 else 6e-4)
    nfr, nb = plan.n_frames(ns), n // 2 + 1
    d_in, d_bp = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(clips * nfr * 4)
    d_in.upload(x)
    h = n // 2
    spec = np.moveaxis(s, -1, -2).astype(np.float64)
    for k_lo, k_hi in [(0, h), (1, 7), (h, h), (h // 2, h - 1), (63, 64)]:
        _capi.check(_capi.lib().sg_memset(C.c_void_p(d_bp.ptr), 0xFF, clips * nfr * 4, None))
        plan.band_power(d_in.ptr, ns, ns, clips, k_lo, k_hi, d_bp.ptr, nfr)
        bp = np.empty((clips, nfr), np.float32)
        d_bp.download(bp)
        _capi.stream_sync()
        ref = spec[:, :, k_lo:k_hi + 1].sum(-1)
        assert np.all(np.abs(bp - ref) <= 2e-6 * spec.sum(-1) + 1e-30), (k_lo, k_hi)
    d_in.free(); d_bp.free()
    xi = np.round(x[:, :n + hop * min(frames - 1, 40) + 1] * 8000).astype(np.int16)
    _, _, s_i = sp.spectrogram(xi, **kw)
    _, _, s_f = sp.spectrogram(xi.astype(np.float32), **kw)
    np.testing.assert_array_equal(s_i, s_f)
    if clips > 1:
        xs = x[:, :ns - 1] if (ns - 1) % 2 else x[:, :ns - 2]
        _, _, s1 = sp.spectrogram(xs, **kw)
        _, _, so1 = orc.spectrogram(xs, **kw)
        assert_spec_close(s1, so1, time_axis=-1, bin_rtol=3e-4 if n < 8192 else 6e-4)


@pytest.mark.parametrize("n,hop,detrend,mode,window", [
    (1056, 264, "constant", "psd", "hann"), (2000, 1750, "constant", "psd", ("tukey", 0.25)), (1028, 257, False, "magnitude", "hann"),
    (2044, 512, "constant", "psd", "boxcar"),
    (3000, 750, "constant", "psd", "hann"), (4088, 3577, "constant", "psd", ("tukey", 0.25)), (2056, 515, False, "magnitude", "hann"),
    (6000, 1500, "constant", "psd", "hann"), (8160, 7140, "constant", "psd", ("tukey", 0.25)), (4112, 1029, False, "magnitude", "hann"),
    (8176, 2044, "constant", "psd", "boxcar"), (8192, 7168, "constant", "psd", ("tukey", 0.25)), (8192, 1024, False, "magnitude", "hann")])
def test_rbluew_f64_kernel(sp, n, hop, detrend, mode, window):
    """The reference's own flow at nperseg 1056 ... 8160 (float64 recordings, SweepManager.py:135-136; the spin box runs to 8192 in steps
    of 32, GUI.py:87-89): the wide double-precision register chirp-z kernel (stft_rbluew_f64.hip, round 4: two / four / eight wavefronts
    per frame) against the oracle at f64 tolerance, against the LDS kernel of the same plan, the fused band power, odd hops (the
    reference's literal hop is n - n // 8) and clips at an odd stride."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(n * 3 + hop)
    N = n + hop * 11 + 6
    x = rng.standard_normal((5, N)) * 0.4 + 0.2
    x[2] = 0.0
    x[3] = -7.5
    kw = dict(fs=48000.0, nperseg=n, window=window, noverlap=n - hop, detrend=detrend, mode=mode)
    plan = plan_for(get_window(window, n), n, n, hop, _capi.DETREND[detrend], 48000.0, 0, _capi.MODE[mode], _capi.F64)
    assert plan.kernel == "rbluewd"
    f, t, s = sp.spectrogram(x, **kw)
    fo, to, so = orc.spectrogram(x, **kw)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    keep = [0, 1, 4] if detrend else [0, 1, 3, 4]
    _check(s[keep], so[keep], np.float64)
    assert np.all(s[2] == 0.0)
    if detrend:
        assert np.abs(s[3]).max() <= 1e-20 * max(np.abs(s[0]).max(), 1e-300) + 1e-28
    plan.force_kernel("bluestein")
    try:
        _, _, s_lds = sp.spectrogram(x, **kw)
    finally:
        plan.force_kernel("rbluewd")
    _check(s[keep], s_lds[keep], np.float64)
    xs = x[:, :N - 1] if (N - 1) % 2 else x[:, :N - 2]       # clips at an odd stride: 8-byte loads
    _, _, s1 = sp.spectrogram(xs, **kw)
    _, _, so1 = orc.spectrogram(xs, **kw)
    _check(s1[keep], so1[keep], np.float64)
    if mode == "psd":                                        # fused band power == the sum over the written bins
        nfr = plan.n_frames(N)
        d_in, d_bp = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(5 * nfr * 8)
        d_in.upload(np.ascontiguousarray(x))
        h = n // 2
        for k_lo, k_hi in [(0, h), (1, 7), (h, h), (h // 2, h - 1), (63, 64)]:
            plan.band_power(d_in.ptr, N, N, 5, k_lo, k_hi, d_bp.ptr, nfr)
            bp = np.empty((5, nfr), np.float64)
            d_bp.download(bp)
            _capi.stream_sync()
            ref = np.moveaxis(s, -1, -2)[:, :, k_lo:k_hi + 1].sum(-1)
            assert np.all(np.abs(bp - ref) <= 1e-12 * np.moveaxis(s, -1, -2).sum(-1) + 1e-300), (k_lo, k_hi)
        d_in.free(); d_bp.free()


@pytest.mark.parametrize("n,hop,clips,frames", [(1056, 32, 3, 701), (2080, 48, 2, 531), (4112, 16, 1, 259), (4112, 2000, 1, 1), (8192, 32, 3, 301)])
def test_rbluew_f64_many_frames(sp, n, hop, clips, frames):
    """More frames than the launch has frame groups (1 024 / 512 / 256: every group loops, the last pass is ragged) and a single frame."""
    rng = np.random.default_rng(n * 5 + hop)
    ns = n + hop * (frames - 1) + 3
    x = rng.standard_normal((clips, ns)) * 0.3 + 0.5
    kw = dict(fs=48000.0, nperseg=n, window="hann", noverlap=n - hop)
    _, _, s = sp.spectrogram(x, **kw)
    _, _, so = orc.spectrogram(x, **kw)
    assert s.shape == so.shape == (clips, n // 2 + 1, frames)
    _check(s, so, np.float64)


@pytest.mark.parametrize("n,hop,detrend,mode,window", [
    (1000, 250, "constant", "psd", "hann"), (1000, 875, "constant", "psd", ("tukey", 0.25)), (960, 240, False, "magnitude", "hann"),
    (288, 72, "constant", "psd", ("tukey", 0.25)), (480, 419, "constant", "psd", "hann"), (6, 2, "constant", "psd", "boxcar"),
    (34, 30, False, "psd", "hann"), (1022, 2, "constant", "magnitude", "hann"), (514, 128, "constant", "psd", "boxcar")])
def test_rblue_f64_kernel(sp, n, hop, detrend, mode, window):
    """The reference's own flow at a non-power-of-two nperseg: float64 recordings (SweepManager.py:135-136) and the spin box's 32-steps
    (GUI.py:87-89).  The double-precision register chirp-z kernel (stft_rblue_f64.hip, round 4) against the oracle at f64 tolerance, against the
    LDS kernel of the same plan, the fused band power, an odd hop (the reference's literal hop at nperseg 1000 is 875) and clips at an odd stride."""
    from spectro import _capi
    from spectro.signal import plan_for
    from spectro.windows import get_window
    rng = np.random.default_rng(n * 3 + hop)
    N = n + hop * 23 + 6
    x = rng.standard_normal((5, N)) * 0.4 + 0.2
    x[2] = 0.0
    x[3] = -7.5
    kw = dict(fs=48000.0, nperseg=n, window=window, noverlap=n - hop, detrend=detrend, mode=mode)
    plan = plan_for(get_window(window, n), n, n, hop, _capi.DETREND[detrend], 48000.0, 0, _capi.MODE[mode], _capi.F64)
    assert plan.kernel == "rblued"
    f, t, s = sp.spectrogram(x, **kw)
    fo, to, so = orc.spectrogram(x, **kw)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(t, to)
    keep = [0, 1, 4] if detrend else [0, 1, 3, 4]
    _check(s[keep], so[keep], np.float64)
    assert np.all(s[2] == 0.0)
    if detrend:
        assert np.abs(s[3]).max() <= 1e-20 * max(np.abs(s[0]).max(), 1e-300) + 1e-28
    plan.force_kernel("bluestein")
    try:
        _, _, s_lds = sp.spectrogram(x, **kw)
    finally:
        plan.force_kernel("rblued")
    _check(s[keep], s_lds[keep], np.float64)
    # clips at an odd stride (N - 1 samples of each row of the same buffer): 8-byte loads
    xs = x[:, :N - 1] if (N - 1) % 2 else x[:, :N - 2]
    _, _, s1 = sp.spectrogram(xs, **kw)
    _, _, so1 = orc.spectrogram(xs, **kw)
    _check(s1[keep], so1[keep], np.float64)
    if mode == "psd":                                        # fused band power == the sum over the written bins
        nfr, nb = plan.n_frames(N), n // 2 + 1
        d_in, d_bp = _capi.DeviceBuffer(x.nbytes), _capi.DeviceBuffer(5 * nfr * 8)
        d_in.upload(np.ascontiguousarray(x))
        h = n // 2
        for k_lo, k_hi in [(0, h), (1, min(7, h)), (h, h), (h // 2, h - 1)]:
            plan.band_power(d_in.ptr, N, N, 5, k_lo, k_hi, d_bp.ptr, nfr)
            bp = np.empty((5, nfr), np.float64)
            d_bp.download(bp)
            _capi.stream_sync()
            ref = np.moveaxis(s, -1, -2)[:, :, k_lo:k_hi + 1].sum(-1)
            assert np.all(np.abs(bp - ref) <= 1e-12 * np.moveaxis(s, -1, -2).sum(-1) + 1e-300), (k_lo, k_hi)
        d_in.free(); d_bp.free()
