"""Maximum sizes (GPU): outputs beyond 2^31 elements / 2^32 bytes and clip strides beyond 2^32 bytes, for every kernel family.

Size-independent property used: a recording that is periodic with period ``hop * K`` samples has a spectrogram that is periodic
with period ``K`` frames -- frame f is frame f mod K, bit for bit (same samples, same arithmetic).  So
  * the first K device frames are checked against the oracle at the usual tolerance;
  * frames sampled around the 2^29 / 2^30 / 2^31-element marks, at the very end and at random must EQUAL device frame f mod K;
  * the output buffer is pre-filled with a large negative pattern and min / max over ALL of it (``sg_minmax``, itself a 64-bit
    walk) must equal min / max of the first K frames: one unwritten, misplaced or clobbered element anywhere changes the minimum.
The recording is built on the device (one period uploaded, doubled by device-to-device copies), so the host never holds it.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_spec_close
from oracle import stft_oracle as orc

pytestmark = pytest.mark.gpu

K = 7                               # frames per period
FILL = 0xFE                         # 0xFEFEFEFE = -1.7e38 as f32, 0xFEFE... = -5.8e303 as f64: below any PSD / magnitude


@pytest.fixture(scope="module")
def capi():
    from spectro import _capi
    _capi.ensure_device()
    yield _capi
    _capi.device_pool_clear()


def _fill_periodic(capi, d_buf, pattern, total):
    """d_buf[0:total] = pattern repeated (pattern.size doubles per copy, so every copy starts on a period boundary)"""
    L, isz = capi.lib(), pattern.itemsize
    d_buf.upload(pattern)
    filled = pattern.size
    while filled < total:
        n = min(filled, total - filled)
        capi.check(L.sg_memcpy_d2d(C.c_void_p(d_buf.ptr + filled * isz), C.c_void_p(d_buf.ptr), n * isz, None))
        filled += n


def _rows(capi, d_out, first_elem, n_rows, nb, dt):
    out = np.empty((n_rows, nb), dt)
    capi.check(capi.lib().sg_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(d_out.ptr + first_elem * out.itemsize), out.nbytes, None))
    capi.stream_sync()
    return out


def _minmax(capi, ptr, code, n_frames, nb, dt):
    d_mm = capi.DeviceBuffer(16)
    capi.check(capi.lib().sg_minmax(C.c_void_p(ptr), code, int(n_frames), nb, 0, nb - 1, C.c_void_p(d_mm.ptr), None))
    mm = np.empty(2, dt)
    d_mm.download(mm)
    capi.stream_sync()
    d_mm.free()
    return mm


def _close(got, ref, dt):
    if dt == np.float64:
        fmax = np.abs(ref).max(axis=-1, keepdims=True)
        assert np.all(np.abs(got - ref) <= 1e-11 * fmax + 1e-300)
    else:
        assert_spec_close(got, ref, time_axis=0)


CASES = [  # dtype, nperseg, hop, window, forced family, expected family
    ("f32", 1024, 256, "hann", None, "r8x3"),
    ("f32", 256, 64, "hann", None, "rsmall"),
    ("f32", 4096, 1024, "hann", None, "rbig"),
    ("f32", 1024, 256, ("tukey", 0.25), "stockham", "stockham"),
    ("f32", 1000, 250, "hann", None, "rblue"),
    ("f32", 1000, 250, "hann", "bluestein", "bluestein"),
    ("f64", 1024, 256, ("tukey", 0.25), None, "r8x3d"),
    ("f64", 1000, 250, "hann", None, "rblued"),
    ("f32", 64, 16, "hann", None, "rtiny"),
    ("f32", 4000, 1000, "hann", None, "rbluew"),
    ("f64", 4000, 1000, "hann", None, "rbluewd"),
]


@pytest.mark.parametrize("dtn,nperseg,hop,window,force,family", CASES, ids=[c[5] for c in CASES])
def test_output_beyond_2_31_elements(capi, dtn, nperseg, hop, window, force, family):
    from spectro.windows import get_window
    dt, code = (np.float32, capi.F32) if dtn == "f32" else (np.float64, capi.F64)
    nb = nperseg // 2 + 1
    n_frames = -(-((1 << 31) + (1 << 20)) // nb)
    n_frames += (-n_frames) % K
    n_samples = (n_frames - 1) * hop + nperseg
    rng = np.random.default_rng(nperseg + hop)
    pattern = (rng.standard_normal(hop * K) * 0.3 + 0.2).astype(dt)
    plan = capi.Plan(nperseg, nperseg, hop, get_window(window, nperseg), 1, 48000.0, 0, 0, code)
    if force:
        plan.force_kernel(force)
    assert plan.kernel == family and plan.n_frames(n_samples) == n_frames
    isz = np.dtype(dt).itemsize
    d_in, d_out = capi.DeviceBuffer(n_samples * isz), capi.DeviceBuffer(n_frames * nb * isz)
    try:
        _fill_periodic(capi, d_in, pattern, n_samples)
        capi.check(capi.lib().sg_memset(C.c_void_p(d_out.ptr), FILL, n_frames * nb * isz, None))
        plan.stft(d_in.ptr, n_samples, n_samples, 1, d_out.ptr, n_frames * nb)
        head = _rows(capi, d_out, 0, K, nb, dt)
        x_small = np.tile(pattern, -(-(nperseg + K * hop) // pattern.size) + 1)[:nperseg + (K - 1) * hop]
        _, _, so = orc.spectrogram(x_small, fs=48000.0, nperseg=nperseg, window=window, noverlap=nperseg - hop)
        assert so.shape == (nb, K)
        _close(head, so.T, dt)
        marks = [0, n_frames - K] + [max(0, (1 << b) // nb - 2) for b in (29, 30, 31)]
        picks = sorted({f for m in marks for f in range(m, min(m + 5, n_frames))} | set(rng.integers(0, n_frames, 48).tolist())
                       | set(range(n_frames - K, n_frames)))
        for f in picks:
            np.testing.assert_array_equal(_rows(capi, d_out, f * nb, 1, nb, dt)[0], head[f % K], err_msg=f"frame {f}")
        mm = _minmax(capi, d_out.ptr, code, n_frames, nb, dt)
        assert mm[0] == head.min() and mm[1] == head.max(), (mm, head.min(), head.max())
    finally:
        d_in.free(); d_out.free(); plan.close()
        capi.device_pool_clear()


@pytest.mark.parametrize("dtn", ["f32", "f64"])
def test_clip_strides_beyond_2_32_bytes(capi, dtn):
    """Three clips of one buffer, input stride 2^30 + 901 elements and output stride 2^31 + 1027 elements: the clip offsets do
    not fit 32 bits in elements (output) or bytes (input), and the stride is no multiple of the period, so a truncated offset
    would read different samples."""
    from spectro.windows import get_window
    dt, code = (np.float32, capi.F32) if dtn == "f32" else (np.float64, capi.F64)
    nperseg, hop, nb, n_clips, n_samples = 1024, 256, 513, 3, 30000
    in_stride, out_stride = (1 << 30) + 901 + (0 if dtn == "f32" else 1), (1 << 31) + 1027     # f64 register kernel wants even strides
    rng = np.random.default_rng(5)
    pattern = (rng.standard_normal(hop * K) * 0.3 - 0.1).astype(dt)
    plan = capi.Plan(nperseg, nperseg, hop, get_window("hann", nperseg), 1, 48000.0, 0, 0, code)
    assert plan.kernel == ("r8x3" if dtn == "f32" else "r8x3d")
    n_frames = plan.n_frames(n_samples)
    isz = np.dtype(dt).itemsize
    total_in = (n_clips - 1) * in_stride + n_samples
    total_out = (n_clips - 1) * out_stride + n_frames * nb
    d_in, d_out = capi.DeviceBuffer(total_in * isz), capi.DeviceBuffer(total_out * isz)
    try:
        _fill_periodic(capi, d_in, pattern, total_in)
        capi.check(capi.lib().sg_memset(C.c_void_p(d_out.ptr), FILL, total_out * isz, None))
        plan.stft(d_in.ptr, n_samples, in_stride, n_clips, d_out.ptr, out_stride)
        fill_value = np.frombuffer(bytes([FILL]) * isz, dt)[0]
        for c in range(n_clips):
            shift = (c * in_stride) % pattern.size
            x_c = np.tile(pattern, n_samples // pattern.size + 3)[shift:shift + n_samples]
            _, _, so = orc.spectrogram(x_c, fs=48000.0, nperseg=nperseg, window="hann", noverlap=nperseg - hop)
            got = _rows(capi, d_out, c * out_stride, n_frames, nb, dt)
            _close(got, so.T, dt)
            if c:                                            # nothing written just below this clip's rows
                guard = _rows(capi, d_out, c * out_stride - nb, 1, nb, dt)[0]
                assert np.all(guard == fill_value)
    finally:
        d_in.free(); d_out.free(); plan.close()
        capi.device_pool_clear()
