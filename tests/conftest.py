"""pytest configuration: markers, import paths, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "spectrogram-generator_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # libspectro.so is a build product (git-ignored): (re)build it in-tree when it is missing or older than its sources,
    # so that a fresh checkout can run the suite directly.  hipcc cross-compiles without a GPU.
    import importlib.util
    spec = importlib.util.spec_from_file_location("spectro_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        mod.build(force=False, verbose=False)
    except Exception as e:                      # no hipcc on this machine: the ABI tests will say so loudly
        print(f"[conftest] could not build libspectro.so: {e}", file=sys.stderr)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


# ---- seeded inputs shared with tests/golden/make_golden.py (same recipes) ----
def cfg1_signal():
    return np.random.default_rng(0).standard_normal(16000)


def cfg2_clips(n_clips=2):
    x = np.random.default_rng(1234).standard_normal((64, 480000)).astype(np.float32) * np.float32(0.1)
    return np.ascontiguousarray(x[:n_clips])


def sweep_clip():
    return (np.random.default_rng(77).standard_normal(48000) * 0.1).astype(np.float32)


def eeg_like():
    rng = np.random.default_rng(5)
    n = 10000
    t = np.arange(n) / 500.0
    x = np.cumsum(rng.standard_normal(n)) * 0.01 + 0.2 * rng.standard_normal(n)
    x += np.where((t > 5) & (t < 8), 1.5 * np.sin(2 * np.pi * 10 * t), 0.0)
    return x + 3.0


def assert_spec_close(got, ref, tol_frame=1e-4, tol_norm=1e-5, bin_rtol=1e-4, time_axis=-1, bin_floor=1e-3):
    """The parity criterion of BASELINE.md §2 / SURVEY H2 for fp32 spectra.

    per frame: |got-ref| <= tol_frame * max_k ref[k];  normwise ||d||/||ref|| <= tol_norm;
    per-bin rtol only for bins >= 1e-3 * frame max (near-zero bins suffer cancellation in
    fp32 for ANY implementation, scipy's own f32 path included).  For magnitude spectra pass
    bin_floor=sqrt(1e-3): the same bins in power terms.
    """
    wide = np.complex128 if (np.iscomplexobj(got) or np.iscomplexobj(ref)) else np.float64
    got = np.asarray(got, wide)
    ref = np.asarray(ref, wide)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if ref.size == 0:
        return
    g = np.moveaxis(got, time_axis, 0).reshape(got.shape[time_axis], -1) if got.ndim > 1 else got[None]
    r = np.moveaxis(ref, time_axis, 0).reshape(ref.shape[time_axis], -1) if ref.ndim > 1 else ref[None]
    fmax = np.abs(r).max(axis=1, keepdims=True)
    d = np.abs(g - r)
    assert np.all(d <= tol_frame * fmax + 1e-300), f"per-frame err {np.max(d / (fmax + 1e-300)):.3e}"
    nr = np.linalg.norm(r)
    if nr > 0:
        assert np.linalg.norm(g - r) / nr <= tol_norm, f"normwise {np.linalg.norm(g - r) / nr:.3e}"
    big = (np.abs(r) >= bin_floor * fmax) & (fmax > 0)
    if big.any():
        rel = d[big] / np.abs(r[big])
        assert rel.max() <= bin_rtol, f"per-bin rel {rel.max():.3e}"
