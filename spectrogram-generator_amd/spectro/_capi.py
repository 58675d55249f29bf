"""ctypes binding of libspectro.so (include/spectro.h) -- the only door to the device.

There is NO CPU fallback: if the library is missing, cannot be loaded, or no gfx950
device is visible, every compute entry point raises.  (The CPU oracle under oracle/
is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("SPECTRO_LIB", os.path.join(_PKG, "lib", "libspectro.so"))

SG_OK, SG_ERR_ARG, SG_ERR_HIP, SG_ERR_UNSUPPORTED, SG_ERR_NO_DEVICE = 0, -1, -2, -3, -4
DETREND = {False: 0, None: 0, "none": 0, "constant": 1, "c": 1, "linear": 2, "l": 2}
SCALING = {"density": 0, "spectrum": 1}
MODE = {"psd": 0, "magnitude": 1, "complex": 2, "angle": 3}
F32, F64 = 0, 1

_vp, _i, _i64, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t
_pvp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/spectro.h declaration by declaration
SIGNATURES = {
    "sg_version": (_i, []),
    "sg_last_error": (C.c_char_p, []),
    "sg_device_count": (_i, [C.POINTER(_i)]),
    "sg_init": (_i, [_i]),
    "sg_device_info": (_i, [C.c_char_p, _sz, C.POINTER(_i), C.POINTER(C.c_uint64)]),
    "sg_mem_info": (_i, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "sg_device_pci_bus_id": (_i, [C.c_char_p, _sz]),
    "sg_malloc": (_i, [_pvp, _sz]),
    "sg_free": (_i, [_vp]),
    "sg_host_alloc": (_i, [_pvp, _sz]),
    "sg_host_free": (_i, [_vp]),
    "sg_host_register": (_i, [_vp, _sz]),
    "sg_host_unregister": (_i, [_vp]),
    "sg_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "sg_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "sg_memcpy_d2d": (_i, [_vp, _vp, _sz, _vp]),
    "sg_memcpy2d": (_i, [_vp, _sz, _vp, _sz, _sz, _sz, _i, _vp]),
    "sg_memset": (_i, [_vp, _i, _sz, _vp]),
    "sg_convert_i16": (_i, [_vp, _vp, _i64, _vp]),
    "sg_workspace_release": (_i, []),
    "sg_stream_create": (_i, [_pvp]),
    "sg_stream_destroy": (_i, [_vp]),
    "sg_stream_sync": (_i, [_vp]),
    "sg_plan_create": (_i, [_pvp, _i, _i, _i, C.POINTER(_d), _i, _d, _i, _i, _i]),
    "sg_plan_destroy": (_i, [_vp]),
    "sg_plan_n_frames": (_i, [_vp, _i64, C.POINTER(_i64)]),
    "sg_plan_n_bins": (_i, [_vp, C.POINTER(_i)]),
    "sg_plan_scale": (_i, [_vp, C.POINTER(_d)]),
    "sg_plan_kernel": (C.c_char_p, [_vp]),
    "sg_plan_force_kernel": (_i, [_vp, C.c_char_p]),
    "sg_freqs": (_i, [_i, _d, C.POINTER(_d)]),
    "sg_times": (_i, [_i64, _i, _i, _d, C.POINTER(_d)]),
    "sg_stft": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _i64, _vp]),
    "sg_stft_i16": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _i64, _vp]),
    "sg_stft_band_power": (_i, [_vp, _vp, _i64, _i64, _i, _i, _i, _vp, _i64, _vp]),
    "sg_stft_db": (_i, [_vp, _vp, _i64, _i64, _i, _i, _i, _d, _vp, _i64, _vp, _vp]),
    "sg_db_rescale": (_i, [_vp, _i64, _vp, _vp]),
    "sg_colormap_db": (_i, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "sg_minmax": (_i, [_vp, _i, _i64, _i, _i, _i, _vp, _vp]),
    "sg_normalise_image": (_i, [_vp, _i, _i64, _i, _i, _i, _i, _d, _vp, _vp, _vp]),
    "sg_band_features": (_i, [_vp, _i, _i64, _vp, _vp]),
    "sg_band_features_batch": (_i, [_vp, _i, _i, _i64, _vp, _vp]),
    "sg_band_sum": (_i, [_vp, _i, _i64, _i, _i, _i, _vp, _vp]),
    "sg_band_totals": (_i, [_vp, _i, _i64, _i, _i, C.POINTER(_i), C.POINTER(_i), _vp, _vp]),
    "sg_slice_bins": (_i, [_vp, _i, _i64, _i, _i, _i, _vp, _vp]),
    "sg_jet_lut": (_i, [C.POINTER(C.c_uint8)]),
    "sg_colormap": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "sg_mel_weights": (_i, [_i, _d, _i, _d, _d, C.POINTER(_d)]),
    "sg_mel_pack_weights": (_i, [C.POINTER(_d), _i, _i, C.POINTER(C.c_float)]),
    "sg_mel_tile_ranges": (_i, [C.POINTER(_d), _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "sg_mel": (_i, [_vp, _i64, _i, _vp, _i, C.POINTER(_i), C.POINTER(_i), _i, _vp, _vp]),
    "sg_stft_mel": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _i, _i, C.POINTER(_i), C.POINTER(_i), _i, _vp, _i64, _vp]),
    "sg_mel_sparse_pack": (_i, [C.POINTER(_d), _i, _i, C.POINTER(_i), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int32)]),
    "sg_mel_sparse": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "sg_stft_mel_sparse": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp]),
    "sg_time_stft": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _i64, _vp, _i, C.POINTER(C.c_float)]),
}

_lib = None
_lock = threading.Lock()
_device_ready = None


class SpectroError(RuntimeError):
    pass


def _elf_dynamic(path):
    """(DT_SONAME, [DT_NEEDED ...]) of a 64-bit little-endian ELF file, read from its section headers (no readelf needed)."""
    import struct
    with open(path, "rb") as fh:
        ident = fh.read(64)
        if ident[:4] != b"\x7fELF" or ident[4] != 2 or ident[5] != 1:
            raise OSError(f"{path}: not a 64-bit little-endian ELF file")
        e_shoff, = struct.unpack_from("<Q", ident, 0x28)
        e_shentsize, e_shnum = struct.unpack_from("<HH", ident, 0x3A)
        fh.seek(e_shoff)
        heads = [struct.unpack_from("<IIQQQQIIQQ", fh.read(e_shentsize)) for _ in range(e_shnum)]
        dyn = next((h for h in heads if h[1] == 6), None)    # SHT_DYNAMIC; sh_link = its string table
        if dyn is None:
            return None, []
        strtab = heads[dyn[6]]
        fh.seek(strtab[4]); names = fh.read(strtab[5])
        fh.seek(dyn[4]); raw = fh.read(dyn[5])

    def name(off):
        return names[off:names.index(b"\0", off)].decode()
    soname, needed = None, []
    for i in range(0, len(raw) - 15, 16):
        tag, val = struct.unpack_from("<qQ", raw, i)
        if tag == 0:
            break
        if tag == 1:
            needed.append(name(val))
        elif tag == 14:
            soname = name(val)
    return soname, needed


hip_runtime = {"choice": None, "path": None, "why": None}   # which HIP runtime lib() arranged for, and why (read by the tests / for bug reports)


def _share_torchs_hip_runtime():
    """PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` and ask for it by that unversioned name, so a process that has
    loaded libspectro.so first (against the system's ``libamdhip64.so.7``) gets a SECOND HIP / HSA runtime when torch is imported
    later -- and the second one finds no GPU ("No HIP GPUs are available"), which would break ``spectro.dist`` / ``spectro.sweep``
    for a caller who touched the engine before importing torch.  Loaded in this order instead -- torch's runtime first, by path --
    the dynamic linker gives libspectro.so the same copy, and torch finds it loaded.  That only holds when the SONAME of torch's
    copy is the very name libspectro.so asks for (its DT_NEEDED entry): both are read from the ELF files, and a torch built for
    another ROCm major is left alone -- libspectro.so then runs on the system runtime it was built against, with a warning that a
    torch imported later in this process will not see the GPU.
    ``SPECTRO_HIP_RUNTIME``: ``auto`` (default: as above), ``system`` (never preload; processes that never import torch need
    nothing else), ``torch`` (preload or raise).  ``hip_runtime`` records what was done."""
    import sys
    import warnings
    want = os.environ.get("SPECTRO_HIP_RUNTIME", "auto")
    if want not in ("auto", "system", "torch"):
        raise ValueError(f"SPECTRO_HIP_RUNTIME={want!r}: expected auto, system or torch")

    def settle(choice, path, why):
        hip_runtime.update(choice=choice, path=path, why=why)
        if os.environ.get("SPECTRO_VERBOSE"):
            print(f"[spectro] HIP runtime: {choice}" + (f" ({path})" if path else "") + f" -- {why}", file=sys.stderr)
    if "torch" in sys.modules:
        return settle("loaded", None, "torch was imported first: its runtime is already mapped")
    if want == "system":
        return settle("system", None, "SPECTRO_HIP_RUNTIME=system")
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")           # locates the package without importing it
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so") if spec and spec.submodule_search_locations else None
        if not path or not os.path.exists(path):
            if want == "torch":
                raise ImportError("SPECTRO_HIP_RUNTIME=torch: no torch/lib/libamdhip64.so in this environment")
            return settle("system", None, "no torch (or no bundled HIP runtime) here")
        soname, _ = _elf_dynamic(path)
        _, needed = _elf_dynamic(LIB_PATH)
        mine = [n for n in needed if n.startswith("libamdhip64.so")]
        if soname not in mine:
            msg = (f"torch bundles a HIP runtime with SONAME {soname!r}, libspectro.so needs {mine or needed}: they cannot be shared; "
                   "libspectro.so runs on the system runtime and a torch imported later in this process will not see the GPU")
            if want == "torch":
                raise ImportError("SPECTRO_HIP_RUNTIME=torch: " + msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=3)
            return settle("system", None, f"SONAME {soname!r} is not what libspectro.so needs ({mine})")
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        return settle("torch", path, f"SONAME {soname} == libspectro.so's DT_NEEDED")
    except OSError as e:                                     # torch's copy does not load or parse here: the system runtime it is
        if want == "torch":
            raise ImportError(f"SPECTRO_HIP_RUNTIME=torch: {e}") from e
        return settle("system", None, f"torch's copy is unusable here ({e})")


def lib():
    """Load libspectro.so once; raises ImportError with build instructions if it is absent."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        f"{LIB_PATH} not found: build it with `python spectrogram-generator_amd/build.py` "
                        "(hipcc, gfx950).  There is no CPU fallback.")
                _share_torchs_hip_runtime()
                L = C.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(L, name)       # AttributeError = header/library drift
                    fn.restype, fn.argtypes = res, args
                _lib = L
    return _lib


def last_error() -> str:
    return lib().sg_last_error().decode("utf-8", "replace")


def check(rc: int):
    """Map sg_status to the exceptions the reference's GUI handlers expect."""
    if rc == SG_OK:
        return
    msg = last_error()
    if rc == SG_ERR_ARG:
        raise ValueError(msg)
    if rc == SG_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise SpectroError(f"libspectro status {rc}: {msg}")


def ensure_device(device: int | None = None):
    """Select the device (default: LOCAL_RANK or 0) and verify it is gfx950.  Raises if none."""
    global _device_ready
    if device is None:
        if _device_ready is not None:
            return _device_ready
        device = int(os.environ.get("SPECTRO_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if _device_ready is not None and int(device) != _device_ready:
        device_pool_clear()                     # pooled blocks belong to the device they were allocated on
    check(lib().sg_init(int(device)))
    _device_ready = int(device)
    return _device_ready


def device_info():
    buf = C.create_string_buffer(64)
    cu, mem = _i(0), C.c_uint64(0)
    check(lib().sg_device_info(buf, 64, C.byref(cu), C.byref(mem)))
    return {"arch": buf.value.decode(), "compute_units": cu.value, "hbm_bytes": mem.value}


def mem_info():
    """(free, total) bytes of the current device's memory right now"""
    f, t = C.c_uint64(0), C.c_uint64(0)
    check(lib().sg_mem_info(C.byref(f), C.byref(t)))
    return f.value, t.value


def device_pci_bus_id() -> str:
    buf = C.create_string_buffer(32)
    check(lib().sg_device_pci_bus_id(buf, 32))
    return buf.value.decode()


# Device memory comes from a small size-bucketed pool: every shim call used to pay hipMalloc + hipFree, and hipFree
# synchronises the whole device -- the dominant cost of a GUI-sized call.  Buffers are recycled in issue order on the
# stream they were used on (the shim's calls are stream-ordered on the default stream; the pipelined path keeps its own
# per-stream workspace), so a recycled block is never touched by a kernel that is still running on it.
_POOL_MAX_BYTES = int(os.environ.get("SPECTRO_POOL_BYTES", str(2 << 30)))      # 0 disables pooling
_pool_free: dict = {}             # capacity -> [ptr, ...]
_pool_bytes = 0
_pool_lock = threading.Lock()


def _bucket(nbytes: int) -> int:
    """capacity class: powers of two up to 1 MiB, then multiples of 1 MiB (little slack on big spectra)"""
    n = max(int(nbytes), 256)
    if n <= (1 << 20):
        return 1 << (n - 1).bit_length()
    return (n + (1 << 20) - 1) & ~((1 << 20) - 1)


def device_pool_clear():
    """hipFree every pooled block and the library's per-stream workspaces (memory-tight callers, device switches, tests)."""
    global _pool_bytes
    with _pool_lock:
        for ptrs in _pool_free.values():
            for p in ptrs:
                lib().sg_free(C.c_void_p(p))
        _pool_free.clear()
        _pool_bytes = 0
    lib().sg_workspace_release()


class DeviceBuffer:
    """Owning handle of raw device memory obtained through sg_malloc (recycled through the pool above).

    Rule of the pool: a parked block may be handed to the next caller at once, and that caller's work runs on the DEFAULT stream.  A
    buffer that was last used on another stream (``Plan.stft(..., stream=s)``, the pipelined path's non-blocking streams) must
    therefore be freed with ``free(stream=s)``: the stream is synchronised before the block is parked.  ``free()`` without a stream
    is for buffers used on the default stream only (``__del__`` parks that way as well)."""

    def __init__(self, nbytes: int):
        global _pool_bytes
        self.nbytes = int(nbytes)
        self.capacity = _bucket(self.nbytes) if _POOL_MAX_BYTES else max(self.nbytes, 1)
        if _POOL_MAX_BYTES:
            with _pool_lock:
                free = _pool_free.get(self.capacity)
                if free:
                    self.ptr = free.pop()
                    _pool_bytes -= self.capacity
                    return
        p = C.c_void_p()
        rc = lib().sg_malloc(C.byref(p), self.capacity)
        if rc != SG_OK and _POOL_MAX_BYTES:       # out of memory with blocks parked in the pool: release them and retry once
            device_pool_clear()
            rc = lib().sg_malloc(C.byref(p), self.capacity)
        check(rc)
        self.ptr = p.value

    def free(self, stream=None):
        global _pool_bytes
        if getattr(self, "ptr", None):
            ptr, self.ptr = self.ptr, None
            if stream is not None:
                lib().sg_stream_sync(C.c_void_p(stream))     # work queued on that stream may still touch the block
            if _POOL_MAX_BYTES:
                with _pool_lock:
                    if _pool_bytes + self.capacity <= _POOL_MAX_BYTES:
                        _pool_free.setdefault(self.capacity, []).append(ptr)
                        _pool_bytes += self.capacity
                        return
            lib().sg_free(C.c_void_p(ptr))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, arr, stream=None):
        import numpy as np
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        check(lib().sg_memcpy_h2d(C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, C.c_void_p(stream)))
        return a   # keep alive until the stream is synchronised

    def download(self, arr, stream=None, nbytes=None):
        n = arr.nbytes if nbytes is None else nbytes
        check(lib().sg_memcpy_d2h(arr.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), n, C.c_void_p(stream)))
        return arr


def stream_sync(stream=None):
    check(lib().sg_stream_sync(C.c_void_p(stream)))


class Plan:
    """RAII wrapper of sg_plan (argument triage + device tables for one nperseg/nfft/hop/window)."""

    def __init__(self, nperseg, nfft, hop, window, detrend, fs, scaling, mode, dtype):
        import numpy as np
        ensure_device()
        w = np.ascontiguousarray(window, dtype=np.float64)
        if w.ndim != 1 or w.shape[0] != nperseg:
            raise ValueError("window must be 1-D of length nperseg")
        self.nperseg, self.nfft, self.hop = int(nperseg), int(nfft), int(hop)
        self.dtype = int(dtype)
        self.mode = int(mode)
        h = C.c_void_p()
        check(lib().sg_plan_create(C.byref(h), self.nperseg, self.nfft, self.hop,
                                   w.ctypes.data_as(C.POINTER(_d)), int(detrend), float(fs), int(scaling),
                                   int(mode), self.dtype))
        self.handle = h

    @property
    def n_bins(self):
        return self.nfft // 2 + 1

    @property
    def kernel(self):
        return lib().sg_plan_kernel(self.handle).decode()

    @property
    def scale(self):
        v = _d(0)
        check(lib().sg_plan_scale(self.handle, C.byref(v)))
        return v.value

    def force_kernel(self, name: str):
        check(lib().sg_plan_force_kernel(self.handle, name.encode()))

    def n_frames(self, n_samples: int) -> int:
        if n_samples < self.nperseg:
            return 0
        return (int(n_samples) - self.nperseg) // self.hop + 1

    def stft(self, x_ptr, n_samples, clip_stride, n_clips, out_ptr, out_clip_stride, stream=None, int16=False):
        fn = lib().sg_stft_i16 if int16 else lib().sg_stft
        check(fn(self.handle, C.c_void_p(x_ptr), int(n_samples), int(clip_stride), int(n_clips),
                 C.c_void_p(out_ptr), int(out_clip_stride), C.c_void_p(stream)))

    def band_power(self, x_ptr, n_samples, clip_stride, n_clips, k_lo, k_hi, out_ptr, out_clip_stride, stream=None):
        check(lib().sg_stft_band_power(self.handle, C.c_void_p(x_ptr), int(n_samples), int(clip_stride), int(n_clips),
                                       int(k_lo), int(k_hi), C.c_void_p(out_ptr), int(out_clip_stride),
                                       C.c_void_p(stream)))

    def stft_db(self, x_ptr, n_samples, clip_stride, n_clips, k_lo, k_hi, global_max, db_ptr, out_clip_stride, mm_ptr, stream=None):
        check(lib().sg_stft_db(self.handle, C.c_void_p(x_ptr), int(n_samples), int(clip_stride), int(n_clips), int(k_lo), int(k_hi),
                               float(global_max), C.c_void_p(db_ptr), int(out_clip_stride), C.c_void_p(mm_ptr), C.c_void_p(stream)))

    def time_stft(self, x_ptr, n_samples, clip_stride, n_clips, out_ptr, out_clip_stride, iters, stream=None):
        ms = C.c_float(0)
        check(lib().sg_time_stft(self.handle, C.c_void_p(x_ptr), int(n_samples), int(clip_stride), int(n_clips),
                                 C.c_void_p(out_ptr), int(out_clip_stride), C.c_void_p(stream), int(iters),
                                 C.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, "handle", None):
            lib().sg_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def freqs(nfft: int, fs: float):
    import numpy as np
    out = np.empty(nfft // 2 + 1, np.float64)
    check(lib().sg_freqs(int(nfft), float(fs), out.ctypes.data_as(C.POINTER(_d))))
    return out


def times(n_samples: int, nperseg: int, hop: int, fs: float):
    import numpy as np
    n = 0 if n_samples < nperseg else (n_samples - nperseg) // hop + 1
    out = np.empty(n, np.float64)
    if n:
        check(lib().sg_times(int(n_samples), int(nperseg), int(hop), float(fs), out.ctypes.data_as(C.POINTER(_d))))
    return out
