#!/usr/bin/env python3
"""Build libspectro.so (HIP, gfx950 only) in-tree.

    python spectrogram-generator_amd/build.py [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so lands in spectrogram-generator_amd/lib/
(git-ignored, but it travels to the GPU box with the repo snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libspectro.so")
SOURCES = ["host_shim.cpp", "spectro_api.hip", "stft_r8x3.hip", "stft_r8x3_f64.hip", "stft_rsmall.hip", "stft_rbig.hip", "stft_rbig_f64.hip", "stft_stockham.hip", "stft_bluestein.hip", "stft_rblue.hip", "stft_rblue_f64.hip", "stft_rbluew.hip", "stft_rbluew_f64.hip", "stft_rtiny.hip", "epilogue.hip", "mel.hip", "stft_mel_fused.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
if os.environ.get("SG_TUNING") == "1":          # the launch-time tuning variables of the A/B tools (csrc/spectro_internal.h: SG_TUNE_ENV); never in the product build
    FLAGS.append("-DSG_TUNING=1")
# Per-file extras.  stft_r8x3 is VALU-bound: gfx950 issues v_pk_*_f32 at half the rate of the plain ops
# (tools/ubench/valu_rate.hip: 2.2 ns vs 1.2 ns per wave-instruction per SIMD), so SLP packing only adds
# register-pair shuffles (v_pk_mov/v_mov) -- keep the scalar forms.
EXTRA = {"mel.hip": os.environ.get("SG_MEL_DEFS", "").split(),
         "stft_r8x3.hip": (["-fno-slp-vectorize"] if not os.environ.get("SG_SLP") else []) + os.environ.get("SG_R8_DEFS", "").split(),
         "stft_rsmall.hip": ["-fno-slp-vectorize"] + os.environ.get("SG_RSMALL_DEFS", "").split(), "stft_rbig.hip": ["-fno-slp-vectorize"] + os.environ.get("SG_RBIG_DEFS", "").split(), "stft_rblue.hip": ["-fno-slp-vectorize"] + os.environ.get("SG_RBLUE_DEFS", "").split(), "stft_rbluew.hip": ["-fno-slp-vectorize"] + os.environ.get("SG_RBLUEW_DEFS", "").split(), "stft_mel_fused.hip": ["-fno-slp-vectorize"] + os.environ.get("SG_FUSED_DEFS", "").split()}


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps():
    out = [os.path.join(CSRC, s) for s in SOURCES]
    out += [os.path.join(CSRC, "fft_wave.h"), os.path.join(CSRC, "spectro_internal.h"), os.path.join(CSRC, "host_shim.h"), os.path.join(ROOT, "include", "spectro.h"), __file__]
    return out


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(d) <= t for d in _deps())


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    t0 = time.time()
    procs = []
    for s in SOURCES:
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o").replace(".cpp", ".o"))
        flags = [f for f in FLAGS if not f.startswith(("--offload-arch", "-fno-gpu-rdc"))] + ["-x", "c++"] if s.endswith(".cpp") else FLAGS
        cmd = [hipcc(), *flags, *EXTRA.get(s, []), "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out.strip()):
            print(f"--- {s} ---\n{out}", flush=True)
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout)
        raise RuntimeError("link failed")
    if verbose:
        print(f"built {LIB} in {time.time() - t0:.1f}s")
    return LIB


C_CLIENT_SRC = os.path.join(ROOT, "examples", "c_client.c")
C_CLIENT = os.path.join(ROOT, "examples", "c_client")


def build_c_client(force=False):
    """examples/c_client: a plain C99 program against include/spectro.h -- proves the header is C-clean and the
    library links without Python, torch or HIP headers on the caller's side (gcc only)."""
    lib = build()
    if not force and os.path.exists(C_CLIENT) and os.path.getmtime(C_CLIENT) >= max(os.path.getmtime(C_CLIENT_SRC), os.path.getmtime(lib)):
        return C_CLIENT
    cmd = ["gcc", "-O2", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), C_CLIENT_SRC,
           "-L", LIBDIR, "-lspectro", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,$ORIGIN/../spectrogram-generator_amd/lib",
           "-o", C_CLIENT]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout)
        raise RuntimeError("gcc failed on examples/c_client.c")
    return C_CLIENT


SAN_DRIVER_SRC = os.path.join(ROOT, "tests", "asan_driver.cpp")
SAN_DRIVER = os.path.join(LIBDIR, "host_shim_asan")


def build_sanitizer_driver(force=False):
    """tests/asan_driver.cpp + csrc/host_shim.cpp under g++ -fsanitize=address,undefined (CPU only, no HIP): the host
    side of the ABI -- argument triage, f / t vectors, mel bank construction, jet table, error strings."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [SAN_DRIVER_SRC, os.path.join(CSRC, "host_shim.cpp")]
    deps = srcs + [os.path.join(CSRC, "host_shim.h"), os.path.join(ROOT, "include", "spectro.h")]
    if not force and os.path.exists(SAN_DRIVER) and os.path.getmtime(SAN_DRIVER) >= max(os.path.getmtime(d) for d in deps):
        return SAN_DRIVER
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", CSRC, *srcs, "-o", SAN_DRIVER]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout)
        raise RuntimeError("g++ -fsanitize failed on the host shim")
    return SAN_DRIVER


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_c_client(force="--force" in sys.argv)
