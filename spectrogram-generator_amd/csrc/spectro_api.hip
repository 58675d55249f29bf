// spectro_api.hip -- the C ABI of libspectro.so (see include/spectro.h): plans, dispatch, memory helpers.
#include "spectro_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

namespace sg {

int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", static_cast<int>(e), hipGetErrorString(e), what);
    return SG_ERR_HIP;
}

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static size_t esize(int dtype) { return dtype == SG_F64 ? 8 : 4; }

static bool r8x3_ok(const sg_plan& p) {
    return p.dtype == SG_F32 && p.nperseg == 1024 && p.nfft == 1024 &&
           (p.detrend == SG_DETREND_NONE || p.detrend == SG_DETREND_CONSTANT) &&
           (p.mode == SG_MODE_PSD || p.mode == SG_MODE_MAGNITUDE);
}

static bool r8x3d_ok(const sg_plan& p) {
    return p.dtype == SG_F64 && p.nperseg == p.nfft && (p.nfft == 128 || p.nfft == 256 || p.nfft == 512 || p.nfft == 1024) &&
           (p.detrend == SG_DETREND_NONE || p.detrend == SG_DETREND_CONSTANT) &&
           (p.mode == SG_MODE_PSD || p.mode == SG_MODE_MAGNITUDE);
}

static bool rsmall_ok(const sg_plan& p) {
    return p.dtype == SG_F32 && p.nperseg == p.nfft && (p.nfft == 128 || p.nfft == 256 || p.nfft == 512) &&
           (p.detrend == SG_DETREND_NONE || p.detrend == SG_DETREND_CONSTANT) &&
           (p.mode == SG_MODE_PSD || p.mode == SG_MODE_MAGNITUDE);
}

// nperseg 32 / 64, f32 and f64: the quad-DPP register kernel (stft_rtiny.hip)
static bool rtiny_ok(const sg_plan& p) {
    const bool small_np2 = p.nfft == 96 || p.nfft == 160 || p.nfft == 192 || p.nfft == 224;     // Q = 6 / 10 / 12 / 14 lanes per frame
    return (p.dtype == SG_F32 || p.dtype == SG_F64) && p.nperseg == p.nfft && (p.nfft == 32 || p.nfft == 64 || small_np2) &&
           (p.detrend == SG_DETREND_NONE || p.detrend == SG_DETREND_CONSTANT) &&
           (p.mode == SG_MODE_PSD || p.mode == SG_MODE_MAGNITUDE);
}

static bool rbig_ok(const sg_plan& p) {
    return p.dtype == SG_F32 && p.nperseg == p.nfft && (p.nfft == 2048 || p.nfft == 4096) &&
           (p.detrend == SG_DETREND_NONE || p.detrend == SG_DETREND_CONSTANT) &&
           (p.mode == SG_MODE_PSD || p.mode == SG_MODE_MAGNITUDE);
}

static bool rbigd_ok(const sg_plan& p) {
    return p.dtype == SG_F64 && p.nperseg == p.nfft && (p.nfft == 2048 || p.nfft == 4096) &&
           (p.detrend == SG_DETREND_NONE || p.detrend == SG_DETREND_CONSTANT) &&
           (p.mode == SG_MODE_PSD || p.mode == SG_MODE_MAGNITUDE);
}

// even transform lengths that are not a power of two, up to 2048: the register chirp-z kernel (stft_rblue.hip)
static bool rblue_ok(const sg_plan& p) {
    return p.dtype == SG_F32 && p.nperseg == p.nfft && p.nfft % 2 == 0 && !is_pow2(p.nfft) && p.nfft >= 6 && p.nfft <= 2048 &&
           (p.detrend == SG_DETREND_NONE || p.detrend == SG_DETREND_CONSTANT) &&
           (p.mode == SG_MODE_PSD || p.mode == SG_MODE_MAGNITUDE);
}

// ... and in double precision up to 1024 (stft_rblue_f64.hip): the reference's recordings are float64
static bool rblued_ok(const sg_plan& p) {
    return p.dtype == SG_F64 && p.nperseg == p.nfft && p.nfft % 2 == 0 && !is_pow2(p.nfft) && p.nfft >= 6 && p.nfft <= 1024 &&
           (p.detrend == SG_DETREND_NONE || p.detrend == SG_DETREND_CONSTANT) &&
           (p.mode == SG_MODE_PSD || p.mode == SG_MODE_MAGNITUDE);
}

// ... and from 2048 to 8192 with two / four wavefronts per frame (stft_rbluew.hip): nperseg a multiple of 4 / 8
static bool rbluew_ok(const sg_plan& p) {
    // (8192 itself too: a chirp-z transform over four waves is 2.2 x the Stockham kernel there, profiles/r04_pow2_edges.txt)
    return p.dtype == SG_F32 && p.nperseg == p.nfft && (!is_pow2(p.nfft) || p.nfft == 8192) && p.nfft > 2048 && p.nfft <= 8192 &&
           p.nfft % (2 * rbluew_size(p.nfft)) == 0 &&
           (p.detrend == SG_DETREND_NONE || p.detrend == SG_DETREND_CONSTANT) &&
           (p.mode == SG_MODE_PSD || p.mode == SG_MODE_MAGNITUDE);
}

// ... in double precision from 1024: two / four / eight wavefronts per frame (stft_rbluew_f64.hip), nperseg a multiple of 4 / 8 / 16
static bool rbluewd_ok(const sg_plan& p) {
    return p.dtype == SG_F64 && p.nperseg == p.nfft && (!is_pow2(p.nfft) || p.nfft == 8192) && p.nfft > 1024 && p.nfft <= 8192 &&
           p.nfft % (2 * rbluew_f64_size(p.nfft)) == 0 &&
           (p.detrend == SG_DETREND_NONE || p.detrend == SG_DETREND_CONSTANT) &&
           (p.mode == SG_MODE_PSD || p.mode == SG_MODE_MAGNITUDE);
}

static bool stockham_ok(const sg_plan& p) {
    if (!is_pow2(p.nfft) || p.nfft < 2) return false;
    // LDS need of the largest case: one frame per workgroup, two nfft-real buffers + reduction scratch
    const int M = p.nfft / 2;
    int tpf = M / 2 < 1 ? 1 : (M / 2 > 256 ? 256 : M / 2);
    const size_t lds = static_cast<size_t>(256 / tpf) * (2 * static_cast<size_t>(p.nfft) * esize(p.dtype) + 2 * tpf * 8);
    return lds <= 160 * 1024;
}

template <typename T>
static int upload(void** dev, const std::vector<T>& host) {
    SG_HIP(hipMalloc(dev, host.size() * sizeof(T)));
    SG_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return SG_OK;
}

template <typename T>
static int build_common_tables(sg_plan& p, const std::vector<double>& window) {
    std::vector<T> w(window.size());
    for (size_t i = 0; i < window.size(); ++i) w[i] = static_cast<T>(window[i]);
    // scipy:2083-2089: window cast to the data precision, scale computed in that precision
    double acc = 0.0;
    if (p.scaling == SG_SCALING_DENSITY) {
        for (T v : w) acc += static_cast<double>(v) * static_cast<double>(v);
        const T sum = static_cast<T>(acc);
        const T prod = static_cast<T>(p.fs) * sum;
        p.scale = static_cast<double>(static_cast<T>(T(1) / prod));
    } else {
        for (T v : w) acc += static_cast<double>(v);
        const T sum = static_cast<T>(acc);
        p.scale = static_cast<double>(static_cast<T>(T(1) / (sum * sum)));
    }
    if (int rc = upload<T>(&p.win_dev, w)) return rc;
    if (is_pow2(p.nfft) && p.nfft >= 2) {
        const int half = p.nfft / 2;
        std::vector<T> tw(2 * static_cast<size_t>(half));
        const long double two_pi = 6.283185307179586476925286766559005768L;
        for (int k = 0; k < half; ++k) {
            const long double a = -two_pi * static_cast<long double>(k) / static_cast<long double>(p.nfft);
            tw[2 * k] = static_cast<T>(cosl(a));
            tw[2 * k + 1] = static_cast<T>(sinl(a));
        }
        if (int rc = upload<T>(&p.tw_dev, tw)) return rc;
    }
    return SG_OK;
}

static void free_tables(sg_plan* p) {
    void** ptrs[] = {&p->win_dev, &p->tw_dev, &p->r8_tw_dev, &p->r8_win_dev, &p->bs_chirp_dev, &p->bs_filter_dev, &p->bs_tw_dev,
                     &p->rb_wc_dev, &p->rb_filt_dev, &p->rb_stw_dev, &p->rb_tw_dev};
    for (void** q : ptrs) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
}

// int16 PCM -> float, exact (|v| <= 32768): the front end of the register kernels that have no int16 loads of their own
__global__ __launch_bounds__(256) void convert_i16_kernel(const int16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(src) & 7) | (reinterpret_cast<uintptr_t>(dst) & 15)) == 0) {
        const short4* const s4 = reinterpret_cast<const short4*>(src);
        float4* const d4 = reinterpret_cast<float4*>(dst);
        for (int64_t j = i; j < n / 4; j += stride) {
            const short4 v = s4[j];
            d4[j] = make_float4(static_cast<float>(v.x), static_cast<float>(v.y), static_cast<float>(v.z), static_cast<float>(v.w));
        }
        for (int64_t j = (n / 4) * 4 + i; j < n; j += stride) dst[j] = static_cast<float>(src[j]);
    } else {
        for (; i < n; i += stride) dst[i] = static_cast<float>(src[i]);
    }
}

static int convert_i16(const int16_t* src, float* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return SG_OK;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(convert_i16_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, src, dst, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "convert_i16 launch");
}

static int run_stft(const sg_plan* plan, StftArgs& a);

// rsmall / rbig batches of int16 PCM: convert once into the stream's float workspace (same clip stride, so the register
// kernel's alignment rules are unchanged) and run the float kernel -- ~10x the LDS kernel's rate.  Small calls (a GUI sweep)
// keep the LDS kernel's own int16 loads.
static int run_converted(const sg_plan* plan, StftArgs& a) {
    const int64_t span = static_cast<int64_t>(a.n_clips - 1) * a.clip_stride + a.n_samples;
    std::lock_guard<std::recursive_mutex> seq(launch_sequence_mutex(a.stream));
    void* const work = stream_workspace(a.stream, static_cast<size_t>(span) * sizeof(float));
    if (!work) { set_error("sg_stft_i16: no memory for the %lld-sample float workspace", static_cast<long long>(span)); return SG_ERR_HIP; }
    int rc = convert_i16(static_cast<const int16_t*>(a.x), static_cast<float*>(work), span, a.stream);
    if (rc == SG_OK) {
        StftArgs f = a;
        f.x = work;
        f.in_i16 = 0;
        rc = run_stft(plan, f);
    }
    return rc;
}

// A chirp-z plan's LDS kernel: its tables are built the first time a call needs them (the register kernel takes practically every call).
static int launch_bluestein_lazy(const sg_plan* plan, const StftArgs& a) {
    static std::mutex mu;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!plan->bs_chirp_dev) { if (int rc = build_bluestein_tables(*const_cast<sg_plan*>(plan))) return rc; }
    }
    return launch_bluestein(*plan, a);
}

// The fused band power of a chirp-z plan on a call its register kernel cannot take (odd hop, unaligned clips): full spectra by the LDS
// kernel into a block of their own, then the band sums.  A rare path (the register kernel serves every even hop): the block is a plain
// hipMalloc / hipFree pair -- the stream's workspace may be holding the float copy of an int16 call around this one.
static int band_via_spectrum(const sg_plan* plan, StftArgs& a) {
    const int nbins = plan->nfft / 2 + 1;
    void* spec = nullptr;
    SG_HIP(hipMalloc(&spec, static_cast<size_t>(a.n_clips) * a.n_frames * nbins * sizeof(float)));
    StftArgs f = a;
    f.band_mode = 0;
    f.out = spec;
    f.out_clip_stride = a.n_frames * nbins;
    int rc = launch_bluestein_lazy(plan, f);
    for (int c = 0; rc == SG_OK && c < a.n_clips; ++c)
        rc = sg_band_sum(static_cast<const float*>(spec) + static_cast<int64_t>(c) * a.n_frames * nbins, SG_F32, a.n_frames, nbins, a.k_lo, a.k_hi,
                         static_cast<float*>(a.out) + static_cast<int64_t>(c) * a.out_clip_stride, a.stream);
    if (rc == SG_OK) { const hipError_t e = hipStreamSynchronize(a.stream); if (e != hipSuccess) rc = hip_fail(e, "band_via_spectrum"); }
    (void)hipFree(spec);
    return rc;
}

static int run_stft(const sg_plan* plan, StftArgs& a) {
    if (!plan) { set_error("null plan"); return SG_ERR_ARG; }
    if (a.n_clips < 0 || a.n_samples < 0) { set_error("negative sizes"); return SG_ERR_ARG; }
    a.n_frames = a.n_samples < plan->nperseg ? 0 : (a.n_samples - plan->nperseg) / plan->hop + 1;
    if (a.n_frames == 0 || a.n_clips == 0) return SG_OK;
    if (!a.x || !a.out) { set_error("null device pointer"); return SG_ERR_ARG; }
    if (a.n_clips > 1 && a.clip_stride < a.n_samples) { set_error("clip_stride < n_samples"); return SG_ERR_ARG; }
    const int nbins = plan->nfft / 2 + 1;
    const int64_t need = a.band_mode ? a.n_frames : a.n_frames * nbins * (plan->mode == SG_MODE_COMPLEX ? 2 : 1);
    if (a.n_clips > 1 && a.out_clip_stride < need) { set_error("out_clip_stride %lld < %lld", (long long)a.out_clip_stride, (long long)need); return SG_ERR_ARG; }
    if (a.band_mode) {
        if (plan->mode != SG_MODE_PSD) { set_error("band power needs a psd plan"); return SG_ERR_ARG; }
        if (a.k_lo < 0 || a.k_hi >= nbins || a.k_lo > a.k_hi) { set_error("bad band [%d,%d] of %d bins", a.k_lo, a.k_hi, nbins); return SG_ERR_ARG; }
    }
    if (a.in_i16 && (plan->kernel == Kernel::RSMALL || plan->kernel == Kernel::RBIG || (plan->kernel == Kernel::RTINY && plan->dtype == SG_F32)) && plan->hop % 2 == 0 &&
        (a.clip_stride % 2 == 0 || a.n_clips == 1) && static_cast<int64_t>(a.n_clips) * a.n_samples >= (1 << 18))
        return run_converted(plan, a);
    if ((plan->kernel == Kernel::RBLUE || plan->kernel == Kernel::RBLUEW || (plan->kernel == Kernel::RTINY && !is_pow2(plan->nfft) && plan->dtype == SG_F32)) && a.in_i16)
        return run_converted(plan, a);                       // no chirp-z kernel loads int16 (nor does rtiny, whose fallback at these sizes is one)
    switch (plan->kernel) {
        case Kernel::R8X3: return launch_r8x3(*plan, a);
        case Kernel::R8X3D: return r8x3_f64_can_run(*plan, a) ? launch_r8x3_f64(*plan, a) : launch_stockham(*plan, a);
        case Kernel::RSMALL: return rsmall_can_run(*plan, a) ? launch_rsmall(*plan, a) : launch_stockham(*plan, a);
        case Kernel::RBIG: return rbig_can_run(*plan, a) ? launch_rbig(*plan, a) : launch_stockham(*plan, a);
        case Kernel::RBIGD: return rbig_f64_can_run(*plan, a) ? launch_rbig_f64(*plan, a) : launch_stockham(*plan, a);
        case Kernel::RTINY:
            if (rtiny_can_run(*plan, a)) return launch_rtiny(*plan, a);
            if (is_pow2(plan->nfft)) return launch_stockham(*plan, a);
            // nperseg 96 / 160 / 192 / 224 on an odd hop or an unaligned clip: the LDS chirp-z kernel
            if (a.band_mode && plan->dtype == SG_F64) { set_error("band power of this plan needs an even hop and 16-byte aligned float64 input"); return SG_ERR_UNSUPPORTED; }
            return a.band_mode ? band_via_spectrum(plan, a) : launch_bluestein_lazy(plan, a);
        case Kernel::STOCKHAM: return launch_stockham(*plan, a);
        case Kernel::BLUESTEIN: return launch_bluestein(*plan, a);
        // odd hops / unaligned clips, GUI-sized int16 calls: the LDS chirp-z kernel (its tables are built with the plan); it writes full spectra only
        // (more than 2^31 frames per clip: the LDS chirp-z kernel, whose tables this plan builds on first need)
        case Kernel::RBLUED:
            if (rblue_f64_can_run(*plan, a)) return launch_rblue_f64(*plan, a);
            if (a.band_mode) { set_error("band power of this chirp-z plan needs 8-byte aligned float64 input"); return SG_ERR_UNSUPPORTED; }
            return launch_bluestein_lazy(plan, a);
        case Kernel::RBLUEWD:
            if (rbluew_f64_can_run(*plan, a)) return launch_rbluew_f64(*plan, a);
            if (a.band_mode) { set_error("band power of this chirp-z plan needs 8-byte aligned float64 input"); return SG_ERR_UNSUPPORTED; }
            return launch_bluestein_lazy(plan, a);
        case Kernel::RBLUE: return rblue_can_run(*plan, a) ? launch_rblue(*plan, a) : a.band_mode ? band_via_spectrum(plan, a) : launch_bluestein_lazy(plan, a);
        case Kernel::RBLUEW: return rbluew_can_run(*plan, a) ? launch_rbluew(*plan, a) : a.band_mode ? band_via_spectrum(plan, a) : launch_bluestein_lazy(plan, a);
    }
    return SG_ERR_UNSUPPORTED;
}

}  // namespace sg

using namespace sg;

extern "C" {

int sg_device_count(int* count) {
    if (!count) { set_error("null pointer"); return SG_ERR_ARG; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return hip_fail(e, "hipGetDeviceCount"); }
    *count = n;
    return SG_OK;
}

int sg_init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible"); return SG_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { set_error("device %d out of range (%d visible)", device, n); return SG_ERR_ARG; }
    SG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SG_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libspectro.so carries gfx950 code only", device, prop.gcnArchName);
        return SG_ERR_NO_DEVICE;
    }
    return SG_OK;
}

int sg_device_info(char* arch, size_t arch_len, int* compute_units, uint64_t* hbm_bytes) {
    int dev = 0;
    SG_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SG_HIP(hipGetDeviceProperties(&prop, dev));
    if (arch && arch_len) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return SG_OK;
}

int sg_mem_info(uint64_t* free_bytes, uint64_t* total_bytes) {
    size_t f = 0, t = 0;
    SG_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return SG_OK;
}

int sg_device_pci_bus_id(char* buf, size_t len) {
    if (!buf || len < 13) { set_error("sg_device_pci_bus_id: buffer of at least 13 bytes needed"); return SG_ERR_ARG; }
    int dev = 0;
    SG_HIP(hipGetDevice(&dev));
    SG_HIP(hipDeviceGetPCIBusId(buf, static_cast<int>(len), dev));
    for (char* c = buf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = static_cast<char>(*c - 'A' + 'a');   // sysfs spells it in lower case
    return SG_OK;
}

int sg_malloc(void** dev_ptr, size_t bytes) {
    if (!dev_ptr) { set_error("null pointer"); return SG_ERR_ARG; }
    SG_HIP(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return SG_OK;
}
int sg_free(void* dev_ptr) { if (dev_ptr) SG_HIP(hipFree(dev_ptr)); return SG_OK; }
int sg_host_alloc(void** host_ptr, size_t bytes) {
    if (!host_ptr) { set_error("null pointer"); return SG_ERR_ARG; }
    SG_HIP(hipHostMalloc(host_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return SG_OK;
}
int sg_host_free(void* host_ptr) { if (host_ptr) SG_HIP(hipHostFree(host_ptr)); return SG_OK; }
int sg_host_register(void* host_ptr, size_t bytes) {
    if (!host_ptr || !bytes) { set_error("null host range"); return SG_ERR_ARG; }
    SG_HIP(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    return SG_OK;
}
int sg_host_unregister(void* host_ptr) { if (host_ptr) SG_HIP(hipHostUnregister(host_ptr)); return SG_OK; }
int sg_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
    if (bytes) SG_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    return SG_OK;
}
int sg_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream) {
    if (bytes) SG_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    return SG_OK;
}
int sg_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream) {
    if (bytes) SG_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return SG_OK;
}
int sg_memcpy2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t height, int kind,
                void* stream) {
    if (kind < 0 || kind > 2) { set_error("sg_memcpy2d: kind must be 0 (h2d), 1 (d2h) or 2 (d2d)"); return SG_ERR_ARG; }
    if (width_bytes > dst_pitch || width_bytes > src_pitch) { set_error("sg_memcpy2d: width exceeds a pitch"); return SG_ERR_ARG; }
    if (!width_bytes || !height) return SG_OK;
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    SG_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, height, k, static_cast<hipStream_t>(stream)));
    return SG_OK;
}
int sg_memset(void* dst_dev, int value, size_t bytes, void* stream) {
    if (bytes) SG_HIP(hipMemsetAsync(dst_dev, value, bytes, static_cast<hipStream_t>(stream)));
    return SG_OK;
}
int sg_stream_create(void** stream) {
    if (!stream) { set_error("null pointer"); return SG_ERR_ARG; }
    hipStream_t s;
    SG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return SG_OK;
}
int sg_stream_destroy(void* stream) {
    if (!stream) return SG_OK;
    auto s = static_cast<hipStream_t>(stream);
    SG_HIP(hipStreamSynchronize(s));
    drop_stream_state(s);                      // its reduction scratch, workspace and launch lock (epilogue.hip)
    SG_HIP(hipStreamDestroy(s));
    return SG_OK;
}
int sg_stream_sync(void* stream) { SG_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream))); return SG_OK; }

int sg_plan_create(sg_plan** plan, int nperseg, int nfft, int hop, const double* window, int detrend, double fs,
                   int scaling, int mode, int dtype) {
    if (!plan || !window) { set_error("null pointer"); return SG_ERR_ARG; }
    *plan = nullptr;
    if (int rc = check_plan_args(nperseg, nfft, hop, detrend, fs, scaling, mode, dtype)) return rc;

    auto* p = new sg_plan();
    p->nperseg = nperseg; p->nfft = nfft; p->hop = hop;
    p->detrend = detrend; p->scaling = scaling; p->mode = mode; p->dtype = dtype; p->fs = fs;
    hipError_t e = hipGetDevice(&p->device);
    if (e != hipSuccess) { delete p; return hip_fail(e, "hipGetDevice"); }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, p->device);
    if (e != hipSuccess) { delete p; return hip_fail(e, "hipGetDeviceProperties"); }
    p->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    std::vector<double> w(window, window + nperseg);
    int rc = dtype == SG_F64 ? build_common_tables<double>(*p, w) : build_common_tables<float>(*p, w);
    if (rc == SG_OK) {
        if (r8x3_ok(*p)) {
            p->kernel = Kernel::R8X3;
            rc = build_r8x3_tables(*p, w);
        } else if (r8x3d_ok(*p)) {
            p->kernel = Kernel::R8X3D;
            rc = build_r8x3_f64_tables(*p);
        } else if (rsmall_ok(*p)) {
            p->kernel = Kernel::RSMALL;
            rc = build_rsmall_tables(*p);
        } else if (rtiny_ok(*p)) {
            p->kernel = Kernel::RTINY;
            rc = build_rtiny_tables(*p);
        } else if (rbig_ok(*p)) {
            p->kernel = Kernel::RBIG;
            rc = build_rbig_tables(*p);
        } else if (rbigd_ok(*p)) {
            p->kernel = Kernel::RBIGD;
            rc = build_rbig_f64_tables(*p);
        } else if (stockham_ok(*p) && !(p->nfft == 8192 && (rbluew_ok(*p) || rbluewd_ok(*p)))) {
            p->kernel = Kernel::STOCKHAM;
        } else if (rblued_ok(*p)) {
            p->kernel = Kernel::RBLUED;
            rc = build_rblue_f64_tables(*p, w);
        } else if (rblue_ok(*p)) {
            p->kernel = Kernel::RBLUE;
            rc = build_rblue_tables(*p, w);                   // (the LDS kernel's tables are built on first need: launch_bluestein_lazy)
        } else if (rbluew_ok(*p)) {
            p->kernel = Kernel::RBLUEW;
            rc = build_rbluew_tables(*p, w);
        } else if (rbluewd_ok(*p)) {
            p->kernel = Kernel::RBLUEWD;
            rc = build_rbluew_f64_tables(*p, w);
        } else {
            p->kernel = Kernel::BLUESTEIN;
            rc = build_bluestein_tables(*p);
        }
    }
    if (rc != SG_OK) { free_tables(p); delete p; return rc; }
    *plan = p;
    return SG_OK;
}

int sg_plan_destroy(sg_plan* plan) {
    if (!plan) return SG_OK;
    free_tables(plan);
    delete plan;
    return SG_OK;
}

int sg_plan_n_frames(const sg_plan* plan, int64_t n_samples, int64_t* n_frames) {
    if (!plan || !n_frames) { set_error("null pointer"); return SG_ERR_ARG; }
    *n_frames = n_samples < plan->nperseg ? 0 : (n_samples - plan->nperseg) / plan->hop + 1;
    return SG_OK;
}

int sg_plan_n_bins(const sg_plan* plan, int* n_bins) {
    if (!plan || !n_bins) { set_error("null pointer"); return SG_ERR_ARG; }
    *n_bins = plan->nfft / 2 + 1;
    return SG_OK;
}

int sg_plan_scale(const sg_plan* plan, double* scale) {
    if (!plan || !scale) { set_error("null pointer"); return SG_ERR_ARG; }
    *scale = plan->scale;
    return SG_OK;
}

const char* sg_plan_kernel(const sg_plan* plan) {
    if (!plan) return "";
    switch (plan->kernel) {
        case Kernel::R8X3: return "r8x3";
        case Kernel::R8X3D: return plan->nfft == 1024 ? "r8x3d" : "rsmalld";
        case Kernel::RSMALL: return "rsmall";
        case Kernel::RBIG: return "rbig";
        case Kernel::RBIGD: return "rbigd";
        case Kernel::STOCKHAM: return "stockham";
        case Kernel::BLUESTEIN: return "bluestein";
        case Kernel::RBLUE: return "rblue";
        case Kernel::RBLUED: return "rblued";
        case Kernel::RTINY: return plan->dtype == SG_F64 ? "rtinyd" : "rtiny";
        case Kernel::RBLUEW: return "rbluew";
        case Kernel::RBLUEWD: return "rbluewd";
    }
    return "";
}

int sg_plan_force_kernel(sg_plan* plan, const char* name) {
    if (!plan || !name) { set_error("null pointer"); return SG_ERR_ARG; }
    if (!strcmp(name, "r8x3")) {
        if (!r8x3_ok(*plan)) { set_error("plan cannot run on r8x3"); return SG_ERR_UNSUPPORTED; }
        if (!plan->r8_tw_dev) { std::vector<double> none; if (int rc = build_r8x3_tables(*plan, none)) return rc; }
        plan->kernel = Kernel::R8X3;
        return SG_OK;
    }
    if (!strcmp(name, "r8x3d") || !strcmp(name, "rsmalld")) {
        if (!r8x3d_ok(*plan) || (plan->nfft == 1024) != !strcmp(name, "r8x3d")) { set_error("plan cannot run on %s", name); return SG_ERR_UNSUPPORTED; }
        if (!plan->r8_tw_dev) { if (int rc = build_r8x3_f64_tables(*plan)) return rc; }
        plan->kernel = Kernel::R8X3D;
        return SG_OK;
    }
    if (!strcmp(name, "rsmall")) {
        if (!rsmall_ok(*plan)) { set_error("plan cannot run on rsmall"); return SG_ERR_UNSUPPORTED; }
        if (!plan->r8_tw_dev) { if (int rc = build_rsmall_tables(*plan)) return rc; }
        plan->kernel = Kernel::RSMALL;
        return SG_OK;
    }
    if (!strcmp(name, "rbig")) {
        if (!rbig_ok(*plan)) { set_error("plan cannot run on rbig"); return SG_ERR_UNSUPPORTED; }
        if (!plan->r8_tw_dev) { if (int rc = build_rbig_tables(*plan)) return rc; }
        plan->kernel = Kernel::RBIG;
        return SG_OK;
    }
    if (!strcmp(name, "rbigd")) {
        if (!rbigd_ok(*plan)) { set_error("plan cannot run on rbigd"); return SG_ERR_UNSUPPORTED; }
        if (!plan->r8_tw_dev) { if (int rc = build_rbig_f64_tables(*plan)) return rc; }
        plan->kernel = Kernel::RBIGD;
        return SG_OK;
    }
    if (!strcmp(name, "stockham")) {
        if (!stockham_ok(*plan)) { set_error("plan cannot run on stockham"); return SG_ERR_UNSUPPORTED; }
        plan->kernel = Kernel::STOCKHAM;
        return SG_OK;
    }
    if (!strcmp(name, "rblued")) {
        if (!rblued_ok(*plan) || !plan->rb_wc_dev) { set_error("plan cannot run on rblued"); return SG_ERR_UNSUPPORTED; }
        plan->kernel = Kernel::RBLUED;
        return SG_OK;
    }
    if (!strcmp(name, "rblue")) {
        if (!rblue_ok(*plan)) { set_error("plan cannot run on rblue"); return SG_ERR_UNSUPPORTED; }
        if (!plan->rb_wc_dev) { set_error("rblue tables are built with the plan only"); return SG_ERR_UNSUPPORTED; }
        plan->kernel = Kernel::RBLUE;
        return SG_OK;
    }
    if (!strcmp(name, "rtiny") || !strcmp(name, "rtinyd")) {
        if (!rtiny_ok(*plan) || (plan->dtype == SG_F64) != (name[5] == 'd')) { set_error("plan cannot run on %s", name); return SG_ERR_UNSUPPORTED; }
        if (!plan->r8_tw_dev) { if (int rc = build_rtiny_tables(*plan)) return rc; }
        plan->kernel = Kernel::RTINY;
        return SG_OK;
    }
    if (!strcmp(name, "rbluewd")) {
        if (!rbluewd_ok(*plan) || !plan->rb_wc_dev) { set_error("plan cannot run on rbluewd"); return SG_ERR_UNSUPPORTED; }
        plan->kernel = Kernel::RBLUEWD;
        return SG_OK;
    }
    if (!strcmp(name, "rbluew")) {
        if (!rbluew_ok(*plan)) { set_error("plan cannot run on rbluew"); return SG_ERR_UNSUPPORTED; }
        if (!plan->rb_wc_dev) { set_error("rbluew tables are built with the plan only"); return SG_ERR_UNSUPPORTED; }
        plan->kernel = Kernel::RBLUEW;
        return SG_OK;
    }
    if (!strcmp(name, "bluestein")) {
        if (!plan->bs_chirp_dev) { if (int rc = build_bluestein_tables(*plan)) return rc; }
        plan->kernel = Kernel::BLUESTEIN;
        return SG_OK;
    }
    set_error("unknown kernel family '%s'", name);
    return SG_ERR_ARG;
}

int sg_stft(const sg_plan* plan, const void* x_dev, int64_t n_samples, int64_t clip_stride, int n_clips, void* out_dev,
            int64_t out_clip_stride, void* stream) {
    StftArgs a{};
    a.x = x_dev; a.in_i16 = 0; a.n_samples = n_samples; a.clip_stride = clip_stride; a.n_clips = n_clips;
    a.out = out_dev; a.out_clip_stride = out_clip_stride; a.stream = static_cast<hipStream_t>(stream);
    return run_stft(plan, a);
}

int sg_stft_i16(const sg_plan* plan, const int16_t* x_dev, int64_t n_samples, int64_t clip_stride, int n_clips,
                float* out_dev, int64_t out_clip_stride, void* stream) {
    if (plan && plan->dtype != SG_F32) { set_error("int16 input needs an f32 plan"); return SG_ERR_ARG; }
    if (plan && plan->kernel == Kernel::BLUESTEIN) { set_error("int16 input is not wired to the Bluestein path"); return SG_ERR_UNSUPPORTED; }
    StftArgs a{};
    a.x = x_dev; a.in_i16 = 1; a.n_samples = n_samples; a.clip_stride = clip_stride; a.n_clips = n_clips;
    a.out = out_dev; a.out_clip_stride = out_clip_stride; a.stream = static_cast<hipStream_t>(stream);
    return run_stft(plan, a);
}

int sg_convert_i16(const int16_t* src_dev, float* dst_dev, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!src_dev || !dst_dev))) { set_error("sg_convert_i16: bad arguments"); return SG_ERR_ARG; }
    return convert_i16(src_dev, dst_dev, n, static_cast<hipStream_t>(stream));
}

int sg_stft_band_power(const sg_plan* plan, const void* x_dev, int64_t n_samples, int64_t clip_stride, int n_clips,
                       int k_lo, int k_hi, void* band_out_dev, int64_t out_clip_stride, void* stream) {
    if (plan && plan->kernel == Kernel::BLUESTEIN) { set_error("fused band power is not wired to the Bluestein path"); return SG_ERR_UNSUPPORTED; }
    StftArgs a{};
    a.x = x_dev; a.in_i16 = 0; a.n_samples = n_samples; a.clip_stride = clip_stride; a.n_clips = n_clips;
    a.out = band_out_dev; a.out_clip_stride = out_clip_stride; a.band_mode = 1; a.k_lo = k_lo; a.k_hi = k_hi;
    a.stream = static_cast<hipStream_t>(stream);
    return run_stft(plan, a);
}

int sg_stft_db(const sg_plan* plan, const float* x_dev, int64_t n_samples, int64_t clip_stride, int n_clips, int k_lo, int k_hi,
               double global_max, float* db_dev, int64_t out_clip_stride, float* mm_dev, void* stream) {
    if (!plan) { set_error("null plan"); return SG_ERR_ARG; }
    if (plan->kernel != Kernel::R8X3 || plan->mode != SG_MODE_PSD) {
        set_error("sg_stft_db needs an f32 nperseg = nfft = 1024 psd plan (r8x3); use sg_stft + sg_normalise_image");
        return SG_ERR_UNSUPPORTED;
    }
    if (!(global_max > 0.0) || !std::isfinite(global_max)) { set_error("sg_stft_db: global_max must be positive and finite (the batch-global base of PlotEngine.py:126)"); return SG_ERR_ARG; }
    const int nbins = plan->nfft / 2 + 1;
    if (k_lo < 0 || k_hi >= nbins || k_lo > k_hi) { set_error("bad band [%d,%d] of %d bins", k_lo, k_hi, nbins); return SG_ERR_ARG; }
    if (!mm_dev) { set_error("null device pointer"); return SG_ERR_ARG; }
    if (n_clips < 0 || n_samples < 0) { set_error("negative sizes"); return SG_ERR_ARG; }
    auto s = static_cast<hipStream_t>(stream);
    const int64_t n_frames = n_samples < plan->nperseg ? 0 : (n_samples - plan->nperseg) / plan->hop + 1;
    if (n_frames == 0 || n_clips == 0) {       // empty image: (min, max) = (+inf, -inf), like an empty reduction
        const float e[2] = {INFINITY, -INFINITY};
        SG_HIP(hipMemcpyAsync(mm_dev, e, sizeof(e), hipMemcpyHostToDevice, s));
        SG_HIP(hipStreamSynchronize(s));
        return SG_OK;
    }
    if (!x_dev || !db_dev) { set_error("null device pointer"); return SG_ERR_ARG; }
    if (n_clips > 1 && clip_stride < n_samples) { set_error("clip_stride < n_samples"); return SG_ERR_ARG; }
    const int64_t need = n_frames * (k_hi - k_lo + 1);
    if (n_clips > 1 && out_clip_stride < need) { set_error("out_clip_stride %lld < %lld", (long long)out_clip_stride, (long long)need); return SG_ERR_ARG; }
    std::lock_guard<std::recursive_mutex> seq(launch_sequence_mutex(s));
    void* parts = reduction_scratch(s);
    if (!parts) { set_error("sg_stft_db: no scratch memory"); return SG_ERR_HIP; }
    StftArgs a{};
    a.x = x_dev; a.in_i16 = 0; a.n_samples = n_samples; a.clip_stride = clip_stride; a.n_clips = n_clips;
    a.out = db_dev; a.out_clip_stride = out_clip_stride; a.n_frames = n_frames; a.k_lo = k_lo; a.k_hi = k_hi;
    a.db_mode = 1; a.inv_base = 1.0f / (static_cast<float>(global_max) + 1e-20f); a.mm_parts = parts; a.stream = s;
    if (n_frames > INT32_MAX) { set_error("more than 2^31 frames per clip"); return SG_ERR_ARG; }
    const int n_parts = r8x3_grid_waves(*plan, n_frames * n_clips);      // <= 4 waves x 4 SIMDs x CUs = 4096 pairs of 8 B
    if (static_cast<size_t>(n_parts) * 8 > sizeof(double) * 16 * 2048) { set_error("sg_stft_db: partials exceed the scratch"); return SG_ERR_HIP; }
    if (int rc = launch_r8x3(*plan, a)) return rc;
    return fold_minmax_f32(static_cast<const float*>(parts), n_parts, mm_dev, s);
}

int sg_stft_mel_sparse(const sg_plan* plan, const float* x_dev, int64_t n_samples, int64_t clip_stride, int n_clips,
                       const int32_t* item_start_dev, const float* item_w_dev, const int32_t* band_first_dev,
                       const int32_t* band_count_dev, int items_per_lane, int n_mels, int log_scale, float* mel_dev,
                       int64_t out_clip_stride, void* stream) {
    if (!plan) { set_error("null plan"); return SG_ERR_ARG; }
    if (plan->kernel != Kernel::R8X3 || plan->mode != SG_MODE_PSD) {
        set_error("fused STFT+mel needs an f32 nperseg = nfft = 1024 PSD plan (kernel r8x3)");
        return SG_ERR_UNSUPPORTED;
    }
    if (items_per_lane < 1 || items_per_lane > 4 || n_mels < 1 || n_mels > 128 || n_clips < 0 || n_samples < 0) { set_error("bad sizes"); return SG_ERR_ARG; }
    const int64_t n_frames = n_samples < plan->nperseg ? 0 : (n_samples - plan->nperseg) / plan->hop + 1;
    if (n_frames == 0 || n_clips == 0) return SG_OK;
    if (!x_dev || !mel_dev || !item_start_dev || !item_w_dev || !band_first_dev || !band_count_dev) { set_error("null device pointer"); return SG_ERR_ARG; }
    if (n_clips > 1 && (clip_stride < n_samples || out_clip_stride < n_frames * n_mels)) { set_error("bad strides"); return SG_ERR_ARG; }
    StftArgs a{};
    a.x = x_dev; a.in_i16 = 0; a.n_samples = n_samples; a.clip_stride = clip_stride; a.n_clips = n_clips;
    a.out = mel_dev; a.out_clip_stride = out_clip_stride; a.n_frames = n_frames; a.stream = static_cast<hipStream_t>(stream);
    a.mel_ipl = items_per_lane; a.mel_start = item_start_dev; a.mel_w = item_w_dev; a.mel_first = band_first_dev;
    a.mel_count = band_count_dev; a.n_mels = n_mels; a.log_scale = log_scale;
    return launch_r8x3(*plan, a);
}

int sg_time_stft(const sg_plan* plan, const void* x_dev, int64_t n_samples, int64_t clip_stride, int n_clips,
                 void* out_dev, int64_t out_clip_stride, void* stream, int iters, float* ms_per_launch) {
    if (!ms_per_launch || iters < 1) { set_error("bad argument"); return SG_ERR_ARG; }
    auto s = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    SG_HIP(hipEventCreate(&e0));
    SG_HIP(hipEventCreate(&e1));
    int rc = SG_OK;
    SG_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters && rc == SG_OK; ++i)
        rc = sg_stft(plan, x_dev, n_samples, clip_stride, n_clips, out_dev, out_clip_stride, stream);
    SG_HIP(hipEventRecord(e1, s));
    SG_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SG_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = ms / static_cast<float>(iters);
    return rc;
}

}  // extern "C"
