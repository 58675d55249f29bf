// epilogue.hip -- the reductions and elementwise maps the reference applies to the spectrogram
// after scipy returns it: PlotEngine.py:114-131 (mask, normalise, dB, min-max), :238-241 (band
// log-power features), :686-719 (absolute / relative band powers).  All HBM-bound; one pass each.
#include "spectro_internal.h"

namespace sg {
namespace {

constexpr int kThreads = 256;

// order-preserving float <-> unsigned key so that atomicMin/Max on integers order floats
template <typename T> struct Key;
template <> struct Key<float> {
    using U = unsigned int;
    static __device__ __forceinline__ U enc(float f) {
        const U b = __builtin_bit_cast(U, f);
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    static __device__ __forceinline__ float dec(U k) {
        const U b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        return __builtin_bit_cast(float, b);
    }
    static constexpr U kMaxKey = 0xffffffffu;
};
template <> struct Key<double> {
    using U = unsigned long long;
    static __device__ __forceinline__ U enc(double f) {
        const U b = __builtin_bit_cast(U, f);
        return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
    }
    static __device__ __forceinline__ double dec(U k) {
        const U b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
        return __builtin_bit_cast(double, b);
    }
    static constexpr U kMaxKey = 0xffffffffffffffffull;
};

template <typename T>
__device__ __forceinline__ T wave_min(T v) {
    for (int o = 32; o > 0; o >>= 1) { const T w = __shfl_xor(v, o); v = w < v ? w : v; }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
    for (int o = 32; o > 0; o >>= 1) { const T w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_add(T v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename T>
__global__ void minmax_init_kernel(typename Key<T>::U* mm) {
    mm[0] = Key<T>::kMaxKey;
    mm[1] = 0;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void minmax_kernel(const T* spec, int64_t n_frames, int n_bins, int k_lo, int width,
                                                          typename Key<T>::U* mm) {
    const int64_t total = n_frames * width;
    T lo = INFINITY, hi = -INFINITY;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * kThreads) {
        const int64_t f = i / width;
        const int k = static_cast<int>(i - f * width) + k_lo;
        const T v = spec[f * n_bins + k];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], Key<T>::enc(lo));
        atomicMax(&mm[1], Key<T>::enc(hi));
    }
}

template <typename T>
__global__ void minmax_decode_kernel(typename Key<T>::U* mm) {
    const T lo = Key<T>::dec(mm[0]), hi = Key<T>::dec(mm[1]);
    reinterpret_cast<T*>(mm)[0] = lo;
    reinterpret_cast<T*>(mm)[1] = hi;
}

template <typename T> __device__ __forceinline__ T t_log10(T v);
template <> __device__ __forceinline__ float t_log10<float>(float v) { return log10f(v); }
template <> __device__ __forceinline__ double t_log10<double>(double v) { return log10(v); }

template <typename T>
__device__ __forceinline__ T norm_lin(T s, T base) {
    T v = s / (base + T(1e-20));
    v = v < T(0) ? T(0) : v;
    return v > T(1) ? T(1) : v;
}
template <typename T>
__device__ __forceinline__ T norm_db(T s, T base) {
    const T d = T(10) * t_log10<T>(norm_lin(s, base) + T(1e-12));
    return d != d ? T(0) : d;      // np.nan_to_num
}

// A9/A10 in one pass given the band's min/max (PlotEngine.py:126-131)
template <typename T>
__global__ __launch_bounds__(kThreads) void normalise_kernel(const T* spec, int64_t n_frames, int n_bins, int k_lo, int width,
                                                             int log_scale, T global_max, const T* mm, T* img) {
    const T smin = mm[0], smax = mm[1];
    const T base = global_max > T(0) ? global_max : smax;
    T db_lo = T(0), range = T(1);
    bool degenerate = false;
    if (log_scale) {
        db_lo = norm_db(smin, base);
        range = norm_db(smax, base) - db_lo;
        degenerate = !(range > T(1e-6));
    }
    const int64_t total = n_frames * width;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * kThreads) {
        const int64_t f = i / width;
        const int k = static_cast<int>(i - f * width) + k_lo;
        const T s = spec[f * n_bins + k];
        T v;
        if (!log_scale) v = norm_lin(s, base);
        else v = degenerate ? T(0) : (norm_db(s, base) - db_lo) / range;
        img[i] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void slice_kernel(const T* spec, int64_t n_frames, int n_bins, int k_lo, int width, T* dst) {
    const int64_t total = n_frames * width;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * kThreads) {
        const int64_t f = i / width;
        dst[i] = spec[f * n_bins + (i - f * width) + k_lo];
    }
}

// one wavefront per frame
template <typename T>
__global__ __launch_bounds__(kThreads) void band_sum_kernel(const T* spec, int64_t n_frames, int n_bins, int k_lo, int k_hi, T* band) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kThreads / 64);
    for (int64_t f = wave; f < n_frames; f += n_waves) {
        T s = T(0);
        for (int k = k_lo + lane; k <= k_hi; k += 64) s += spec[f * n_bins + k];
        s = wave_add(s);
        if (lane == 0) band[f] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void band_features_kernel(const T* band, int64_t n_frames, T* feat) {
    for (int64_t f = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; f < n_frames;
         f += static_cast<int64_t>(gridDim.x) * kThreads) {
        const T lp = t_log10<T>(band[f] + T(1e-20));
        const T lp_prev = f > 0 ? t_log10<T>(band[f - 1] + T(1e-20)) : lp;
        feat[2 * f] = lp;
        feat[2 * f + 1] = lp - lp_prev;
    }
}

struct Bands { int n; int lo[16]; int hi[16]; };

template <typename T>
__global__ __launch_bounds__(kThreads) void band_totals_kernel(const T* spec, int64_t n_frames, int n_bins, Bands b, double* sums) {
    __shared__ double part[kThreads / 64];
    for (int ib = 0; ib < b.n; ++ib) {
        const int lo = b.lo[ib], width = b.hi[ib] - b.lo[ib];
        double acc = 0.0;
        if (width > 0) {
            const int64_t total = n_frames * width;
            for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
                 i += static_cast<int64_t>(gridDim.x) * kThreads) {
                const int64_t f = i / width;
                const T v = spec[f * n_bins + (i - f * width) + lo];
                acc += v > T(0) ? static_cast<double>(v) : 0.0;
            }
        }
        acc = wave_add(acc);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < kThreads / 64; ++w) t += part[w];
            if (t != 0.0) atomicAdd(&sums[ib], t);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kThreads) void colormap_kernel(const float* img, int64_t n, const uchar4* lut, uchar4* rgba) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads) {
        const float v = img[i];
        uchar4 c = make_uchar4(0, 0, 0, 0);
        if (v == v) {
            int idx = static_cast<int>(v * 256.0f);              // matplotlib: int(x * N), x == 1 -> N - 1
            idx = idx < 0 ? 0 : (idx > 255 ? 255 : idx);
            c = lut[idx];
        }
        rgba[i] = c;
    }
}

inline unsigned grid_for(int64_t items, int cap = 256 * 8) {
    int64_t g = (items + kThreads - 1) / kThreads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return static_cast<unsigned>(g);
}

inline int check_band(int64_t n_frames, int n_bins, int k_lo, int k_hi) {
    if (n_frames < 0 || n_bins <= 0 || k_lo < 0 || k_hi >= n_bins || k_lo > k_hi) {
        set_error("bad band: n_frames=%lld n_bins=%d k_lo=%d k_hi=%d", static_cast<long long>(n_frames), n_bins, k_lo, k_hi);
        return SG_ERR_ARG;
    }
    return SG_OK;
}

inline int after_launch(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, what);
}

template <typename T>
int minmax_t(const void* spec, int64_t n_frames, int n_bins, int k_lo, int k_hi, void* mm, hipStream_t s) {
    using U = typename Key<T>::U;
    const int width = k_hi - k_lo + 1;
    hipLaunchKernelGGL(minmax_init_kernel<T>, dim3(1), dim3(1), 0, s, static_cast<U*>(mm));
    if (n_frames > 0)
        hipLaunchKernelGGL(minmax_kernel<T>, dim3(grid_for(n_frames * width)), dim3(kThreads), 0, s,
                           static_cast<const T*>(spec), n_frames, n_bins, k_lo, width, static_cast<U*>(mm));
    hipLaunchKernelGGL(minmax_decode_kernel<T>, dim3(1), dim3(1), 0, s, static_cast<U*>(mm));
    return after_launch("minmax");
}

template <typename T>
int normalise_t(const void* spec, int64_t n_frames, int n_bins, int k_lo, int k_hi, int log_scale, double gmax,
                void* img, void* mm, hipStream_t s) {
    int rc = minmax_t<T>(spec, n_frames, n_bins, k_lo, k_hi, mm, s);
    if (rc != SG_OK || n_frames == 0) return rc;
    const int width = k_hi - k_lo + 1;
    hipLaunchKernelGGL(normalise_kernel<T>, dim3(grid_for(n_frames * width)), dim3(kThreads), 0, s,
                       static_cast<const T*>(spec), n_frames, n_bins, k_lo, width, log_scale, static_cast<T>(gmax),
                       static_cast<const T*>(mm), static_cast<T*>(img));
    return after_launch("normalise");
}

}  // namespace
}  // namespace sg

using namespace sg;

#define SG_DISPATCH(dtype, expr_f32, expr_f64)                                 \
    do {                                                                       \
        if ((dtype) == SG_F32) return (expr_f32);                              \
        if ((dtype) == SG_F64) return (expr_f64);                              \
        set_error("bad dtype %d", (dtype));                                    \
        return SG_ERR_ARG;                                                     \
    } while (0)

extern "C" {

int sg_jet_lut(uint8_t* rgba_host) {
    if (!rgba_host) { set_error("null pointer"); return SG_ERR_ARG; }
    // matplotlib _cm.py 'jet' segment data: (x, y) breakpoints, linear in between
    static const double R[][2] = {{0, 0}, {0.35, 0}, {0.66, 1}, {0.89, 1}, {1, 0.5}};
    static const double G[][2] = {{0, 0}, {0.125, 0}, {0.375, 1}, {0.64, 1}, {0.91, 0}, {1, 0}};
    static const double B[][2] = {{0, 0.5}, {0.11, 1}, {0.34, 1}, {0.65, 0}, {1, 0}};
    // same arithmetic as matplotlib.colors._create_lookup_table (N = 256, gamma = 1): breakpoints scaled by N-1,
    // sample i at (N-1) * (i * (1/(N-1))), left searchsorted, linear blend; end points taken verbatim
    auto interp = [](const double (*seg)[2], int n, int i) {
        if (i == 0) return seg[0][1];
        if (i == 255) return seg[n - 1][1];
        const double xind = 255.0 * (static_cast<double>(i) * (1.0 / 255.0));
        int ind = 0;
        while (ind < n && seg[ind][0] * 255.0 < xind) ++ind;
        const double x0 = seg[ind - 1][0] * 255.0, x1 = seg[ind][0] * 255.0;
        const double distance = (xind - x0) / (x1 - x0);
        return distance * (seg[ind][1] - seg[ind - 1][1]) + seg[ind - 1][1];
    };
    for (int i = 0; i < 256; ++i) {
        const double c[3] = {interp(R, 5, i), interp(G, 6, i), interp(B, 5, i)};
        for (int k = 0; k < 3; ++k) {
            double v = c[k] < 0 ? 0 : (c[k] > 1 ? 1 : c[k]);
            rgba_host[4 * i + k] = static_cast<uint8_t>(v * 255.0);      // matplotlib bytes=True: (lut * 255).astype(uint8)
        }
        rgba_host[4 * i + 3] = 255;
    }
    return SG_OK;
}

int sg_colormap(const float* img_dev, int64_t n, const uint8_t* lut_dev, uint8_t* rgba_dev, void* stream) {
    if (!img_dev || !lut_dev || !rgba_dev || n < 0) { set_error("bad argument"); return SG_ERR_ARG; }
    if (n == 0) return SG_OK;
    hipLaunchKernelGGL(colormap_kernel, dim3(grid_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), img_dev, n,
                       reinterpret_cast<const uchar4*>(lut_dev), reinterpret_cast<uchar4*>(rgba_dev));
    return after_launch("colormap");
}

int sg_minmax(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int k_lo, int k_hi, void* mm_dev, void* stream) {
    if (!spec_dev || !mm_dev) { set_error("null pointer"); return SG_ERR_ARG; }
    if (int rc = check_band(n_frames, n_bins, k_lo, k_hi)) return rc;
    auto s = static_cast<hipStream_t>(stream);
    SG_DISPATCH(dtype, minmax_t<float>(spec_dev, n_frames, n_bins, k_lo, k_hi, mm_dev, s),
                minmax_t<double>(spec_dev, n_frames, n_bins, k_lo, k_hi, mm_dev, s));
}

int sg_normalise_image(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int k_lo, int k_hi, int log_scale,
                       double global_max, void* img_dev, void* mm_dev, void* stream) {
    if (!spec_dev || !mm_dev || !img_dev) { set_error("null pointer"); return SG_ERR_ARG; }
    if (int rc = check_band(n_frames, n_bins, k_lo, k_hi)) return rc;
    auto s = static_cast<hipStream_t>(stream);
    SG_DISPATCH(dtype, normalise_t<float>(spec_dev, n_frames, n_bins, k_lo, k_hi, log_scale, global_max, img_dev, mm_dev, s),
                normalise_t<double>(spec_dev, n_frames, n_bins, k_lo, k_hi, log_scale, global_max, img_dev, mm_dev, s));
}

int sg_slice_bins(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int k_lo, int k_hi, void* dst_dev, void* stream) {
    if (!spec_dev || !dst_dev) { set_error("null pointer"); return SG_ERR_ARG; }
    if (int rc = check_band(n_frames, n_bins, k_lo, k_hi)) return rc;
    if (n_frames == 0) return SG_OK;
    auto s = static_cast<hipStream_t>(stream);
    const int width = k_hi - k_lo + 1;
    if (dtype == SG_F32)
        hipLaunchKernelGGL(slice_kernel<float>, dim3(grid_for(n_frames * width)), dim3(kThreads), 0, s,
                           static_cast<const float*>(spec_dev), n_frames, n_bins, k_lo, width, static_cast<float*>(dst_dev));
    else if (dtype == SG_F64)
        hipLaunchKernelGGL(slice_kernel<double>, dim3(grid_for(n_frames * width)), dim3(kThreads), 0, s,
                           static_cast<const double*>(spec_dev), n_frames, n_bins, k_lo, width, static_cast<double*>(dst_dev));
    else { set_error("bad dtype %d", dtype); return SG_ERR_ARG; }
    return after_launch("slice_bins");
}

int sg_band_sum(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int k_lo, int k_hi, void* band_dev, void* stream) {
    if (!spec_dev || !band_dev) { set_error("null pointer"); return SG_ERR_ARG; }
    if (int rc = check_band(n_frames, n_bins, k_lo, k_hi)) return rc;
    if (n_frames == 0) return SG_OK;
    auto s = static_cast<hipStream_t>(stream);
    const unsigned g = grid_for(n_frames * 64);
    if (dtype == SG_F32)
        hipLaunchKernelGGL(band_sum_kernel<float>, dim3(g), dim3(kThreads), 0, s, static_cast<const float*>(spec_dev), n_frames,
                           n_bins, k_lo, k_hi, static_cast<float*>(band_dev));
    else if (dtype == SG_F64)
        hipLaunchKernelGGL(band_sum_kernel<double>, dim3(g), dim3(kThreads), 0, s, static_cast<const double*>(spec_dev), n_frames,
                           n_bins, k_lo, k_hi, static_cast<double*>(band_dev));
    else { set_error("bad dtype %d", dtype); return SG_ERR_ARG; }
    return after_launch("band_sum");
}

int sg_band_features(const void* band_dev, int dtype, int64_t n_frames, void* feat_dev, void* stream) {
    if (!band_dev || !feat_dev) { set_error("null pointer"); return SG_ERR_ARG; }
    if (n_frames < 0) { set_error("n_frames < 0"); return SG_ERR_ARG; }
    if (n_frames == 0) return SG_OK;
    auto s = static_cast<hipStream_t>(stream);
    if (dtype == SG_F32)
        hipLaunchKernelGGL(band_features_kernel<float>, dim3(grid_for(n_frames)), dim3(kThreads), 0, s,
                           static_cast<const float*>(band_dev), n_frames, static_cast<float*>(feat_dev));
    else if (dtype == SG_F64)
        hipLaunchKernelGGL(band_features_kernel<double>, dim3(grid_for(n_frames)), dim3(kThreads), 0, s,
                           static_cast<const double*>(band_dev), n_frames, static_cast<double*>(feat_dev));
    else { set_error("bad dtype %d", dtype); return SG_ERR_ARG; }
    return after_launch("band_features");
}

int sg_band_totals(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int n_bands, const int* k_lo_host,
                   const int* k_hi_host, double* sums_dev, void* stream) {
    if (!spec_dev || !sums_dev || !k_lo_host || !k_hi_host) { set_error("null pointer"); return SG_ERR_ARG; }
    if (n_bands < 1 || n_bands > 16 || n_frames < 0 || n_bins <= 0) { set_error("bad band table (1..16 bands)"); return SG_ERR_ARG; }
    Bands b{};
    b.n = n_bands;
    for (int i = 0; i < n_bands; ++i) {
        int lo = k_lo_host[i], hi = k_hi_host[i];
        if (lo < 0) lo = 0;
        if (hi > n_bins) hi = n_bins;
        if (hi < lo) hi = lo;
        b.lo[i] = lo;
        b.hi[i] = hi;
    }
    auto s = static_cast<hipStream_t>(stream);
    SG_HIP(hipMemsetAsync(sums_dev, 0, sizeof(double) * n_bands, s));
    if (n_frames == 0) return SG_OK;
    const unsigned g = grid_for(n_frames * n_bins, 256 * 4);
    if (dtype == SG_F32)
        hipLaunchKernelGGL(band_totals_kernel<float>, dim3(g), dim3(kThreads), 0, s, static_cast<const float*>(spec_dev), n_frames,
                           n_bins, b, sums_dev);
    else if (dtype == SG_F64)
        hipLaunchKernelGGL(band_totals_kernel<double>, dim3(g), dim3(kThreads), 0, s, static_cast<const double*>(spec_dev), n_frames,
                           n_bins, b, sums_dev);
    else { set_error("bad dtype %d", dtype); return SG_ERR_ARG; }
    return after_launch("band_totals");
}

}  // extern "C"
