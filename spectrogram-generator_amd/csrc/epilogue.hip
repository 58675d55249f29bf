// epilogue.hip -- the reductions and elementwise maps the reference applies to the spectrogram
// after scipy returns it: PlotEngine.py:114-131 (mask, normalise, dB, min-max), :238-241 (band
// log-power features), :686-719 (absolute / relative band powers).  All HBM-bound; one pass each.
#include "spectro_internal.h"
#include <memory>

#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace sg {
namespace {

constexpr int kThreads = 256;

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

// Reductions run in two stages: every workgroup leaves one partial in a per-stream scratch buffer, a single
// workgroup folds the partials.  (16 K same-address atomics cost 180 us on this part -- four times the HBM
// pass they would guard -- and the two-stage result does not depend on arrival order.)
constexpr int kMaxParts = 2048;

// All band kernels walk the spectrum one wavefront per row, lanes striding over the bins of the band: no per-element
// 64-bit division, 256 contiguous bytes per wave-instruction on a wide band.  A band that spans every bin is one
// contiguous array; it is then re-cut into rows of kFlatRow elements (last one ragged) and read 16 B per lane.
struct Rows {
    int64_t n_rows;      // rows to walk
    int64_t pitch;       // elements between row starts in the spectrum
    int width;           // elements per row (and the row pitch of a packed output)
    int last_width;      // elements in the last row
};
constexpr int kFlatRow = 2048;

template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };

template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void minmax_kernel(const T* spec, Rows r, T* parts) {
    using V = typename Vec16<T>::type;
    constexpr int N = VEC ? Vec16<T>::N : 1;
    __shared__ T part[2][kThreads / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + wv;
    const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kThreads / 64);
    T lo = INFINITY, hi = -INFINITY;
    for (int64_t f = wave; f < r.n_rows; f += n_waves) {
        const T* const row = spec + f * r.pitch;
        const int w = f == r.n_rows - 1 ? r.last_width : r.width;
        for (int k = lane * N; k < w; k += 64 * N) {
            if (VEC && k + N <= w) {
                const V v = *reinterpret_cast<const V*>(row + k);
                const T* const e = reinterpret_cast<const T*>(&v);
#pragma unroll
                for (int i = 0; i < N; ++i) { lo = e[i] < lo ? e[i] : lo; hi = e[i] > hi ? e[i] : hi; }
            } else {
                for (int i = 0; i < N && k + i < w; ++i) {
                    const T v = row[k + i];
                    lo = v < lo ? v : lo;
                    hi = v > hi ? v : hi;
                }
            }
        }
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane == 0) { part[0][wv] = lo; part[1][wv] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w) {
            lo = part[0][w] < lo ? part[0][w] : lo;
            hi = part[1][w] > hi ? part[1][w] : hi;
        }
        parts[2 * blockIdx.x] = lo;
        parts[2 * blockIdx.x + 1] = hi;
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void minmax_fold_kernel(const T* parts, int n_parts, T* mm) {
    __shared__ T part[2][kThreads / 64];
    T lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < n_parts; i += kThreads) {
        const T a = parts[2 * i], b = parts[2 * i + 1];
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = lo; part[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w) {
            lo = part[0][w] < lo ? part[0][w] : lo;
            hi = part[1][w] > hi ? part[1][w] : hi;
        }
        mm[0] = lo;
        mm[1] = hi;
    }
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
template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void normalise_kernel(const T* spec, Rows r, int log_scale, T global_max, const T* mm, T* img) {
    using V = typename Vec16<T>::type;
    constexpr int N = VEC ? Vec16<T>::N : 1;
    const T smin = mm[0], smax = mm[1];
    const T base = global_max > T(0) ? global_max : smax;
    T db_lo = T(0), range = T(1);
    bool degenerate = false;
    if (log_scale) {
        db_lo = norm_db(smin, base);
        range = norm_db(smax, base) - db_lo;
        degenerate = !(range > T(1e-6));
    }
    auto map = [&](T s) -> T {
        if (!log_scale) return norm_lin(s, base);
        return degenerate ? T(0) : (norm_db(s, base) - db_lo) / range;
    };
    const int lane = threadIdx.x & 63;
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kThreads / 64);
    for (int64_t f = wave; f < r.n_rows; f += n_waves) {
        const T* const row = spec + f * r.pitch;
        T* const orow = img + f * r.width;
        const int w = f == r.n_rows - 1 ? r.last_width : r.width;
        for (int k = lane * N; k < w; k += 64 * N) {
            if (VEC && k + N <= w) {
                V v = *reinterpret_cast<const V*>(row + k);
                T* const e = reinterpret_cast<T*>(&v);
#pragma unroll
                for (int i = 0; i < N; ++i) e[i] = map(e[i]);
                *reinterpret_cast<V*>(orow + k) = v;
            } else {
                for (int i = 0; i < N && k + i < w; ++i) orow[k + i] = map(row[k + i]);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void slice_kernel(const T* spec, int64_t n_frames, int n_bins, int k_lo, int width, T* dst) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kThreads / 64);
    for (int64_t f = wave; f < n_frames; f += n_waves)
        for (int k = lane; k < width; k += 64) dst[f * width + k] = spec[f * n_bins + k_lo + k];
}

// one wavefront per 4 frames (four rows of loads in flight per wave step)
template <typename T>
__global__ __launch_bounds__(kThreads) void band_sum_kernel(const T* spec, int64_t n_frames, int n_bins, int k_lo, int k_hi, T* band) {
    constexpr int R = 4;
    const int lane = threadIdx.x & 63;
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kThreads / 64);
    for (int64_t f0 = wave * R; f0 < n_frames; f0 += n_waves * R) {
        const int nr = n_frames - f0 < R ? static_cast<int>(n_frames - f0) : R;
        const T* const rows = spec + f0 * n_bins;
        T s[R];
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] = T(0);
        for (int k = k_lo + lane; k <= k_hi; k += 64) {
#pragma unroll
            for (int r = 0; r < R; ++r) s[r] += r < nr ? rows[static_cast<int64_t>(r) * n_bins + k] : T(0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const T t = wave_add(s[r]);
            if (lane == 0 && r < nr) band[f0 + r] = t;
        }
    }
}

// `per_clip` frames form one clip: the first difference does not reach across a clip boundary (diff(prepend=lp[0]))
template <typename T>
__global__ __launch_bounds__(kThreads) void band_features_kernel(const T* band, int64_t n_frames, int64_t per_clip, T* feat) {
    for (int64_t f = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; f < n_frames;
         f += static_cast<int64_t>(gridDim.x) * kThreads) {
        const T lp = t_log10<T>(band[f] + T(1e-20));
        const T lp_prev = (f % per_clip) > 0 ? t_log10<T>(band[f - 1] + T(1e-20)) : lp;
        feat[2 * f] = lp;
        feat[2 * f + 1] = lp - lp_prev;
    }
}

constexpr int kRowsPerStep = 16;        // rows a wave has in flight per 64-bin step (8: 86-90 us, 16: 73 us, 32: 78-80 us per cfg2 spectrum, balanced runs)
struct Bands { int n; int k_begin; int k_end; int lo[16]; int hi[16]; };     // [lo, hi) each; [k_begin, k_end) their hull

// A13: every band total from one read of the rows.  A wave takes kRowsPerStep rows at a time, 64 bins per step:
// the rows' (clamped) values are added first, and only the bands that touch the 64-bin chunk (a wave-uniform test,
// shared by the rows -- the scalar unit is per CU and was the bound when every row paid for it) cost vector work.
template <typename T>
__global__ __launch_bounds__(kThreads) void band_totals_kernel(const T* spec, int64_t n_frames, int n_bins, Bands b, double* parts) {
    constexpr int R = kRowsPerStep;
    __shared__ double part[16][kThreads / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + wv;
    const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kThreads / 64);
    double acc[16];
#pragma unroll
    for (int ib = 0; ib < 16; ++ib) acc[ib] = 0.0;
    const int c_begin = b.k_begin & ~63;
    // every wave owns one contiguous, balanced run of rows (round 3): with R-row groups dealt round-robin over a grid larger than the chip
    // holds, cfg2's 14 976 groups were 1.8 per wave -- some waves did two, some one, and the workgroups of the second round waited for
    // the first: 105 us; one resident round of equal runs, 16 rows per step: 73 us (profiles/r03_band_totals_runs.txt)
    const int64_t r_end = n_frames * (wave + 1) / n_waves;
    for (int64_t f0 = n_frames * wave / n_waves; f0 < r_end; f0 += R) {
        const T* const rows = spec + f0 * n_bins;
        const int nr = r_end - f0 < R ? static_cast<int>(r_end - f0) : R;
        T seg[16];                                 // a lane's share of R rows (few terms); row groups add up in double
#pragma unroll
        for (int ib = 0; ib < 16; ++ib) seg[ib] = T(0);
        for (int c = c_begin; c < b.k_end; c += 64) {
            const int k = c + lane;
            const bool in = k >= b.k_begin && k < b.k_end;
            T v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = (in && r < nr) ? rows[static_cast<int64_t>(r) * n_bins + k] : T(0);
            T x = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) x += v[r] > T(0) ? v[r] : T(0);
#pragma unroll
            for (int ib = 0; ib < 16; ++ib) {
                if (ib >= b.n) break;
                if (b.hi[ib] > c && b.lo[ib] < c + 64) seg[ib] += (k >= b.lo[ib] && k < b.hi[ib]) ? x : T(0);
            }
        }
#pragma unroll
        for (int ib = 0; ib < 16; ++ib)
            if (ib < b.n) acc[ib] += static_cast<double>(seg[ib]);
    }
#pragma unroll
    for (int ib = 0; ib < 16; ++ib) {
        if (ib < b.n) {
            const double t = wave_add(acc[ib]);
            if (lane == 0) part[ib][wv] = t;
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        double t = 0.0;
        if (threadIdx.x < b.n)
            for (int w = 0; w < kThreads / 64; ++w) t += part[threadIdx.x][w];
        parts[16 * blockIdx.x + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(kThreads) void band_totals_fold_kernel(const double* parts, int n_parts, int n_bands, double* sums) {
    __shared__ double part[16][16];
    const int ib = threadIdx.x & 15, sub = threadIdx.x >> 4;           // 16 bands x 16 strided sub-sums
    double t = 0.0;
    for (int i = sub; i < n_parts; i += 16) t += parts[16 * i + ib];
    part[sub][ib] = t;
    __syncthreads();
    if (threadIdx.x < n_bands) {
        double tot = 0.0;
        for (int j = 0; j < 16; ++j) tot += part[j][threadIdx.x];
        sums[threadIdx.x] = tot;
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

}  // namespace

// Everything the library keeps per (device, stream): launches on a stream are ordered, so successive calls on it may share
//   * a small scratch for reduction partials (lives until the library is unloaded),
//   * a workspace that grows on demand: the float copy of an int16 batch (sg_stft_i16 on rsmall / rbig plans), the chirp-z
//     convolution buffers beyond LDS size.  Growing it synchronises THAT stream and frees the old block (include/spectro.h says
//     so).  hipMallocAsync / hipFreeAsync around each call is NOT used ("A runtime finding").  Held until
//     sg_workspace_release(),
//   * one lock: a call that hands data from one launch to the next through any of the above SUBMITS its launches under it, so
//     two host threads sharing a stream cannot interleave their sequences.  Calls on different streams or devices never
//     wait for each other.
namespace {
struct StreamState {
    std::recursive_mutex mu;
    int device = 0;
    void* scratch = nullptr;
    void* ws = nullptr; size_t ws_bytes = 0;
};
// Lock order: g_streams_mu is only ever taken for the map itself (look-up, insert, erase) and released before a state's `mu` is taken;
// nothing takes g_streams_mu while holding a state's `mu`.  States are never destroyed: a dropped stream's state moves to g_retired,
// so a reference another thread still holds (launch_sequence_mutex) stays valid whatever the caller does.
std::mutex g_streams_mu;
std::map<std::pair<int, hipStream_t>, std::shared_ptr<StreamState>> g_streams;
std::vector<std::shared_ptr<StreamState>> g_retired;

// the device a stream belongs to (NOT the calling thread's current device: sg_stream_destroy / sg_workspace_release may run after
// the caller has moved on to another GPU); the null stream belongs to the current device
int stream_device(hipStream_t s) {
    int dev = 0;
    if (s != nullptr && hipStreamGetDevice(s, &dev) == hipSuccess) return dev;
    (void)hipGetLastError();
    return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}

StreamState* stream_state(hipStream_t s) {
    const int dev = stream_device(s);
    if (dev < 0) return nullptr;
    std::lock_guard<std::mutex> lock(g_streams_mu);
    auto& slot = g_streams[{dev, s}];
    if (!slot) { slot = std::make_shared<StreamState>(); slot->device = dev; }
    return slot.get();
}

// frees what a state holds, on the state's own device; the caller holds st.mu
void free_state_memory(StreamState& st, bool sync_device) {
    if (!st.ws && !st.scratch) return;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (have_cur && cur != st.device) (void)hipSetDevice(st.device);
    if (sync_device) (void)hipDeviceSynchronize();
    if (st.ws) (void)hipFree(st.ws);
    if (st.scratch) (void)hipFree(st.scratch);
    if (have_cur && cur != st.device) (void)hipSetDevice(cur);
    st.ws = nullptr;
    st.ws_bytes = 0;
    st.scratch = nullptr;
}
}  // namespace

int device_cu_count() {
    static std::mutex mu;
    static int cached[64];                 // 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lock(mu);
    if (cached[dev] == 0) {
        hipDeviceProp_t prop;
        cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cached[dev];
}

std::recursive_mutex& launch_sequence_mutex(hipStream_t s) {
    static std::recursive_mutex no_device;
    StreamState* st = stream_state(s);
    return st ? st->mu : no_device;
}

void* reduction_scratch(hipStream_t s) {
    StreamState* st = stream_state(s);
    if (!st) return nullptr;
    std::lock_guard<std::recursive_mutex> lock(st->mu);
    if (!st->scratch && hipMalloc(&st->scratch, sizeof(double) * 16 * kMaxParts) != hipSuccess) { (void)hipGetLastError(); st->scratch = nullptr; }
    return st->scratch;
}

void* stream_workspace(hipStream_t s, size_t bytes) {
    StreamState* st = stream_state(s);
    if (!st) return nullptr;
    std::lock_guard<std::recursive_mutex> lock(st->mu);
    if (st->ws_bytes < bytes) {
        if (st->ws) { (void)hipStreamSynchronize(s); (void)hipFree(st->ws); st->ws = nullptr; st->ws_bytes = 0; }
        const size_t want = bytes + bytes / 4;
        if (hipMalloc(&st->ws, want) != hipSuccess) {
            (void)hipGetLastError();
            if (hipMalloc(&st->ws, bytes) != hipSuccess) { (void)hipGetLastError(); st->ws = nullptr; return nullptr; }
            st->ws_bytes = bytes;
        } else {
            st->ws_bytes = want;
        }
    }
    return st->ws;
}

// sg_stream_destroy: the stream's reduction scratch and workspace go with it and its state leaves the map (the runtime may hand the same
// handle value to a later stream; that one starts from an empty state).  The caller has synchronised the stream.
void drop_stream_state(hipStream_t s) {
    const int dev = stream_device(s);
    std::shared_ptr<StreamState> st;
    {
        std::lock_guard<std::mutex> lock(g_streams_mu);
        auto it = dev >= 0 ? g_streams.find({dev, s}) : g_streams.end();
        if (it == g_streams.end()) {                         // (a stream whose device can no longer be asked: look for the handle on every device)
            for (it = g_streams.begin(); it != g_streams.end() && it->first.second != s; ++it) {}
            if (it == g_streams.end()) return;
        }
        st = it->second;
        g_streams.erase(it);
        g_retired.push_back(st);                             // ~100 bytes per destroyed stream, kept so that no reference to its lock dangles
    }
    std::lock_guard<std::recursive_mutex> l2(st->mu);
    free_state_memory(*st, false);
}

extern "C" int sg_workspace_release(void) {
    std::vector<std::shared_ptr<StreamState>> all;
    {
        std::lock_guard<std::mutex> lock(g_streams_mu);
        for (auto& kv : g_streams) all.push_back(kv.second);
    }
    for (auto& st : all) {                                   // one state at a time, g_streams_mu released: a launch sequence on another thread
        std::lock_guard<std::recursive_mutex> l2(st->mu);    // holds its state's lock and may look the map up meanwhile without a deadlock
        free_state_memory(*st, true);                        // (the reduction scratch comes back on its next use)
    }
    return SG_OK;
}

namespace {

// how to walk band [k_lo, k_hi] of an (n_frames x n_bins) spectrum; `flat` = the band is every bin, so the data
// is contiguous and both it and a packed output can be read 16 B per lane
inline Rows rows_for(int64_t n_frames, int n_bins, int k_lo, int k_hi, bool* flat) {
    const int width = k_hi - k_lo + 1;
    *flat = width == n_bins && n_frames > 0;
    if (!*flat) return Rows{n_frames, n_bins, width, width};
    const int64_t total = n_frames * n_bins;
    const int64_t n_rows = (total + kFlatRow - 1) / kFlatRow;
    return Rows{n_rows, kFlatRow, kFlatRow, static_cast<int>(total - (n_rows - 1) * kFlatRow)};
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
int minmax_t(const void* spec, int64_t n_frames, int n_bins, int k_lo, int k_hi, void* mm, hipStream_t s) {
    std::lock_guard<std::recursive_mutex> seq(launch_sequence_mutex(s));
    T* const parts = static_cast<T*>(reduction_scratch(s));
    if (!parts) { set_error("minmax: no scratch memory"); return SG_ERR_HIP; }
    bool flat;
    const Rows r = rows_for(n_frames, n_bins, k_lo, k_hi, &flat);
    const T* const src = static_cast<const T*>(spec) + (flat ? 0 : k_lo);
    const unsigned g = n_frames > 0 ? grid_for(r.n_rows * 64, kMaxParts) : 0;
    if (g && flat && aligned16(src))
        hipLaunchKernelGGL((minmax_kernel<T, true>), dim3(g), dim3(kThreads), 0, s, src, r, parts);
    else if (g)
        hipLaunchKernelGGL((minmax_kernel<T, false>), dim3(g), dim3(kThreads), 0, s, src, r, parts);
    hipLaunchKernelGGL(minmax_fold_kernel<T>, dim3(1), dim3(kThreads), 0, s, parts, static_cast<int>(g), static_cast<T*>(mm));
    return after_launch("minmax");
}

template <typename T>
int normalise_t(const void* spec, int64_t n_frames, int n_bins, int k_lo, int k_hi, int log_scale, double gmax,
                void* img, void* mm, hipStream_t s) {
    int rc = minmax_t<T>(spec, n_frames, n_bins, k_lo, k_hi, mm, s);
    if (rc != SG_OK || n_frames == 0) return rc;
    bool flat;
    const Rows r = rows_for(n_frames, n_bins, k_lo, k_hi, &flat);
    const T* const src = static_cast<const T*>(spec) + (flat ? 0 : k_lo);
    const unsigned g = grid_for(r.n_rows * 64);
    if (flat && aligned16(src) && aligned16(img))
        hipLaunchKernelGGL((normalise_kernel<T, true>), dim3(g), dim3(kThreads), 0, s, src, r, log_scale, static_cast<T>(gmax),
                           static_cast<const T*>(mm), static_cast<T*>(img));
    else
        hipLaunchKernelGGL((normalise_kernel<T, false>), dim3(g), dim3(kThreads), 0, s, src, r, log_scale, static_cast<T>(gmax),
                           static_cast<const T*>(mm), static_cast<T*>(img));
    return after_launch("normalise");
}

// img <- (img - lo) / (hi - lo), or 0 when hi - lo <= 1e-6 (PlotEngine.py:130-131), mm = (lo, hi) on the device
__global__ __launch_bounds__(kThreads) void rescale_kernel(float* img, int64_t n, const float* mm) {
    const float lo = mm[0], range = mm[1] - mm[0];
    const bool degenerate = !(range > 1e-6f);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    const int64_t i0 = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(img) & 15) == 0) {
        float4* const v4 = reinterpret_cast<float4*>(img);
        for (int64_t i = i0; i < n / 4; i += stride) {
            float4 v = v4[i];
            v.x = degenerate ? 0.f : (v.x - lo) / range; v.y = degenerate ? 0.f : (v.y - lo) / range;
            v.z = degenerate ? 0.f : (v.z - lo) / range; v.w = degenerate ? 0.f : (v.w - lo) / range;
            v4[i] = v;
        }
        for (int64_t i = (n / 4) * 4 + i0; i < n; i += stride) img[i] = degenerate ? 0.f : (img[i] - lo) / range;
    } else {
        for (int64_t i = i0; i < n; i += stride) img[i] = degenerate ? 0.f : (img[i] - lo) / range;
    }
}

// colour mapping with the min-max rescale folded in: rgba = lut[int(((img - lo) / (hi - lo)) * 256)]
__global__ __launch_bounds__(kThreads) void colormap_affine_kernel(const float* img, int64_t n, const float* mm, const uchar4* lut, uchar4* rgba) {
    const float lo = mm[0], range = mm[1] - mm[0];
    const bool degenerate = !(range > 1e-6f);
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads) {
        const float d = img[i];
        const float v = degenerate ? 0.f : (d - lo) / range;
        uchar4 c = make_uchar4(0, 0, 0, 0);
        if (v == v) {
            int idx = static_cast<int>(v * 256.0f);
            idx = idx < 0 ? 0 : (idx > 255 ? 255 : idx);
            c = lut[idx];
        }
        rgba[i] = c;
    }
}

}  // namespace

int fold_minmax_f32(const float* parts, int n_parts, float* mm_dev, hipStream_t s) {
    hipLaunchKernelGGL(minmax_fold_kernel<float>, dim3(1), dim3(kThreads), 0, s, parts, n_parts, mm_dev);
    return after_launch("minmax fold");
}

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

int sg_colormap(const float* img_dev, int64_t n, const uint8_t* lut_dev, uint8_t* rgba_dev, void* stream) {
    if (!img_dev || !lut_dev || !rgba_dev || n < 0) { set_error("bad argument"); return SG_ERR_ARG; }
    if (n == 0) return SG_OK;
    hipLaunchKernelGGL(colormap_kernel, dim3(grid_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), img_dev, n,
                       reinterpret_cast<const uchar4*>(lut_dev), reinterpret_cast<uchar4*>(rgba_dev));
    return after_launch("colormap");
}

int sg_db_rescale(float* db_dev, int64_t n, const float* mm_dev, void* stream) {
    if (n < 0 || (n > 0 && (!db_dev || !mm_dev))) { set_error("sg_db_rescale: bad argument"); return SG_ERR_ARG; }
    if (n == 0) return SG_OK;
    hipLaunchKernelGGL(rescale_kernel, dim3(grid_for((n + 3) / 4)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), db_dev, n, mm_dev);
    return after_launch("db rescale");
}

int sg_colormap_db(const float* db_dev, int64_t n, const float* mm_dev, const uint8_t* lut_dev, uint8_t* rgba_dev, void* stream) {
    if (n < 0 || (n > 0 && (!db_dev || !mm_dev || !lut_dev || !rgba_dev))) { set_error("sg_colormap_db: bad argument"); return SG_ERR_ARG; }
    if (n == 0) return SG_OK;
    hipLaunchKernelGGL(colormap_affine_kernel, dim3(grid_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), db_dev, n, mm_dev,
                       reinterpret_cast<const uchar4*>(lut_dev), reinterpret_cast<uchar4*>(rgba_dev));
    return after_launch("colormap db");
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
        hipLaunchKernelGGL(slice_kernel<float>, dim3(grid_for(n_frames * 64)), dim3(kThreads), 0, s,
                           static_cast<const float*>(spec_dev), n_frames, n_bins, k_lo, width, static_cast<float*>(dst_dev));
    else if (dtype == SG_F64)
        hipLaunchKernelGGL(slice_kernel<double>, dim3(grid_for(n_frames * 64)), dim3(kThreads), 0, s,
                           static_cast<const double*>(spec_dev), n_frames, n_bins, k_lo, width, static_cast<double*>(dst_dev));
    else { set_error("bad dtype %d", dtype); return SG_ERR_ARG; }
    return after_launch("slice_bins");
}

int sg_band_sum(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int k_lo, int k_hi, void* band_dev, void* stream) {
    if (!spec_dev || !band_dev) { set_error("null pointer"); return SG_ERR_ARG; }
    if (int rc = check_band(n_frames, n_bins, k_lo, k_hi)) return rc;
    if (n_frames == 0) return SG_OK;
    auto s = static_cast<hipStream_t>(stream);
    const unsigned g = grid_for((n_frames + 3) / 4 * 64);
    if (dtype == SG_F32)
        hipLaunchKernelGGL(band_sum_kernel<float>, dim3(g), dim3(kThreads), 0, s, static_cast<const float*>(spec_dev), n_frames,
                           n_bins, k_lo, k_hi, static_cast<float*>(band_dev));
    else if (dtype == SG_F64)
        hipLaunchKernelGGL(band_sum_kernel<double>, dim3(g), dim3(kThreads), 0, s, static_cast<const double*>(spec_dev), n_frames,
                           n_bins, k_lo, k_hi, static_cast<double*>(band_dev));
    else { set_error("bad dtype %d", dtype); return SG_ERR_ARG; }
    return after_launch("band_sum");
}

int sg_band_features_batch(const void* band_dev, int dtype, int n_clips, int64_t n_frames, void* feat_dev, void* stream) {
    if (!band_dev || !feat_dev) { set_error("null pointer"); return SG_ERR_ARG; }
    if (n_frames < 0 || n_clips < 0) { set_error("negative sizes"); return SG_ERR_ARG; }
    const int64_t total = n_frames * n_clips;
    if (total == 0) return SG_OK;
    auto s = static_cast<hipStream_t>(stream);
    if (dtype == SG_F32)
        hipLaunchKernelGGL(band_features_kernel<float>, dim3(grid_for(total)), dim3(kThreads), 0, s,
                           static_cast<const float*>(band_dev), total, n_frames, static_cast<float*>(feat_dev));
    else if (dtype == SG_F64)
        hipLaunchKernelGGL(band_features_kernel<double>, dim3(grid_for(total)), dim3(kThreads), 0, s,
                           static_cast<const double*>(band_dev), total, n_frames, static_cast<double*>(feat_dev));
    else { set_error("bad dtype %d", dtype); return SG_ERR_ARG; }
    return after_launch("band_features");
}

int sg_band_features(const void* band_dev, int dtype, int64_t n_frames, void* feat_dev, void* stream) {
    return sg_band_features_batch(band_dev, dtype, 1, n_frames, feat_dev, stream);
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
        if (hi > lo) {
            if (b.k_end == 0 || lo < b.k_begin) b.k_begin = lo;
            if (hi > b.k_end) b.k_end = hi;
        }
    }
    auto s = static_cast<hipStream_t>(stream);
    if (n_frames == 0) { SG_HIP(hipMemsetAsync(sums_dev, 0, sizeof(double) * n_bands, s)); return SG_OK; }
    std::lock_guard<std::recursive_mutex> seq(launch_sequence_mutex(s));
    double* const parts = static_cast<double*>(reduction_scratch(s));
    if (!parts) { set_error("band_totals: no scratch memory"); return SG_ERR_HIP; }
    // one round of resident workgroups (4 per CU at this kernel's register use), never more waves than row groups
    int cap = device_cu_count() * 4;
    if (const char* e = SG_TUNE_ENV("SPECTRO_TOTALS_WG_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 8) cap = device_cu_count() * v; }   // tuning aid
    if (cap > kMaxParts) cap = kMaxParts;
    const unsigned g = grid_for((n_frames + kRowsPerStep - 1) / kRowsPerStep * 64, cap);
    if (dtype == SG_F32)
        hipLaunchKernelGGL(band_totals_kernel<float>, dim3(g), dim3(kThreads), 0, s, static_cast<const float*>(spec_dev), n_frames,
                           n_bins, b, parts);
    else if (dtype == SG_F64)
        hipLaunchKernelGGL(band_totals_kernel<double>, dim3(g), dim3(kThreads), 0, s, static_cast<const double*>(spec_dev), n_frames,
                           n_bins, b, parts);
    else { set_error("bad dtype %d", dtype); return SG_ERR_ARG; }
    hipLaunchKernelGGL(band_totals_fold_kernel, dim3(1), dim3(kThreads), 0, s, parts, static_cast<int>(g), n_bands, sums_dev);
    return after_launch("band_totals");
}

}  // extern "C"
