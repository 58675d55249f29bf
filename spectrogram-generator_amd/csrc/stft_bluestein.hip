// stft_bluestein.hip -- STFT for transform lengths the radix-2 kernels cannot take: any nfft that is
// not a power of two (the GUI's nperseg spin box steps by 32 and accepts typed values, GUI.py:87-89, and
// scipy clamps nperseg to the signal length for short signals, _spectral_py.py:2245-2249).
//
// Chirp-z (Bluestein):  with b[j] = exp(+i*pi*j^2/n)
//     X[k] = conj(b[k]) * sum_m (y[m]*conj(b[m])) * b[k-m]
// i.e. one circular convolution of length L = 2^q >= 2n-1 done with two in-place radix-2 FFTs in LDS (round 4: two stages per pass):
//     forward DIF (natural in, bit-reversed out)  ->  * H (filter spectrum stored bit-reversed, 1/L folded in)
//     -> inverse DIT (bit-reversed in, natural out).  No bit-reversal pass, a single L-point complex buffer.
// One 256-thread workgroup owns one frame at a time.  Framing / detrend / window / epilogue are the same
// code path as the Stockham kernel (scipy/signal/_spectral_py.py:2180-2202, :2125-2134).
// The L-point buffer lives in LDS while it fits (f32: nfft <= 8192, f64: nfft <= 4096); larger transforms -- f64 signals
// with a non-power-of-two nperseg above 4096, which the GUI's 32..8192 spin box and scipy's nperseg := len(x) clamp both
// reach, and every power of two beyond the Stockham kernel -- run the same butterflies on a per-workgroup slice of a
// per-stream HBM workspace (stream_workspace(): grows on demand, shared by the calls of a stream, so a plan stays re-entrant per stream);
// that slice is L2-resident (<= 1 MiB per workgroup at the largest GUI size) and the path is a correctness net for
// GUI-sized calls, not a throughput kernel.
#include "spectro_internal.h"

#include <cmath>

namespace sg {
namespace {

constexpr int kThreads = 256;

template <typename T> struct Cx { T x, y; };

template <typename T> __device__ __forceinline__ T t_sqrt(T v);
template <> __device__ __forceinline__ float t_sqrt<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double t_sqrt<double>(double v) { return sqrt(v); }
template <typename T> __device__ __forceinline__ T t_atan2(T y, T x);
template <> __device__ __forceinline__ float t_atan2<float>(float y, float x) { return atan2f(y, x); }
template <> __device__ __forceinline__ double t_atan2<double>(double y, double x) { return atan2(y, x); }

template <typename T>
struct BsParams {
    const T* x;
    int64_t clip_stride, n_frames, total_frames;
    int nperseg, nfft, hop, L, log2L;
    int detrend, mode;
    T* out;
    int64_t out_clip_stride;
    const T* win;            // [nperseg]
    const Cx<T>* chirp;      // [nfft]   b[j] = exp(+i*pi*j^2/nfft)
    const Cx<T>* filt;       // [L]      FFT_L(h)/L in bit-reversed order
    const Cx<T>* tw;         // [L/2]    exp(-2*pi*i*k/L)
    T scale;
    Cx<T>* work;             // GLOBAL: [gridDim.x][L] convolution buffers in HBM
};

// GLOBAL: the buffer is this workgroup's slice of p.work (visible to the workgroup's waves across __syncthreads():
// they share the CU's write-through L1), only the reduction scratch is in LDS.
template <typename T, bool GLOBAL>
__global__ __launch_bounds__(kThreads) void stft_bluestein_kernel(const BsParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Cx<T>* const C = GLOBAL ? p.work + static_cast<size_t>(blockIdx.x) * p.L : reinterpret_cast<Cx<T>*>(smem_raw);
    double* const red = GLOBAL ? reinterpret_cast<double*>(smem_raw) : reinterpret_cast<double*>(reinterpret_cast<Cx<T>*>(smem_raw) + p.L);
    const int tid = threadIdx.x;
    const int n = p.nperseg, L = p.L, halfL = L >> 1;
    const int nbins = p.nfft / 2 + 1;
    const bool odd = (p.nfft & 1) != 0;

    for (int64_t fr = blockIdx.x; fr < p.total_frames; fr += gridDim.x) {
        const int64_t clip = fr / p.n_frames, f = fr - clip * p.n_frames;
        const T* src = p.x + clip * p.clip_stride + f * p.hop;

        // ---- load, partial sums, zero fill ----
        double s0 = 0.0, s1 = 0.0;
        for (int i = tid; i < L; i += kThreads) {
            T v = T(0);
            if (i < n) {
                v = src[i];
                s0 += static_cast<double>(v);
                s1 += static_cast<double>(v) * static_cast<double>(i + 1);
            }
            C[i] = {v, T(0)};
        }
        T c0 = T(0), c1 = T(0);
        if (p.detrend != SG_DETREND_NONE) {
            red[tid] = s0;
            red[kThreads + tid] = s1;
            __syncthreads();
            for (int s = kThreads >> 1; s > 0; s >>= 1) {
                if (tid < s) { red[tid] += red[tid + s]; red[kThreads + tid] += red[kThreads + tid + s]; }
                __syncthreads();
            }
            const double sx = red[0], sux = red[kThreads] / n;
            if (p.detrend == SG_DETREND_CONSTANT) {
                c0 = static_cast<T>(sx / n);
            } else {
                const double dn = n, su = (dn + 1.0) * 0.5, suu = (dn + 1.0) * (2.0 * dn + 1.0) / (6.0 * dn);
                const double den = dn * suu - su * su;
                const double beta = den != 0.0 ? (dn * sux - su * sx) / den : 0.0;
                c0 = static_cast<T>((sx - beta * su) / dn);
                c1 = static_cast<T>(beta / dn);
            }
        }
        __syncthreads();

        // ---- detrend, window, pre-chirp: a[m] = y[m] * conj(b[m]) ----
        for (int i = tid; i < n; i += kThreads) {
            const T y = (C[i].x - (c0 + c1 * static_cast<T>(i + 1))) * p.win[i];
            const Cx<T> b = p.chirp[i];
            C[i] = {y * b.x, -y * b.y};
        }
        __syncthreads();

        // ---- forward DIF: two radix-2 stages per pass (round 4: half the workgroup barriers; the data order after a fused pass is that of the two
        //      single passes, so the bit-reversed filter table is unchanged); one single stage closes an odd log2 L ----
        {
            int st = 0;
            for (; st + 1 < p.log2L; st += 2) {
                const int q = L >> (st + 2);
                for (int i = tid; i < (L >> 2); i += kThreads) {
                    const int r = i & (q - 1);
                    const int j = ((i - r) << 2) + r;
                    const Cx<T> x0 = C[j], x1 = C[j + q], x2 = C[j + 2 * q], x3 = C[j + 3 * q];
                    const Cx<T> wa0 = p.tw[static_cast<size_t>(r) << st], wa1 = p.tw[static_cast<size_t>(r + q) << st], wb = p.tw[static_cast<size_t>(r) << (st + 1)];
                    const Cx<T> y0 = {x0.x + x2.x, x0.y + x2.y}, d02 = {x0.x - x2.x, x0.y - x2.y};
                    const Cx<T> y1 = {x1.x + x3.x, x1.y + x3.y}, d13 = {x1.x - x3.x, x1.y - x3.y};
                    const Cx<T> y2 = {d02.x * wa0.x - d02.y * wa0.y, d02.x * wa0.y + d02.y * wa0.x};
                    const Cx<T> y3 = {d13.x * wa1.x - d13.y * wa1.y, d13.x * wa1.y + d13.y * wa1.x};
                    const Cx<T> e01 = {y0.x - y1.x, y0.y - y1.y}, e23 = {y2.x - y3.x, y2.y - y3.y};
                    C[j] = {y0.x + y1.x, y0.y + y1.y};
                    C[j + q] = {e01.x * wb.x - e01.y * wb.y, e01.x * wb.y + e01.y * wb.x};
                    C[j + 2 * q] = {y2.x + y3.x, y2.y + y3.y};
                    C[j + 3 * q] = {e23.x * wb.x - e23.y * wb.y, e23.x * wb.y + e23.y * wb.x};
                }
                __syncthreads();
            }
            for (; st < p.log2L; ++st) {
                const int half = halfL >> st;
                for (int i = tid; i < halfL; i += kThreads) {
                    const int r = i & (half - 1);
                    const int j = ((i - r) << 1) + r;
                    const Cx<T> u = C[j], v = C[j + half];
                    const Cx<T> w = p.tw[static_cast<size_t>(r) << st];
                    const T dx = u.x - v.x, dy = u.y - v.y;
                    C[j] = {u.x + v.x, u.y + v.y};
                    C[j + half] = {dx * w.x - dy * w.y, dx * w.y + dy * w.x};
                }
                __syncthreads();
            }
        }
        // ---- pointwise multiply with the filter spectrum (both bit-reversed) ----
        for (int i = tid; i < L; i += kThreads) {
            const Cx<T> a = C[i], h = p.filt[i];
            C[i] = {a.x * h.x - a.y * h.y, a.x * h.y + a.y * h.x};
        }
        __syncthreads();
        // ---- inverse DIT: the mirror image -- a single stage first when log2 L is odd, then two stages per pass ----
        {
            int st = p.log2L - 1;
            if (p.log2L & 1) {
                const int half = halfL >> st;
                for (int i = tid; i < halfL; i += kThreads) {
                    const int r = i & (half - 1);
                    const int j = ((i - r) << 1) + r;
                    const Cx<T> w = p.tw[static_cast<size_t>(r) << st];       // conj applied below
                    const Cx<T> u = C[j], t = C[j + half];
                    const Cx<T> v = {t.x * w.x + t.y * w.y, t.y * w.x - t.x * w.y};
                    C[j] = {u.x + v.x, u.y + v.y};
                    C[j + half] = {u.x - v.x, u.y - v.y};
                }
                __syncthreads();
                --st;
            }
            for (; st >= 1; st -= 2) {                       // stages st (half q) then st - 1 (half 2q)
                const int q = halfL >> st;
                const int s0 = st - 1;
                for (int i = tid; i < (L >> 2); i += kThreads) {
                    const int r = i & (q - 1);
                    const int j = ((i - r) << 2) + r;
                    const Cx<T> x0 = C[j], x1 = C[j + q], x2 = C[j + 2 * q], x3 = C[j + 3 * q];
                    const Cx<T> wb = p.tw[static_cast<size_t>(r) << st], wa0 = p.tw[static_cast<size_t>(r) << s0], wa1 = p.tw[static_cast<size_t>(r + q) << s0];
                    const Cx<T> v1 = {x1.x * wb.x + x1.y * wb.y, x1.y * wb.x - x1.x * wb.y};      // x * conj(w)
                    const Cx<T> v3 = {x3.x * wb.x + x3.y * wb.y, x3.y * wb.x - x3.x * wb.y};
                    const Cx<T> a0 = {x0.x + v1.x, x0.y + v1.y}, a1 = {x0.x - v1.x, x0.y - v1.y};
                    const Cx<T> a2 = {x2.x + v3.x, x2.y + v3.y}, a3 = {x2.x - v3.x, x2.y - v3.y};
                    const Cx<T> u2 = {a2.x * wa0.x + a2.y * wa0.y, a2.y * wa0.x - a2.x * wa0.y};
                    const Cx<T> u3 = {a3.x * wa1.x + a3.y * wa1.y, a3.y * wa1.x - a3.x * wa1.y};
                    C[j] = {a0.x + u2.x, a0.y + u2.y};
                    C[j + 2 * q] = {a0.x - u2.x, a0.y - u2.y};
                    C[j + q] = {a1.x + u3.x, a1.y + u3.y};
                    C[j + 3 * q] = {a1.x - u3.x, a1.y - u3.y};
                }
                __syncthreads();
            }
        }

        // ---- post-chirp + epilogue ----
        const int64_t row = clip * p.out_clip_stride + f * (p.mode == SG_MODE_COMPLEX ? 2 * static_cast<int64_t>(nbins) : nbins);
        for (int k = tid; k < nbins; k += kThreads) {
            const Cx<T> a = C[k], b = p.chirp[k];
            const T xr = a.x * b.x + a.y * b.y, xi = a.y * b.x - a.x * b.y;      // a * conj(b)
            if (p.mode == SG_MODE_PSD) {
                T v = (xr * xr + xi * xi) * p.scale;
                if (k != 0 && (odd || k != p.nfft / 2)) v *= T(2);
                p.out[row + k] = v;
            } else if (p.mode == SG_MODE_MAGNITUDE) {
                p.out[row + k] = t_sqrt<T>(xr * xr + xi * xi) * p.scale;
            } else if (p.mode == SG_MODE_COMPLEX) {
                p.out[row + 2 * k] = xr * p.scale;
                p.out[row + 2 * k + 1] = xi * p.scale;
            } else {
                p.out[row + k] = t_atan2<T>(xi * p.scale, xr * p.scale);
            }
        }
        __syncthreads();
    }
}

template <typename T>
int launch_t(const sg_plan& p, const StftArgs& a) {
    BsParams<T> prm{};
    prm.x = static_cast<const T*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = a.n_frames;
    prm.total_frames = a.n_frames * a.n_clips;
    prm.nperseg = p.nperseg;
    prm.nfft = p.nfft;
    prm.hop = p.hop;
    prm.L = p.bs_len;
    int lg = 0;
    while ((1 << lg) < p.bs_len) ++lg;
    prm.log2L = lg;
    prm.detrend = p.detrend;
    prm.mode = p.mode;
    prm.out = static_cast<T*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.win = static_cast<const T*>(p.win_dev);
    prm.chirp = static_cast<const Cx<T>*>(p.bs_chirp_dev);
    prm.filt = static_cast<const Cx<T>*>(p.bs_filter_dev);
    prm.tw = static_cast<const Cx<T>*>(p.bs_tw_dev);
    prm.scale = static_cast<T>(p.mode == SG_MODE_PSD ? p.scale : std::sqrt(p.scale));
    const size_t red_bytes = 2 * kThreads * sizeof(double);
    const size_t lds = static_cast<size_t>(p.bs_len) * 2 * sizeof(T) + red_bytes;
    int64_t n_wg = prm.total_frames;
    if (lds <= 160 * 1024) {
        auto kern = stft_bluestein_kernel<T, false>;
        if (lds > 64 * 1024)
            SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        const int64_t cap = static_cast<int64_t>(p.n_cu) * 8;
        if (n_wg > cap) n_wg = cap;
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(n_wg)), dim3(kThreads), lds, a.stream, prm);
    } else {
        // oversized: one L-point buffer per workgroup in the stream's HBM workspace
        const int64_t cap = static_cast<int64_t>(p.n_cu);
        if (n_wg > cap) n_wg = cap;
        const size_t bytes = static_cast<size_t>(n_wg) * p.bs_len * 2 * sizeof(T);
        std::lock_guard<std::recursive_mutex> seq(launch_sequence_mutex(a.stream));      // (growing the workspace and launching on it: one step)
        void* const work = stream_workspace(a.stream, bytes);
        if (!work) { set_error("stft_bluestein: no memory for the %zu-byte convolution workspace", bytes); return SG_ERR_HIP; }
        prm.work = static_cast<Cx<T>*>(work);
        hipLaunchKernelGGL((stft_bluestein_kernel<T, true>), dim3(static_cast<unsigned>(n_wg)), dim3(kThreads), red_bytes, a.stream, prm);
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return hip_fail(le, "stft_bluestein (HBM workspace) launch");
        return SG_OK;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "stft_bluestein launch");
    return SG_OK;
}

template <typename T>
int upload_vec(void** dev, const std::vector<T>& host) {
    SG_HIP(hipMalloc(dev, host.size() * sizeof(T)));
    SG_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return SG_OK;
}

// Host-side double-precision radix-2 FFT used once per plan for the filter spectrum.
void host_fft(std::vector<double>& re, std::vector<double>& im) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    const long double pi = 3.14159265358979323846264338327950288L;
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t k = 0; k < len / 2; ++k) {
            const long double ang = -2.0L * pi * static_cast<long double>(k) / static_cast<long double>(len);
            const double wr = static_cast<double>(cosl(ang)), wi = static_cast<double>(sinl(ang));
            for (size_t i = k; i < n; i += len) {
                const size_t j = i + len / 2;
                const double tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;
                re[j] = re[i] - tr; im[j] = im[i] - ti;
                re[i] += tr; im[i] += ti;
            }
        }
    }
}

template <typename T>
int build_tables_t(sg_plan& p) {
    const int n = p.nfft, L = p.bs_len;
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<double> cr(n), ci(n);
    for (int j = 0; j < n; ++j) {
        const long long q = (static_cast<long long>(j) * j) % (2LL * n);          // j^2 mod 2n keeps the angle small
        const long double ang = pi * static_cast<long double>(q) / static_cast<long double>(n);
        cr[j] = static_cast<double>(cosl(ang));
        ci[j] = static_cast<double>(sinl(ang));
    }
    std::vector<T> chirp(2 * static_cast<size_t>(n));
    for (int j = 0; j < n; ++j) { chirp[2 * j] = static_cast<T>(cr[j]); chirp[2 * j + 1] = static_cast<T>(ci[j]); }
    std::vector<double> hr(L, 0.0), hi(L, 0.0);
    hr[0] = cr[0]; hi[0] = ci[0];
    for (int j = 1; j < n; ++j) { hr[j] = hr[L - j] = cr[j]; hi[j] = hi[L - j] = ci[j]; }
    host_fft(hr, hi);
    int lg = 0;
    while ((1 << lg) < L) ++lg;
    std::vector<T> filt(2 * static_cast<size_t>(L));
    for (int i = 0; i < L; ++i) {
        unsigned r = 0;
        for (int b = 0; b < lg; ++b) r |= ((static_cast<unsigned>(i) >> b) & 1u) << (lg - 1 - b);
        filt[2 * static_cast<size_t>(i)] = static_cast<T>(hr[r] / L);
        filt[2 * static_cast<size_t>(i) + 1] = static_cast<T>(hi[r] / L);
    }
    std::vector<T> tw(static_cast<size_t>(L));
    for (int k = 0; k < L / 2; ++k) {
        const long double ang = -2.0L * pi * static_cast<long double>(k) / static_cast<long double>(L);
        tw[2 * static_cast<size_t>(k)] = static_cast<T>(cosl(ang));
        tw[2 * static_cast<size_t>(k) + 1] = static_cast<T>(sinl(ang));
    }
    if (int rc = upload_vec<T>(&p.bs_chirp_dev, chirp)) return rc;
    if (int rc = upload_vec<T>(&p.bs_filter_dev, filt)) return rc;
    return upload_vec<T>(&p.bs_tw_dev, tw);
}

}  // namespace

int build_bluestein_tables(sg_plan& p) {
    int L = 1;
    while (L < 2 * p.nfft - 1) L <<= 1;
    if (p.nfft > (1 << 20)) {        // tables of 2^22 complex entries and a 64 MiB buffer per workgroup are where this stops
        set_error("nfft=%d is beyond the chirp-z path (nfft <= 1048576)", p.nfft);
        return SG_ERR_UNSUPPORTED;
    }
    p.bs_len = L;
    return p.dtype == SG_F64 ? build_tables_t<double>(p) : build_tables_t<float>(p);
}

int launch_bluestein(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    if (a.in_i16 || a.band_mode) { set_error("bluestein path takes float input and writes full spectra"); return SG_ERR_UNSUPPORTED; }
    return p.dtype == SG_F64 ? launch_t<double>(p, a) : launch_t<float>(p, a);
}

}  // namespace sg
