// stft_stockham.hip -- general STFT kernel: any nperseg <= nfft = 2^q (2 <= nfft <= 8192), f32 or f64,
// every detrend / scaling / mode of scipy.signal.spectrogram for real one-sided input.
//
// Same algorithm as the reference's call chain (scipy/signal/_spectral_py.py:2180-2202 and :2125-2134),
// organised for a workgroup instead of a wavefront: a group of TPF threads owns one frame in LDS,
//   load -> (sum x, sum u*x) -> detrend (none | constant | linear) -> window -> zero pad
//   -> nfft/2-point complex Stockham autosort FFT (radix-4 passes + one radix-2 when needed) on the even/odd packed signal
//   -> split pass -> |X|^2*scale (one-sided doubling) | |X| | X | arg X -> HBM.
// Small transforms pack several frames into one 256-thread workgroup.  The r8x3 kernel
// (stft_r8x3.hip) is the fast path for the headline size; this kernel is the correctness net for
// everything else and the building block of the Bluestein path.
#include "spectro_internal.h"

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
struct GenParams {
    const void* x;            // TIn samples
    int64_t clip_stride;
    int64_t n_frames;         // per clip
    int64_t total_frames;     // n_frames * n_clips
    int nperseg, nfft, hop, log2m;
    int tpf;                  // threads per frame (power of two, <= 256)
    int detrend, mode;
    T* out;
    int64_t out_clip_stride;
    const T* win;             // [nperseg]
    const Cx<T>* tw;          // [nfft/2]  exp(-2*pi*i*k/nfft)
    T scale;                  // psd scale (already sqrt'ed for the non-psd modes)
    int band_mode, k_lo, k_hi;
    int tw_lds;               // 1: the twiddle table (nfft/2 complex) is staged in LDS once per workgroup
};

// Sum of `v` over the tpf threads of a frame group (tpf a power of two, groups aligned): xor butterflies inside a wavefront,
// and for groups of 128 / 256 threads one trip through `red` (2 or 4 doubles per group).  Every thread gets the total.
// One workgroup barrier in the wide case, none otherwise (the tree over LDS it replaces took log2(tpf) barriers).
__device__ __forceinline__ double group_sum(double v, int tpf, int tid, double* red) {
    const int w = tpf < 64 ? tpf : 64;
    for (int o = w >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (tpf > 64) {
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        v = 0.0;
        for (int i = 0; i < (tpf >> 6); ++i) v += red[i];
        __syncthreads();                        // red is reused by the next reduction
    }
    return v;
}

template <typename T, typename TIn>
__global__ __launch_bounds__(kThreads) void stft_stockham_kernel(const GenParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = p.nfft >> 1;
    const int tpf = p.tpf;
    const int groups = kThreads / tpf;
    const int g = threadIdx.x / tpf;
    const int tid = threadIdx.x - g * tpf;

    // LDS carve: per group two complex buffers of M (= nfft reals each), then the reduction scratch
    const size_t buf_elems = static_cast<size_t>(p.nfft);            // reals per buffer
    T* const base = reinterpret_cast<T*>(smem_raw);
    T* bufA = base + static_cast<size_t>(g) * 2 * buf_elems;
    T* bufB = bufA + buf_elems;
    double* const red_base = reinterpret_cast<double*>(base + static_cast<size_t>(groups) * 2 * buf_elems);
    double* const red = red_base + static_cast<size_t>(g) * 2 * tpf;
    // twiddles: staged in LDS once per workgroup when there is room (a stage's butterflies otherwise wait for L1 / L2)
    const Cx<T>* twp = p.tw;
    if (p.tw_lds) {
        Cx<T>* const tw_l = reinterpret_cast<Cx<T>*>(red_base + static_cast<size_t>(groups) * 2 * tpf);
        for (int i = threadIdx.x; i < M; i += kThreads) tw_l[i] = p.tw[i];
        twp = tw_l;
        __syncthreads();
    }

    const int nbins = M + 1;
    const int n = p.nperseg;

    for (int64_t fbase = static_cast<int64_t>(blockIdx.x) * groups; fbase < p.total_frames;
         fbase += static_cast<int64_t>(gridDim.x) * groups) {
        const int64_t fr = fbase + g;
        const bool active = fr < p.total_frames;
        const int64_t clip = active ? fr / p.n_frames : 0;
        const int64_t f = active ? fr - clip * p.n_frames : 0;
        const TIn* src = static_cast<const TIn*>(p.x) + clip * p.clip_stride + f * p.hop;

        // ---- load + partial sums (A2, A3) -----------------------------------
        double s0 = 0.0, s1 = 0.0;
        if (active) {
            for (int i = tid; i < p.nfft; i += tpf) {
                T v = T(0);
                if (i < n) {
                    v = static_cast<T>(src[i]);
                    s0 += static_cast<double>(v);
                    if (p.detrend == SG_DETREND_LINEAR) s1 += static_cast<double>(v) * static_cast<double>(i + 1);
                }
                bufA[i] = v;
            }
        }
        T c0 = T(0), c1 = T(0);    // trend = c0 + c1 * (i+1)/n
        if (p.detrend != SG_DETREND_NONE) {
            const double sx = group_sum(s0, tpf, tid, red);
            if (p.detrend == SG_DETREND_CONSTANT) {
                c0 = static_cast<T>(sx / n);
            } else {
                const double sux = group_sum(s1, tpf, tid, red) / n;             // u_i = (i+1)/n
                // least squares line through (u_i, x_i): x ~ beta*u + alpha
                const double dn = n;
                const double su = (dn + 1.0) * 0.5;                              // sum u
                const double suu = (dn + 1.0) * (2.0 * dn + 1.0) / (6.0 * dn);   // sum u^2
                const double den = dn * suu - su * su;
                const double beta = den != 0.0 ? (dn * sux - su * sx) / den : 0.0;
                const double alpha = (sx - beta * su) / dn;
                c0 = static_cast<T>(alpha);
                c1 = static_cast<T>(beta / dn);
            }
        }

        // ---- detrend + window (A3, A4), in place; bufA viewed as M complex ----
        if (active) {
            for (int i = tid; i < n; i += tpf) {
                const T v = bufA[i] - (c0 + c1 * static_cast<T>(i + 1));
                bufA[i] = v * p.win[i];
            }
        }
        __syncthreads();

        // ---- M-point complex Stockham autosort FFT (A5): radix-4 passes (two radix-2 stages per trip through LDS and per
        //      barrier), one radix-2 pass at the end when log2(M) is odd ---------------------------------------------------
        Cx<T>* src_c = reinterpret_cast<Cx<T>*>(bufA);
        Cx<T>* dst_c = reinterpret_cast<Cx<T>*>(bufB);
        const int half = M >> 1, quarter = M >> 2;
        int st = 0;
        for (; st + 1 < p.log2m; st += 2) {
            const int pp = 1 << st;
            if (active) {
                for (int i = tid; i < quarter; i += tpf) {
                    const int k = i & (pp - 1);
                    const int j = ((i - k) << 2) + k;
                    // w1 = exp(-2*pi*i*k/(4pp)) = tw[k * M/(2pp)], w2 = w1^2, w3 = w1^3 (index past the half table: negated)
                    const size_t i1 = static_cast<size_t>(k) * (M >> (st + 1));
                    const Cx<T> w1 = twp[i1], w2 = twp[2 * i1];
                    Cx<T> w3;
                    if (3 * i1 < static_cast<size_t>(M)) { w3 = twp[3 * i1]; }
                    else { const Cx<T> t = twp[3 * i1 - M]; w3 = {-t.x, -t.y}; }
                    const Cx<T> a0 = src_c[i], a1 = src_c[i + quarter], a2 = src_c[i + half], a3 = src_c[i + half + quarter];
                    const Cx<T> b1 = {a1.x * w1.x - a1.y * w1.y, a1.x * w1.y + a1.y * w1.x};
                    const Cx<T> b2 = {a2.x * w2.x - a2.y * w2.y, a2.x * w2.y + a2.y * w2.x};
                    const Cx<T> b3 = {a3.x * w3.x - a3.y * w3.y, a3.x * w3.y + a3.y * w3.x};
                    const Cx<T> s02 = {a0.x + b2.x, a0.y + b2.y}, d02 = {a0.x - b2.x, a0.y - b2.y};
                    const Cx<T> s13 = {b1.x + b3.x, b1.y + b3.y}, d13 = {b1.x - b3.x, b1.y - b3.y};
                    dst_c[j] = {s02.x + s13.x, s02.y + s13.y};
                    dst_c[j + pp] = {d02.x + d13.y, d02.y - d13.x};              // a0 - i b1 - b2 + i b3
                    dst_c[j + 2 * pp] = {s02.x - s13.x, s02.y - s13.y};
                    dst_c[j + 3 * pp] = {d02.x - d13.y, d02.y + d13.x};          // a0 + i b1 - b2 - i b3
                }
            }
            __syncthreads();
            Cx<T>* t = src_c; src_c = dst_c; dst_c = t;
        }
        if (st < p.log2m) {
            const int pp = 1 << st;
            if (active) {
                for (int i = tid; i < half; i += tpf) {
                    const int k = i & (pp - 1);
                    const int j = ((i - k) << 1) + k;
                    const Cx<T> w = twp[static_cast<size_t>(k) * (M >> st)];   // exp(-i*pi*k/pp)
                    const Cx<T> u0 = src_c[i];
                    const Cx<T> v = src_c[i + half];
                    const Cx<T> u1 = {v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x};
                    dst_c[j] = {u0.x + u1.x, u0.y + u1.y};
                    dst_c[j + pp] = {u0.x - u1.x, u0.y - u1.y};
                }
            }
            __syncthreads();
            Cx<T>* t = src_c; src_c = dst_c; dst_c = t;
        }
        const Cx<T>* Z = src_c;

        // ---- split pass + epilogue (A6) ---------------------------------------
        double bsum = 0.0;
        if (active) {
            const int64_t row = clip * p.out_clip_stride +
                                f * (p.mode == SG_MODE_COMPLEX ? 2 * static_cast<int64_t>(nbins) : nbins);
            for (int k = tid; k < nbins; k += tpf) {
                const Cx<T> A = Z[k == M ? 0 : k];
                const Cx<T> Bz = Z[(M - k) == M ? 0 : (M - k)];
                const T sx = A.x + Bz.x, sy = A.y - Bz.y;      // A + conj(B)
                const T dx = A.x - Bz.x, dy = A.y + Bz.y;      // A - conj(B)
                Cx<T> w;
                if (k == M) { w.x = T(-1); w.y = T(0); } else { w = twp[k]; }
                // X = (S - i*W*D)/2 ; i*W*D = i*(wx*dx - wy*dy + i*(wx*dy + wy*dx))
                const T tx = -(w.x * dy + w.y * dx), ty = w.x * dx - w.y * dy;
                const T xr = T(0.5) * (sx - tx), xi = T(0.5) * (sy - ty);
                if (p.mode == SG_MODE_PSD) {
                    T v = (xr * xr + xi * xi) * p.scale;
                    if (k != 0 && k != M) v *= T(2);
                    if (p.band_mode) {
                        if (k >= p.k_lo && k <= p.k_hi) bsum += static_cast<double>(v);
                    } else {
                        p.out[row + k] = v;
                    }
                } else if (p.mode == SG_MODE_MAGNITUDE) {
                    p.out[row + k] = t_sqrt<T>(xr * xr + xi * xi) * p.scale;
                } else if (p.mode == SG_MODE_COMPLEX) {
                    p.out[row + 2 * k] = xr * p.scale;
                    p.out[row + 2 * k + 1] = xi * p.scale;
                } else {
                    p.out[row + k] = t_atan2<T>(xi * p.scale, xr * p.scale);
                }
            }
        }
        if (p.band_mode) {
            const double tot = group_sum(bsum, tpf, tid, red);
            if (active && tid == 0) p.out[clip * p.out_clip_stride + f] = static_cast<T>(tot);
        }
        __syncthreads();     // buffers are rewritten by the next frame
    }
}

template <typename T, typename TIn>
int launch_t(const sg_plan& p, const StftArgs& a) {
    GenParams<T> prm{};
    const int M = p.nfft / 2;
    int log2m = 0;
    while ((1 << log2m) < M) ++log2m;
    prm.x = a.x;
    prm.clip_stride = a.clip_stride;
    prm.n_frames = a.n_frames;
    prm.total_frames = a.n_frames * a.n_clips;
    prm.nperseg = p.nperseg;
    prm.nfft = p.nfft;
    prm.hop = p.hop;
    prm.log2m = log2m;
    int tpf = M / 2;
    if (tpf < 1) tpf = 1;
    if (tpf > kThreads) tpf = kThreads;
    prm.tpf = tpf;
    prm.detrend = p.detrend;
    prm.mode = p.mode;
    prm.out = static_cast<T*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.win = static_cast<const T*>(p.win_dev);
    prm.tw = static_cast<const Cx<T>*>(p.tw_dev);
    prm.scale = static_cast<T>(p.mode == SG_MODE_PSD ? p.scale : std::sqrt(p.scale));
    prm.band_mode = a.band_mode;
    prm.k_lo = a.k_lo;
    prm.k_hi = a.k_hi;

    const int groups = kThreads / tpf;
    size_t lds = static_cast<size_t>(groups) * (2 * static_cast<size_t>(p.nfft) * sizeof(T) + 2 * tpf * sizeof(double));
    const size_t tw_bytes = static_cast<size_t>(M) * 2 * sizeof(T);
    // only when the table costs no workgroup per CU (f64 nfft 4096: 64 + 32 KiB would drop from 2 to 1 and measured 0.72x)
    prm.tw_lds = (lds + tw_bytes <= 160 * 1024 && (160 * 1024) / (lds + tw_bytes) == (160 * 1024) / lds) ? 1 : 0;
    if (prm.tw_lds) lds += tw_bytes;
    if (lds > 160 * 1024) {
        set_error("stockham: nfft=%d in %s needs %zu B of LDS (> 160 KiB)", p.nfft, sizeof(T) == 8 ? "f64" : "f32", lds);
        return SG_ERR_UNSUPPORTED;
    }
    auto kern = stft_stockham_kernel<T, TIn>;
    if (lds > 64 * 1024) {
        SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(lds)));
    }
    int64_t n_wg = (prm.total_frames + groups - 1) / groups;
    const int64_t cap = static_cast<int64_t>(p.n_cu) * 16;
    if (n_wg > cap) n_wg = cap;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(n_wg)), dim3(kThreads), lds, a.stream, prm);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "stft_stockham launch");
    return SG_OK;
}

}  // namespace

int launch_stockham(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    if (p.dtype == SG_F64) {
        if (a.in_i16) { set_error("int16 input needs an f32 plan"); return SG_ERR_ARG; }
        return launch_t<double, double>(p, a);
    }
    return a.in_i16 ? launch_t<float, int16_t>(p, a) : launch_t<float, float>(p, a);
}

}  // namespace sg
