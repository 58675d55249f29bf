// stft_rblue_f64.hip -- the register chirp-z kernel in double precision ("rblued"): float64 STFT for EVEN nperseg = nfft <= 1024 that is no
// power of two.  Why: the reference's recordings arrive as float64 (SweepManager.py:135-136), scipy computes in the input's precision
// (scipy/signal/_spectral_py.py:1976-1981) and the GUI's nperseg spin box steps by 32 (GUI.py:87-89) -- the reference's OWN flow at nperseg
// 96, 480, 1000 ... ran on the LDS chirp-z kernel (one workgroup per frame, a barrier per radix-2 pass) until round 4.
//
// The frame structure is stft_rblue.hip's (read that file): N2 = n/2 packed complex points, their DFT as a circular convolution of length
// L = 512 (T = 1, N2 <= 256) or 1024 (T = 2, N2 <= 512) done by two passes of the register FFT, the second starting from the registers of the
// first; real-input split through the wave's slab.  The transform and its LDS handling are stft_rbig_f64.hip's: a complex double is 16 bytes,
// so slab and tables are separate real / imaginary planes of 8-byte elements and every LDS access is a ds_*_b64 with the f32 kernel's index
// maps (tools/sim_rbig.py).  Any hop (odd hops and unaligned clips take 8-byte loads).  L = 2048 (nperseg up to 2048) would need 256 value
// registers per lane and more LDS than a CU has for its tables: those sizes stay on the LDS kernel.
// Algorithmic HBM bytes per frame: hop*8 + (n/2+1)*8.
#include "spectro_internal.h"
#include "cfft_wave_f64.h"

#include <cmath>
#include <vector>

namespace sg {
namespace {

using namespace wavefft64;

template <int T> struct BlueDCfg {
    static constexpr int R = 8 * T, M = 64 * R;                          // L = M complex points
    static constexpr int kWaves = T == 1 ? 12 : 8;                       // one workgroup per CU (LDS: 146 / 138 KiB): three / two waves per SIMD
    static constexpr int kRowsIn = R / 2, kRowsOut = R / 2 + 1;
    // complex table entries (each a real and an imaginary plane element)
    static constexpr int kWin = 0;                                       // [kRowsIn][64]: (w[2m], w[2m+1]) -- a pair of reals, kept as one "complex"
    static constexpr int kChirp = kWin + kRowsIn * 64;                   // [kRowsIn][64]: c[m] = exp(-i pi m^2 / N2)
    static constexpr int kFilt = kChirp + kRowsIn * 64;                  // [R][64]: FFT_M(b) / M
    static constexpr int kStw = kFilt + M;                               // [kRowsOut][64]: exp(-2 pi i k / n)
    static constexpr int kTw1 = kStw + kRowsOut * 64;                    // [R - 1][64]
    static constexpr int kTw2 = kTw1 + (R - 1) * 64;                     // [7][64]
    static constexpr int kTabs = kTw2 + 7 * 64;
};

struct BlueDParams {
    const double* x;
    int64_t clip_stride;
    int n_frames, hop;
    int64_t total_frames;
    int n_waves;
    double* out;
    int64_t out_clip_stride;
    int n2;                    // nperseg / 2
    int aligned;               // every frame starts on a 16-byte boundary: one 16-byte load per point, else two 8-byte loads
    const double2* tabs;       // [kTabs] in the order of BlueDCfg (the window rows unscaled)
    double scale;
    int k_lo, k_hi;            // MODE 2: bins of the band
};

// MODE: 0 psd, 1 magnitude, 2 band power (A11)
template <int T, bool DETREND, int MODE>
__global__ __launch_bounds__((64 * BlueDCfg<T>::kWaves), (BlueDCfg<T>::kWaves / 4)) void stft_rblue_f64_kernel(const BlueDParams p) {
    using C = BlueDCfg<T>;
    constexpr int R = C::R, kWaves = C::kWaves;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Planes tab{lds, lds + C::kTabs};
    double* const slab = lds + 2 * C::kTabs + wave * 2 * kSlab;
    const Planes sl{slab, slab + kSlab};

    {   // tables into LDS, once per workgroup; sqrt of the PSD scale rides on the window (stft_r8x3.hip)
        const double sq = sqrt(MODE != 1 ? p.scale * 0.5 : p.scale * 0.25);
        for (int i = threadIdx.x; i < C::kTabs; i += 64 * kWaves) {
            const double2 v = p.tabs[i];
            const double f = i < C::kChirp ? sq : 1.0;
            tab.put(i, {v.x * f, v.y * f});
        }
    }
    __syncthreads();

    const int lw = xcd_remap(blockIdx.x, gridDim.x) * kWaves + wave;
    if (lw >= p.n_waves) return;
    int64_t g = p.total_frames * lw / p.n_waves;
    const int64_t g_end = p.total_frames * (lw + 1) / p.n_waves;

    const int n2 = p.n2;
    const double n_f = static_cast<double>(2 * n2);
    auto cfft = [&](cd (&d)[T][8], cd (&e)[T][8]) { cfft_wave_f64<T>(d, e, tab, C::kTw1, C::kTw2, sl, lane); };      // cfft_wave_f64.h

    int clip = static_cast<int>(g / p.n_frames);
    int f = static_cast<int>(g - static_cast<int64_t>(clip) * p.n_frames);
    for (; g < g_end; ++g) {
        double* const orow = p.out + static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(f) * (MODE == 2 ? 1 : n2 + 1);
        cd raw[C::kRowsIn];
        {
            const double* const src = p.x + static_cast<int64_t>(clip) * p.clip_stride + static_cast<int64_t>(f) * p.hop + 2 * lane;
            if (p.aligned) {                                 // wave-uniform
#pragma unroll
                for (int a = 0; a < C::kRowsIn; ++a) {
                    double2 v = make_double2(0.0, 0.0);
                    if (lane + 64 * a < n2) v = *reinterpret_cast<const double2*>(src + 128 * a);
                    raw[a] = {v.x, v.y};
                }
            } else {
#pragma unroll
                for (int a = 0; a < C::kRowsIn; ++a) raw[a] = lane + 64 * a < n2 ? cd{src[128 * a], src[128 * a + 1]} : cd{0.0, 0.0};
            }
        }
        double mean = 0.0;
        if (DETREND) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < C::kRowsIn; ++a) s += raw[a].x + raw[a].y;
            mean = wave_sum(s) / n_f;
        }
        // ---- a[m] = (x[2m] w[2m] + i x[2m+1] w[2m+1]) * c[m]; rows beyond N2 are zero (their window entries are) ----
        cd d[T][8], e[T][8];
#pragma unroll
        for (int a = 0; a < R; ++a) {
            if (a < C::kRowsIn) {
                const cd w = tab.get(C::kWin + lane + 64 * a), c = tab.get(C::kChirp + lane + 64 * a);
                d[a % T][a / T] = cmul(cd{(raw[a].x - mean) * w.x, (raw[a].y - mean) * w.y}, c);
            } else {
                d[a % T][a / T] = cd{0.0, 0.0};
            }
        }
        cfft(d, e);
#pragma unroll
        for (int c = 0; c < R; ++c) {                        // Y = conj(A * B)
            const cd y = cmul(e[c % T][c / T], tab.get(C::kFilt + lane + 64 * c));
            d[c % T][c / T] = cd{y.x, -y.y};
        }
        cfft(d, e);                                          // e = V; the convolution is conj(V) (1 / L is in B)
        // ---- Z[k] = c[k] * conj(V[k]), k = lane + 64 c < N2 -> slab[k]; slab[N2] := Z[0] (stored after every row: DS operations run in issue order) ----
#pragma unroll
        for (int c = 0; c < C::kRowsIn; ++c) {
            const cd w = tab.get(C::kChirp + lane + 64 * c), v = e[c % T][c / T];
            sl.put(lane + 64 * c, cmul(cd{v.x, -v.y}, w));
        }
        {
            const cd w = tab.get(C::kChirp + lane), v = e[0][0];
            if (lane == 0) sl.put(n2, cmul(cd{v.x, -v.y}, w));
        }
        wave_lds_fence();
        // ---- split + epilogue: bins k = lane + 64 c <= N2 ----
        double bsum = 0.0;
#pragma unroll
        for (int c = 0; c < C::kRowsOut; ++c) {
            if (64 * c <= n2) {                              // wave-uniform
                const int k = lane + 64 * c;
                const int kk = k <= n2 ? k : n2;             // lanes beyond the last bin read a valid entry and store nothing
                const cd A = sl.get(kk), B = sl.get(n2 - kk), tw = tab.get(C::kStw + lane + 64 * c);
                const cd S = {A.x + B.x, A.y - B.y};
                const cd D = {A.x - B.x, A.y + B.y};
                const cd X = {S.x + fma(tw.x, D.y, tw.y * D.x), S.y + fma(tw.y, D.y, -tw.x * D.x)};
                double pk = fma(X.x, X.x, X.y * X.y);
                if (MODE != 1 && (k == 0 || k == n2)) pk *= 0.5;
                if (MODE == 1) pk = sqrt(pk);
                if (MODE == 2) {
                    if (k <= n2 && k >= p.k_lo && k <= p.k_hi) bsum += pk;
                } else if (k <= n2) {
                    orow[k] = pk;
                }
            }
        }
        if (MODE == 2) {
            bsum = wave_sum(bsum);
            if (lane == 0) orow[0] = bsum;
        }
        wave_lds_fence();
        if (++f == p.n_frames) { f = 0; ++clip; }
    }
}

template <int T, bool DETREND>
int launch_td(const BlueDParams& prm, hipStream_t s, int mode, bool band, int n_cu) {
    using C = BlueDCfg<T>;
    auto k0 = stft_rblue_f64_kernel<T, DETREND, 0>;
    auto k1 = stft_rblue_f64_kernel<T, DETREND, 1>;
    auto k2 = stft_rblue_f64_kernel<T, DETREND, 2>;
    auto kern = band ? k2 : mode == SG_MODE_PSD ? k0 : k1;
    const size_t lds = (2 * static_cast<size_t>(C::kTabs) + static_cast<size_t>(C::kWaves) * 2 * kSlab) * sizeof(double);
    BlueDParams p = prm;
    int64_t n_waves = static_cast<int64_t>(n_cu) * C::kWaves;           // one workgroup per CU
    if (n_waves > p.total_frames) n_waves = p.total_frames;
    p.n_waves = static_cast<int>(n_waves);
    const int n_wg = static_cast<int>((n_waves + C::kWaves - 1) / C::kWaves);
    SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * C::kWaves), lds, s, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft_rblue_f64 launch");
}

template <int T>
int launch_t(const sg_plan& p, const StftArgs& a) {
    BlueDParams prm{};
    prm.x = static_cast<const double*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.total_frames = a.n_frames * a.n_clips;
    prm.out = static_cast<double*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.n2 = p.nfft / 2;
    prm.aligned = (p.hop % 2 == 0) && (a.clip_stride % 2 == 0 || a.n_clips == 1) && (reinterpret_cast<uintptr_t>(a.x) % 16 == 0);
    prm.tabs = static_cast<const double2*>(p.rb_wc_dev);
    prm.scale = p.scale;
    prm.k_lo = a.k_lo; prm.k_hi = a.k_hi;
    const bool band = a.band_mode != 0;                    // run_stft has checked: psd plan, 0 <= k_lo <= k_hi < n_bins
    return p.detrend == SG_DETREND_CONSTANT ? launch_td<T, true>(prm, a.stream, p.mode, band, p.n_cu)
                                            : launch_td<T, false>(prm, a.stream, p.mode, band, p.n_cu);
}

void host_fft(std::vector<long double>& re, std::vector<long double>& im) {      // radix-2, once per plan, for the filter spectrum
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
            const long double wr = cosl(ang), wi = sinl(ang);
            for (size_t i = k; i < n; i += len) {
                const size_t j = i + len / 2;
                const long double tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;
                re[j] = re[i] - tr; im[j] = im[i] - ti;
                re[i] += tr; im[i] += ti;
            }
        }
    }
}

}  // namespace

int rblue_f64_size(int nfft) { return nfft <= 512 ? 1 : 2; }

bool rblue_f64_can_run(const sg_plan& p, const StftArgs& a) {
    return p.dtype == SG_F64 && !a.in_i16 && !a.db_mode && a.mel_ipl == 0 && (reinterpret_cast<uintptr_t>(a.x) % 8 == 0) && a.n_frames <= INT32_MAX;
}

int launch_rblue_f64(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    return rblue_f64_size(p.nfft) == 1 ? launch_t<1>(p, a) : launch_t<2>(p, a);
}

// one table [kTabs] of (re, im) pairs in the order of BlueDCfg, computed in long double
int build_rblue_f64_tables(sg_plan& p, const std::vector<double>& window) {
    const int n = p.nfft, n2 = n / 2, T = rblue_f64_size(n), R = 8 * T, M = 64 * R;
    const int rows_in = R / 2, rows_out = R / 2 + 1;
    const int kWin = 0, kChirp = kWin + rows_in * 64, kFilt = kChirp + rows_in * 64, kStw = kFilt + M, kTw1 = kStw + rows_out * 64,
              kTw2 = kTw1 + (R - 1) * 64, kTabs = kTw2 + 7 * 64;
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<double> t(static_cast<size_t>(kTabs) * 2, 0.0);
    auto put = [&](int i, long double re, long double im) { t[2 * static_cast<size_t>(i)] = static_cast<double>(re); t[2 * static_cast<size_t>(i) + 1] = static_cast<double>(im); };
    std::vector<long double> cr(n2), ci(n2);                 // b[j] = exp(+i pi j^2 / n2); j^2 mod 2 n2 keeps the angle small
    for (int j = 0; j < n2; ++j) {
        const long long q = (static_cast<long long>(j) * j) % (2LL * n2);
        const long double ang = pi * static_cast<long double>(q) / static_cast<long double>(n2);
        cr[j] = cosl(ang); ci[j] = sinl(ang);
    }
    for (int m = 0; m < n2; ++m) {
        put(kWin + m, window[2 * m], window[2 * m + 1]);
        put(kChirp + m, cr[m], -ci[m]);                      // c[m] = conj b[m]
    }
    std::vector<long double> hr(M, 0.0L), hi(M, 0.0L);
    hr[0] = cr[0]; hi[0] = ci[0];
    for (int j = 1; j < n2; ++j) { hr[j] = hr[M - j] = cr[j]; hi[j] = hi[M - j] = ci[j]; }
    host_fft(hr, hi);
    for (int k = 0; k < M; ++k) put(kFilt + k, hr[k] / M, hi[k] / M);
    for (int k = 0; k <= n2; ++k) {
        const long double ang = -2.0L * pi * static_cast<long double>(k) / static_cast<long double>(n);
        put(kStw + k, cosl(ang), sinl(ang));
    }
    for (int j = 0; j < 64; ++j) {
        for (int r = 1; r < R; ++r) {
            const long double ang = -2.0L * pi * static_cast<long double>((static_cast<long long>(j) * r) % M) / M;
            put(kTw1 + (r - 1) * 64 + j, cosl(ang), sinl(ang));
        }
        for (int s = 1; s < 8; ++s) {
            const long double ang = -2.0L * pi * static_cast<long double>(((j & 7) * s) % 64) / 64.0L;
            put(kTw2 + (s - 1) * 64 + j, cosl(ang), sinl(ang));
        }
    }
    SG_HIP(hipMalloc(&p.rb_wc_dev, t.size() * sizeof(double)));
    SG_HIP(hipMemcpy(p.rb_wc_dev, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    return SG_OK;
}

}  // namespace sg
