// stft_rblue.hip -- register chirp-z kernel: f32 STFT for EVEN transform lengths that are not a power of two, nperseg = nfft <= 2048
// (the GUI's nperseg spin box steps by 32 over 32...8192, /root/reference/GUI.py:87-89: 96, 160, 1000, 1504 ... are first-class
// sizes of the reference, and scipy runs them through pocketfft's mixed radix / Bluestein code, scipy/signal/_spectral_py.py:2200-2202).
// Round 4: the LDS chirp-z kernel (stft_bluestein.hip: one 256-thread workgroup per frame, radix-2 passes with a workgroup barrier
// each) ran these sizes at ~30 M frames/s, 45x below the power-of-two register kernels.
//
// The real frame of n samples is packed into N2 = n/2 complex points z[m] = x[2m] + i x[2m+1] (as every register kernel here does);
// its N2-point DFT -- N2 is not a power of two -- is a chirp-z transform on the REGISTER FFT machinery of stft_rbig.hip:
//     Z[k] = c[k] * sum_m (z[m] c[m]) * b[k - m],     c[m] = exp(-i pi m^2 / N2),  b[j] = exp(+i pi j^2 / N2)
// the convolution being circular of length L = 512 (T = 1: N2 <= 256), 1024 (T = 2: N2 <= 512) or 2048 (T = 4: N2 <= 1024), L >= 2 N2 - 1:
//     A = FFT_L(z c)  ->  Y = conj(A * B), B = FFT_L(b) / L (host, double)  ->  V = FFT_L(Y)  ->  Z[k] = c[k] * conj(V[k]),  k < N2
// (an inverse transform is the forward one between two conjugations; the forward transform leaves Z[lane + 64 c] in register c of
// the lane, which is exactly the input layout, so the second transform starts from the registers of the first: no exchange between
// them).  The real-input split X[k] = 1/2 [(Z[k] + conj Z[N2-k]) - i w_n^k (Z[k] - conj Z[N2-k])], k = 0..N2, pairs k with N2 - k:
// not a lane mirror when N2 is not a multiple of 64, so Z goes through the wave's LDS slab once (contiguous stores, two loads per bin).
// One wavefront = one frame, no workgroup barrier in the frame loop; detrend / window / PSD scale as in stft_rbig.hip (the PSD scale
// rides on the window table).  Index maps of the transform: tools/sim_rbig.py.  Algorithmic HBM bytes per frame: hop*4 + (n/2+1)*4.
#include "spectro_internal.h"
#include "cfft_wave.h"

#include <cmath>
#include <vector>

namespace sg {
namespace {

using namespace wavefft;

template <int T> struct BlueCfg {
    static constexpr int R = 8 * T, M = 64 * R;                          // L = M complex points
    static constexpr int kOcc = T == 4 ? 2 : 4;                          // waves per SIMD
    static constexpr int kWaves = 8;                                     // per workgroup (T = 2: two workgroups per CU)
    static constexpr int kRowsIn = R / 2;                                // rows (of 64 complex) a frame can fill: N2 <= M / 2
    static constexpr int kRowsOut = R / 2 + 1;                           // rows of output bins k = 0..N2
    static constexpr int kSlab = T == 4 ? 2 * 8 * kS1 : 8 * kS1;         // float2 per wave: one exchange group, and N2 + 1 split entries
    // LDS tables, in float2 units
    static constexpr int kWc = 0;                                        // [kRowsIn][64] float4 (w[2m], w[2m+1], cos, -sin of the chirp)
    static constexpr int kFilt = kWc + 2 * kRowsIn * 64;                 // [R/2][64][2]: rows 2m, 2m+1 of a lane side by side
    static constexpr int kStw = kFilt + M;                               // [(kRowsOut+1)/2][64][2]
    static constexpr int kTw1 = kStw + ((kRowsOut + 1) / 2) * 128;       // [R/2][64][2] (R - 1 rows, padded)
    static constexpr int kTw2 = kTw1 + R * 64;                           // [7][64]
    static constexpr int kTabs = kTw2 + 7 * 64;
};

struct BlueParams {
    const float* x;
    int64_t clip_stride;
    int n_frames, hop;
    int64_t total_frames;
    int n_waves;
    float* out;
    int64_t out_clip_stride;
    int n2;                    // nperseg / 2
    int aligned;               // every frame starts on an 8-byte boundary (even hop, even clip stride, aligned base): one 8-byte load per point;
                               // else two 4-byte loads (the reference's own call at nperseg 1000 has hop 875, PlotEngine.py:113)
    const v4f* wc;             // [kRowsIn * 64]
    const float2* filt;        // [M]   FFT_M(b) / M, natural order
    const float2* stw;         // [kRowsOut * 64]  exp(-2 pi i k / n), k <= n2 (zero beyond)
    const float2* tw;          // [(R - 1) + 7][64]: t1[r-1][j] = exp(-2 pi i j r / M), t2[s-1][j] = exp(-2 pi i (j & 7) s / 64)
    float scale;
    int k_lo, k_hi;            // MODE 2: bins of the band
};

// MODE: 0 psd, 1 magnitude, 2 band power (A11)
template <int T, bool DETREND, int MODE>
__global__ __launch_bounds__((64 * BlueCfg<T>::kWaves), (BlueCfg<T>::kOcc)) void stft_rblue_kernel(const BlueParams p) {
    using C = BlueCfg<T>;
    constexpr int R = C::R, M = C::M, kWaves = C::kWaves;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2* const buf = lds + C::kTabs + wave * C::kSlab;

    // ---- tables -> LDS (once per workgroup) -------------------------------------------------------------------------------------
    {
        const float q_in = MODE != 1 ? p.scale * 0.5f : p.scale * 0.25f;     // sqrt of it rides on the window; bins 0 and N2 get 1/2 below
        const float sq = sqrtf(q_in);
        v4f* const wc = reinterpret_cast<v4f*>(lds + C::kWc);
        for (int i = threadIdx.x; i < C::kRowsIn * 64; i += 64 * kWaves) {
            v4f v = p.wc[i];
            v.x *= sq; v.y *= sq;
            wc[i] = v;
        }
        for (int i = threadIdx.x; i < M; i += 64 * kWaves) {                  // rows 2m, 2m+1 of a lane side by side (one ds_read_b128)
            const int r = i >> 6, l = i & 63;
            lds[C::kFilt + (r >> 1) * 128 + 2 * l + (r & 1)] = p.filt[i];
        }
        for (int i = threadIdx.x; i < C::kRowsOut * 64; i += 64 * kWaves) {
            const int r = i >> 6, l = i & 63;
            lds[C::kStw + (r >> 1) * 128 + 2 * l + (r & 1)] = p.stw[i];
        }
        for (int i = threadIdx.x; i < (R - 1 + 7) * 64; i += 64 * kWaves) {
            const int r = i >> 6, l = i & 63;
            const int dst = r < R - 1 ? C::kTw1 + (r >> 1) * 128 + 2 * l + (r & 1) : C::kTw2 + (r - (R - 1)) * 64 + l;
            lds[dst] = p.tw[i];
        }
    }
    __syncthreads();

    const int lw = xcd_remap(blockIdx.x, gridDim.x) * kWaves + wave;
    if (lw >= p.n_waves) return;
    int64_t g = p.total_frames * lw / p.n_waves;
    const int64_t g_end = p.total_frames * (lw + 1) / p.n_waves;

    const CfftLds fl = cfft_lds(lds + C::kTw1, lds + C::kTw2, buf, lane);       // the L-point transform of cfft_wave.h on this wave's slab
    auto cfft = [&](float2 (&d)[T][8], float2 (&e)[T][8]) { cfft_wave<T>(d, e, fl); };
    const int n2 = p.n2;
    const float n_f = static_cast<float>(2 * n2);           // the mean is a true division (n is not a power of two: a constant clip must detrend to 0 exactly, as in scipy)

    auto load_frame = [&](int clip, int f, float2 (&dst)[C::kRowsIn]) {
        const float* const src = p.x + static_cast<int64_t>(clip) * p.clip_stride + static_cast<int64_t>(f) * p.hop + 2 * lane;
        if (p.aligned) {                                     // wave-uniform
#pragma unroll
            for (int a = 0; a < C::kRowsIn; ++a)
                dst[a] = lane + 64 * a < n2 ? *reinterpret_cast<const float2*>(src + 128 * a) : make_float2(0.f, 0.f);
        } else {
#pragma unroll
            for (int a = 0; a < C::kRowsIn; ++a)
                dst[a] = lane + 64 * a < n2 ? make_float2(src[128 * a], src[128 * a + 1]) : make_float2(0.f, 0.f);
        }
    };

    int clip = static_cast<int>(g / p.n_frames);
    int f = static_cast<int>(g - static_cast<int64_t>(clip) * p.n_frames);
    float2 nxt[C::kRowsIn];
    if (g < g_end) load_frame(clip, f, nxt);
    const v4f* const wcl = reinterpret_cast<const v4f*>(lds + C::kWc) + lane;                 // + 64*a

    for (; g < g_end; ++g) {
        float* const orow = p.out + static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(f) * (MODE == 2 ? 1 : n2 + 1);
        const int clip_n = f + 1 == p.n_frames ? clip + 1 : clip, f_n = f + 1 == p.n_frames ? 0 : f + 1;
        float2 raw[C::kRowsIn];
#pragma unroll
        for (int a = 0; a < C::kRowsIn; ++a) raw[a] = nxt[a];
        {   // the run's last frame fetches itself again: an unconditional fetch keeps the old registers out of the loop's live set
            const bool more = g + 1 < g_end;
            load_frame(more ? clip_n : clip, more ? f_n : f, nxt);
        }
        float mean = 0.f;
        if (DETREND) {
            float s = raw[0].x + raw[0].y;
#pragma unroll
            for (int a = 1; a < C::kRowsIn; ++a) s += raw[a].x + raw[a].y;
            mean = wave_sum(s) / n_f;
        }
        // ---- a[m] = (x[2m] w[2m] + i x[2m+1] w[2m+1]) * c[m]; rows beyond N2 are zero (their window entries are) ---------------------
        float2 d[T][8], e[T][8];
#pragma unroll
        for (int a = 0; a < R; ++a) {
            if (a < C::kRowsIn) {
                const v4f w = wcl[64 * a];
                const float x0 = (raw[a].x - mean) * w.x, x1 = (raw[a].y - mean) * w.y;
                d[a % T][a / T] = make_float2(fmaf(x0, w.z, -x1 * w.w), fmaf(x0, w.w, x1 * w.z));
            } else {
                d[a % T][a / T] = make_float2(0.f, 0.f);
            }
        }
        cfft(d, e);
        // ---- Y = conj(A * B) -----------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int c = 0; c < R; c += 2) {
            const v4f fb = lds_get2(lds + C::kFilt + (c >> 1) * 128 + 2 * lane);
            const float2 y0 = cmul(e[c % T][c / T], make_float2(fb.x, fb.y));
            const float2 y1 = cmul(e[(c + 1) % T][(c + 1) / T], make_float2(fb.z, fb.w));
            d[c % T][c / T] = make_float2(y0.x, -y0.y);
            d[(c + 1) % T][(c + 1) / T] = make_float2(y1.x, -y1.y);
        }
        cfft(d, e);                                          // e = V; the convolution is conj(V) (1 / L is in B)
        // ---- Z[k] = c[k] * conj(V[k]), k = lane + 64 c < N2 (rows beyond are never read back) -> slab[k]; slab[N2] := Z[0] ---------------
#pragma unroll
        for (int c = 0; c < C::kRowsIn; ++c) {
            const v4f w = wcl[64 * c];
            const float2 v = e[c % T][c / T];
            const float2 z = make_float2(fmaf(w.z, v.x, w.w * v.y), fmaf(w.w, v.x, -w.z * v.y));      // (w.z + i w.w) * (v.x - i v.y)
            lds_put(buf + lane + 64 * c, z);
            if (c == 0 && lane == 0) lds_put(buf + n2, z);   // (lands after row (N2 / 64)'s stores only if that row is row 0: see below)
        }
        {   // Z[0] at index N2: stored again AFTER every row (row N2 / 64 has put a convolution tail there); DS operations run in issue order
            const v4f w = wcl[0];
            const float2 v = e[0][0];
            const float2 z = make_float2(fmaf(w.z, v.x, w.w * v.y), fmaf(w.w, v.x, -w.z * v.y));
            if (lane == 0) lds_put(buf + n2, z);
        }
        wave_lds_fence();
        // ---- split + epilogue: bins k = lane + 64 c <= N2 ----------------------------------------------------------------------------------
        float bsum = 0.f;
#pragma unroll
        for (int c = 0; c < C::kRowsOut; ++c) {
            if (64 * c <= n2) {                              // wave-uniform
                const int k = lane + 64 * c;
                const int kk = k <= n2 ? k : n2;             // lanes beyond the last bin read a valid entry and store nothing
                const float2 A = lds_get(buf + kk);
                const float2 B = lds_get(buf + (n2 - kk));
                v4f tw2 = {0.f, 0.f, 0.f, 0.f};
                tw2 = lds_get2(lds + C::kStw + (c >> 1) * 128 + 2 * lane);
                const float2 tw = c % 2 == 0 ? make_float2(tw2.x, tw2.y) : make_float2(tw2.z, tw2.w);
                const float2 S = make_float2(A.x + B.x, A.y - B.y);
                const float2 D = make_float2(A.x - B.x, A.y + B.y);
                const float2 X = make_float2(S.x + fmaf(tw.x, D.y, tw.y * D.x), S.y + fmaf(tw.y, D.y, -tw.x * D.x));
                float pk = fmaf(X.x, X.x, X.y * X.y);
                if (MODE != 1 && (k == 0 || k == n2)) pk *= 0.5f;
                if (MODE == 1) pk = sqrtf(pk);
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
        clip = clip_n;
        f = f_n;
    }
}

template <int T, bool DETREND>
int launch_td(const BlueParams& prm, hipStream_t s, int mode, bool band, int n_cu) {
    using C = BlueCfg<T>;
    auto k0 = stft_rblue_kernel<T, DETREND, 0>;
    auto k1 = stft_rblue_kernel<T, DETREND, 1>;
    auto k2 = stft_rblue_kernel<T, DETREND, 2>;
    auto kern = band ? k2 : mode == SG_MODE_PSD ? k0 : k1;
    const size_t lds = (static_cast<size_t>(C::kTabs) + static_cast<size_t>(C::kWaves) * C::kSlab) * sizeof(float2);
    const int wg_per_cu = static_cast<int>((160 * 1024) / lds) < 1 ? 1 : (static_cast<int>((160 * 1024) / lds) > C::kOcc * 4 / C::kWaves ? C::kOcc * 4 / C::kWaves : static_cast<int>((160 * 1024) / lds));
    BlueParams p = prm;
    int64_t n_waves = static_cast<int64_t>(n_cu) * wg_per_cu * C::kWaves;
    if (n_waves > p.total_frames) n_waves = p.total_frames;
    p.n_waves = static_cast<int>(n_waves);
    const int n_wg = static_cast<int>((n_waves + C::kWaves - 1) / C::kWaves);
    if (lds > 64 * 1024)
        SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * C::kWaves), lds, s, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft_rblue launch");
}

template <int T>
int launch_t(const sg_plan& p, const StftArgs& a) {
    BlueParams prm{};
    prm.x = static_cast<const float*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.total_frames = a.n_frames * a.n_clips;
    prm.out = static_cast<float*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.n2 = p.nfft / 2;
    prm.aligned = (p.hop % 2 == 0) && (a.clip_stride % 2 == 0 || a.n_clips == 1) && (reinterpret_cast<uintptr_t>(a.x) % 8 == 0);
    prm.wc = static_cast<const v4f*>(p.rb_wc_dev);
    prm.filt = static_cast<const float2*>(p.rb_filt_dev);
    prm.stw = static_cast<const float2*>(p.rb_stw_dev);
    prm.tw = static_cast<const float2*>(p.rb_tw_dev);
    prm.scale = static_cast<float>(p.scale);
    prm.k_lo = a.k_lo; prm.k_hi = a.k_hi;
    const bool band = a.band_mode != 0;                    // run_stft has checked: psd plan, 0 <= k_lo <= k_hi < n_bins
    return p.detrend == SG_DETREND_CONSTANT ? launch_td<T, true>(prm, a.stream, p.mode, band, p.n_cu)
                                            : launch_td<T, false>(prm, a.stream, p.mode, band, p.n_cu);
}

template <typename V>
int upload(void** dev, const std::vector<V>& host) {
    SG_HIP(hipMalloc(dev, host.size() * sizeof(V)));
    SG_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(V), hipMemcpyHostToDevice));
    return SG_OK;
}

}  // namespace

// Host-side double-precision radix-2 FFT, used once per plan for the filter spectrum
void host_fft_pow2(std::vector<double>& re, std::vector<double>& im) {
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

int rblue_size(int nfft) { return nfft <= 512 ? 1 : nfft <= 1024 ? 2 : 4; }     // T of a plan rblue_ok() accepts

// (odd hops and clips at odd strides run here too, with 4-byte loads; int16 input is converted first, spectro_api.hip)
bool rblue_can_run(const sg_plan&, const StftArgs& a) {
    return !a.in_i16 && (reinterpret_cast<uintptr_t>(a.x) % 4 == 0) && a.n_frames <= INT32_MAX;
}

int launch_rblue(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    const int T = rblue_size(p.nfft);
    return T == 1 ? launch_t<1>(p, a) : T == 2 ? launch_t<2>(p, a) : launch_t<4>(p, a);
}

// Tables of the register chirp-z kernel (computed in double): window pairs + input/output chirp, filter spectrum, split twiddles,
// and the per-lane twiddles of the L-point transform.
int build_rblue_tables(sg_plan& p, const std::vector<double>& window) {
    const int n = p.nfft, n2 = n / 2, T = rblue_size(n), R = 8 * T, M = 64 * R;
    const int rows_in = R / 2, rows_out = R / 2 + 1;
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<double> cr(n2), ci(n2);                      // b[j] = exp(+i pi j^2 / n2); j^2 mod 2 n2 keeps the angle small
    for (int j = 0; j < n2; ++j) {
        const long long q = (static_cast<long long>(j) * j) % (2LL * n2);
        const long double ang = pi * static_cast<long double>(q) / static_cast<long double>(n2);
        cr[j] = static_cast<double>(cosl(ang));
        ci[j] = static_cast<double>(sinl(ang));
    }
    std::vector<float> wc(static_cast<size_t>(rows_in) * 64 * 4, 0.0f);
    for (int m = 0; m < n2; ++m) {
        wc[4 * static_cast<size_t>(m) + 0] = static_cast<float>(window[2 * m]);
        wc[4 * static_cast<size_t>(m) + 1] = static_cast<float>(window[2 * m + 1]);
        wc[4 * static_cast<size_t>(m) + 2] = static_cast<float>(cr[m]);
        wc[4 * static_cast<size_t>(m) + 3] = static_cast<float>(-ci[m]);      // c[m] = conj b[m]
    }
    std::vector<double> hr(M, 0.0), hi(M, 0.0);
    hr[0] = cr[0]; hi[0] = ci[0];
    for (int j = 1; j < n2; ++j) { hr[j] = hr[M - j] = cr[j]; hi[j] = hi[M - j] = ci[j]; }
    host_fft_pow2(hr, hi);
    std::vector<float2> filt(M);
    for (int k = 0; k < M; ++k) filt[k] = make_float2(static_cast<float>(hr[k] / M), static_cast<float>(hi[k] / M));
    std::vector<float2> stw(static_cast<size_t>(rows_out) * 64, make_float2(0.f, 0.f));
    for (int k = 0; k <= n2; ++k) {
        const long double ang = -2.0L * pi * static_cast<long double>(k) / static_cast<long double>(n);
        stw[k] = make_float2(static_cast<float>(cosl(ang)), static_cast<float>(sinl(ang)));
    }
    std::vector<float2> tw(static_cast<size_t>(R - 1 + 7) * 64);
    for (int j = 0; j < 64; ++j) {
        for (int r = 1; r < R; ++r) {
            const long double ang = -2.0L * pi * static_cast<long double>((static_cast<long long>(j) * r) % M) / M;
            tw[(r - 1) * 64 + j] = make_float2(static_cast<float>(cosl(ang)), static_cast<float>(sinl(ang)));
        }
        for (int s = 1; s < 8; ++s) {
            const long double ang = -2.0L * pi * static_cast<long double>(((j & 7) * s) % 64) / 64.0L;
            tw[(R - 1 + s - 1) * 64 + j] = make_float2(static_cast<float>(cosl(ang)), static_cast<float>(sinl(ang)));
        }
    }
    if (int rc = upload(&p.rb_wc_dev, wc)) return rc;
    if (int rc = upload(&p.rb_filt_dev, filt)) return rc;
    if (int rc = upload(&p.rb_stw_dev, stw)) return rc;
    return upload(&p.rb_tw_dev, tw);
}

}  // namespace sg
