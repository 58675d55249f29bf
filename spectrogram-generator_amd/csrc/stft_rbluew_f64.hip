// stft_rbluew_f64.hip -- the wide register chirp-z kernel in double precision ("rbluewd"): float64 STFT for 1024 < nperseg = nfft <= 8192 that
// is no power of two, W = 2 (n <= 2048, n % 4 == 0), 4 (n <= 4096, n % 8 == 0) or 8 (n <= 8192, n % 16 == 0) wavefronts per frame.  Why: the
// reference's recordings arrive as float64 (SweepManager.py:135-136), scipy computes in the input's precision
// (scipy/signal/_spectral_py.py:1976-1981) and the GUI's nperseg spin box runs to 8192 in steps of 32 (GUI.py:87-89): the reference's OWN flow
// at nperseg 1056 ... 8160 ran on the LDS chirp-z kernel (0.9-5 M frames/s) until round 4.
//
// Structure: stft_rbluew.hip's (read that file).  The N2 = n/2 packed complex points are split by decimation in time over the W wavefronts
// of a frame group, wave w computes F_w = DFT_mp(z[W a + w]), mp = N2 / W <= 512, as a chirp-z transform on two passes of the 1024-point
// register FFT (cfft_wave_f64.h), multiplies by W_N2^(w k0) and leaves G_w in LDS; after a workgroup barrier wave r adds the W terms of
// Z[k0 + r mp]; Z goes to LDS and the waves share the real-input split row by row.  Differences from the f32 kernel: complex values in LDS
// are separate real / imaginary planes (ds_*_b64 only); the window rows of a wave are loop-invariant and live in REGISTERS (16 doubles per
// lane), which is what lets tables (57 KB) and exchange slabs (74 KB) of eight waves fit the LDS at every W; four workgroup barriers per frame,
// the frame mean riding on the prefetched samples of the next frame as there (prefetch issued after the second transform, its partial sums
// exchanged with Z: the register file has no room for it during the transforms).  W = 8 has no room for either across the transforms (190-390
// spilled dwords per lane when tried): its waves fetch samples and window rows at the top of the frame and the mean costs a fifth barrier.
// nperseg 8192 itself runs here too, without the chirp: see WideDCfg<W, EX>.
// Algorithmic HBM bytes per frame: hop*8 + (n/2+1)*8.
#include "spectro_internal.h"
#include "cfft_wave_f64.h"

#include <cmath>
#include <vector>

namespace sg {
namespace {

using namespace wavefft64;

// EX ("exact", nperseg 8192 on W = 4 waves): mp = 1024 IS the transform length, so the wave's DFT is one plain 1024-point transform -- no chirp,
// no filter, no second pass, every bin used; 16 rows per wave instead of 8, G slots of 1024 entries, W_N2^k0 as a per-lane times a per-row factor
template <int W, bool EX = false> struct WideDCfg {
    static constexpr int T = 2, R = 8 * T, M = 64 * R;                   // the sub-transform: L = 1024
    static constexpr int kWaves = 8, kGroups = kWaves / W;               // per workgroup
    static constexpr int kRows = EX ? 16 : 8;                            // rows of 64 points a wave fills: mp <= 512 (EX: mp = 1024)
    static constexpr bool kPrefetch = !EX && W < 8;                      // W = 8, EX: no room for the next frame's samples or resident window rows next to the transforms
    static constexpr int kRowsD = EX ? 17 : 9;                           // rows of output bins per wave: (N2 + 1) / 64 / W, rounded up
    static constexpr int kSlot = EX ? 1024 : kSlab;                      // elements per plane and wave: its exchange slab (kSlab), then G_w[k0]
    static constexpr int kRegion = W * kSlot + 8;                        // per plane and frame group; after the G exchange it holds Z[0..N2]
    // complex table entries (each a real and an imaginary plane element) -- the device table has this order, then the window rows
    static constexpr int kChirp = 0;                                     // [512] c[a] = exp(-i pi a^2 / mp)                      (EX: none)
    static constexpr int kFilt = kChirp + (EX ? 0 : 512);                // [R][64]: FFT_M(b) / M                                 (EX: none)
    static constexpr int kTw1 = kFilt + (EX ? 0 : M);                    // [R - 1][64]
    static constexpr int kTw2 = kTw1 + (R - 1) * 64;                     // [7][64]
    static constexpr int kCtw = kTw2 + 7 * 64;                           // [512] exp(-2 pi i k0 / N2)   (EX: [64] exp(-2 pi i lane / N2), then [16] exp(-2 pi i 64 c / N2), padded)
    static constexpr int kSrow = kCtw + (EX ? 128 : 512);                // [64] exp(-2 pi i lane / n), then [.] exp(-2 pi i 64 rho / n), rho <= 71
    static constexpr int kTabs = kSrow + 192;
    static constexpr int kWinStride = EX ? 1024 : 512;
    static constexpr int kWinDev = kTabs;                                // device table only: [W][kWinStride] (w[2j], w[2j+1]), j = W a + w, zero for a >= mp
    static constexpr int kMisc = 16;                                     // doubles: partial sums [kGroups][W], band partials [kGroups][W]
    static constexpr size_t kLdsBytes = (2 * static_cast<size_t>(kTabs) + 2 * static_cast<size_t>(kGroups) * kRegion + kMisc) * sizeof(double);
};
static_assert(WideDCfg<2>::kLdsBytes <= 160 * 1024 && WideDCfg<8>::kLdsBytes <= 160 * 1024 && WideDCfg<4, true>::kLdsBytes <= 160 * 1024, "LDS of a CU");

struct WideDParams {
    const double* x;
    int64_t clip_stride;
    int n_frames, hop;
    int64_t total_frames;
    int n_groups, iters;       // frame groups of the launch (<= total_frames), frames per group rounded up
    double* out;
    int64_t out_clip_stride;
    int n2, mp;                // nperseg / 2, n2 / W
    int aligned;               // every frame starts on a 16-byte boundary: one 16-byte load per point, else two 8-byte loads
    const double2* tabs;       // [kTabs + W * kWinStride]
    double scale;
    int k_lo, k_hi;            // MODE 2: bins of the band
};

// MODE: 0 psd, 1 magnitude, 2 band power (A11)
template <int W, bool DETREND, int MODE, bool EX>
__global__ __launch_bounds__((64 * WideDCfg<W, EX>::kWaves), 2) void stft_rbluew_f64_kernel(const WideDParams p) {
    using C = WideDCfg<W, EX>;
    constexpr int T = C::T, R = C::R, kWaves = C::kWaves;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave / W, w = wave % W;                  // frame group of the workgroup, this wave's residue class / spectrum part
    const Planes tab{lds, lds + C::kTabs};
    double* const rg = lds + 2 * C::kTabs + grp * 2 * C::kRegion;
    const Planes region{rg, rg + C::kRegion};
    const Planes sl{rg + w * C::kSlot, rg + C::kRegion + w * C::kSlot};       // this wave's slot inside the group's region: exchange slab, then G_w
    double* const psum = lds + 2 * C::kTabs + 2 * C::kGroups * C::kRegion + grp * W;
    double* const bpart = psum + kWaves;

    for (int i = threadIdx.x; i < C::kTabs; i += 64 * kWaves) {               // tables into LDS, once per workgroup
        const double2 v = p.tabs[i];
        tab.put(i, {v.x, v.y});
    }
    __syncthreads();

    // per-lane views: every access below is one of these bases plus a compile-time offset (with a full index expression per access the
    // compiler keeps an address register per table row and plane -- ~90 at W = 8 -- and spills them)
    const Planes tab_l = tab.at(lane), sl_l = sl.at(lane);
    const Planes region_l = region.at(lane);
    const int n2 = p.n2, mp = p.mp;
    const double n_f = static_cast<double>(2 * n2);
    const cd lane_tw = tab.get(C::kSrow + lane);
    // this wave's window rows are loop-invariant: kept in registers where the transforms leave room (kPrefetch), else fetched again
    // with every frame's samples (L2 hits); sqrt of the PSD scale rides on them
    cd win[C::kRows];
    const double sq = sqrt(MODE != 1 ? p.scale * 0.5 : p.scale * 0.25);           // bins 0 and N2 get 1/2 below
    auto load_window = [&]() {
#pragma unroll
        for (int a = 0; a < C::kRows; ++a) {
            const double2 v = p.tabs[C::kWinDev + w * C::kWinStride + lane + 64 * a];
            win[a] = {v.x * sq, v.y * sq};
        }
    };
    if (C::kPrefetch) load_window();
    const int lg = xcd_remap(blockIdx.x, gridDim.x) * C::kGroups + grp;
    int64_t g = lg < p.n_groups ? p.total_frames * lg / p.n_groups : 0;
    const int64_t g_end = lg < p.n_groups ? p.total_frames * (lg + 1) / p.n_groups : 0;
    int clip = static_cast<int>(g / p.n_frames);
    int f = static_cast<int>(g - static_cast<int64_t>(clip) * p.n_frames);

    auto load_frame = [&](int cl, int fr, cd (&dst)[C::kRows]) {
        const double* const src = p.x + static_cast<int64_t>(cl) * p.clip_stride + static_cast<int64_t>(fr) * p.hop + 2 * w + 2 * W * lane;
        if (p.aligned) {                                     // wave-uniform
#pragma unroll
            for (int a = 0; a < C::kRows; ++a) {
                double2 v = make_double2(0.0, 0.0);
                if (lane + 64 * a < mp) v = *reinterpret_cast<const double2*>(src + 128 * W * a);
                dst[a] = {v.x, v.y};
            }
        } else {
#pragma unroll
            for (int a = 0; a < C::kRows; ++a) dst[a] = lane + 64 * a < mp ? cd{src[128 * W * a], src[128 * W * a + 1]} : cd{0.0, 0.0};
        }
    };
    auto part_sum = [&](const cd (&v)[C::kRows]) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < C::kRows; ++a) s += v[a].x + v[a].y;
        return wave_sum(s);
    };
    auto group_mean = [&]() {
        double s = psum[0];
#pragma unroll
        for (int v = 1; v < W; ++v) s += psum[v];
        return s / n_f;
    };

    cd nxt[C::kRows];
    if (C::kPrefetch) load_frame(clip, f, nxt);              // (a group without frames reads frame 0 of clip 0 and stores nothing)
    double mean = 0.0;
    if (DETREND && C::kPrefetch) {
        const double s = part_sum(nxt);
        if (lane == 0) psum[w] = s;
        __syncthreads();
        mean = group_mean();
        __syncthreads();                                     // the loop's first partial sums land after every wave has read these
    }

    for (int it = 0; it < p.iters; ++it, ++g) {
        const bool active = g < g_end;                       // uniform over the frame group
        double* const orow = p.out + static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(f) * (MODE == 2 ? 1 : n2 + 1);
        const bool more = g + 1 < g_end;
        const int clip_n = !more ? clip : f + 1 == p.n_frames ? clip + 1 : clip, f_n = !more ? f : f + 1 == p.n_frames ? 0 : f + 1;
        if (!C::kPrefetch) {                                 // samples and window rows fetched here; the mean costs a fifth barrier
            load_window();
            load_frame(clip, f, nxt);
            if (DETREND) {
                const double s = part_sum(nxt);
                if (lane == 0) psum[w] = s;
                __syncthreads();
                mean = group_mean();
            }
        }
        cd d[T][8], e[T][8];
        // ---- a[m] = (x[2j] w[2j] + i x[2j+1] w[2j+1]) * c[a], j = W a + w; rows beyond mp are zero (their window entries are) ----
#pragma unroll
        for (int a = 0; a < R; ++a) {
            if (a < C::kRows) {
                const cd xw = {(nxt[a].x - mean) * win[a].x, (nxt[a].y - mean) * win[a].y};
                d[a % T][a / T] = EX ? xw : cmul(xw, tab_l.get(C::kChirp + 64 * a));
            } else {
                d[a % T][a / T] = cd{0.0, 0.0};
            }
        }
        cfft_wave_f64<T>(d, e, tab, C::kTw1, C::kTw2, sl, lane);
        if (!EX) {
#pragma unroll
            for (int c = 0; c < R; ++c) {                    // Y = conj(A * B)
                const cd y = cmul(e[c % T][c / T], tab_l.get(C::kFilt + 64 * c));
                d[c % T][c / T] = cd{y.x, -y.y};
            }
            cfft_wave_f64<T>(d, e, tab, C::kTw1, C::kTw2, sl, lane);         // e = V; the convolution is conj(V) (1 / L is in B)
        }
        if (C::kPrefetch) load_frame(clip_n, f_n, nxt);      // prefetch, issued where the register file has room for it (the group's last frame fetches itself again)
        // ---- G_w[k0] = W_N2^(w k0) * c[k0] * conj(V[k0]) -> this wave's slot (EX: F_w[k0] is the transform itself) ----
#pragma unroll
        for (int c = 0; c < C::kRows; ++c) {
            const cd v = e[c % T][c / T];
            cd z = EX ? v : cmul(cd{v.x, -v.y}, tab_l.get(C::kChirp + 64 * c));
            if (w != 0) {                                    // wave-uniform: t^w by squaring
                const cd t = EX ? cmul(tab_l.get(C::kCtw), tab.get(C::kCtw + 64 + c)) : tab_l.get(C::kCtw + 64 * c);
                const cd t2 = cmul(t, t);
                cd pw = (w & 1) ? t : cd{1.0, 0.0};
                if (w & 2) pw = (w & 1) ? cmul(pw, t2) : t2;
                if (W > 4 && (w & 4)) { const cd t4 = cmul(t2, t2); pw = (w & 3) ? cmul(pw, t4) : t4; }
                z = cmul(z, pw);
            }
            sl_l.put(64 * c, z);
        }
        __syncthreads();                                     // (1) every G_w of the workgroup is in LDS
        // ---- Z[k0 + w mp] = sum_v W_W^(v w) G_v[k0] ----
        // (the coefficients are rebuilt per frame from an opaque copy of w: held across the transforms they cost 4 W registers the
        //  transforms need -- 240 spilled dwords at W = 8)
        double bw_re[W], bw_im[W];
        {
            int wv = w;
            asm volatile("" : "+s"(wv));
#pragma unroll
            for (int v = 0; v < W; ++v) {
                constexpr double h = 0.70710678118654752440;
                const int q = ((v * wv) % W) * (8 / W);                           // eighth turns clockwise
                bw_re[v] = q == 0 ? 1.0 : q == 4 ? -1.0 : (q == 2 || q == 6) ? 0.0 : (q == 1 || q == 7) ? h : -h;
                bw_im[v] = (q == 0 || q == 4) ? 0.0 : q == 2 ? -1.0 : q == 6 ? 1.0 : (q == 1 || q == 3) ? -h : h;
            }
        }
        cd z[C::kRows];
#pragma unroll
        for (int c = 0; c < C::kRows; ++c) {
            cd acc = region_l.get(64 * c);              // v = 0: coefficient 1
#pragma unroll
            for (int v = 1; v < W; ++v) {
                const cd gv = region_l.get(v * C::kSlot + 64 * c);
                acc.x += gv.x * bw_re[v] - gv.y * bw_im[v];
                acc.y += gv.x * bw_im[v] + gv.y * bw_re[v];
            }
            z[c] = acc;
            if (W > 4) __builtin_amdgcn_sched_barrier(0);    // (the scheduler would otherwise issue all 128 plane reads first and spill their registers)
        }
        __syncthreads();                                     // (2) every wave has taken its G values: the region becomes Z[0..N2]
        const Planes zrow = region.at(w * mp + lane);
#pragma unroll
        for (int c = 0; c < C::kRows; ++c)
            if (lane + 64 * c < mp) zrow.put(64 * c, z[c]);
        if (w == 0 && lane == 0) region.put(n2, z[0]);       // Z[N2] := Z[0]
        if (DETREND && C::kPrefetch) {                       // the next frame's samples have arrived by now: their partial sum travels with Z
            const double s = part_sum(nxt);
            if (lane == 0) psum[w] = s;
        }
        __syncthreads();                                     // (3)
        if (DETREND && C::kPrefetch) mean = group_mean();
        // ---- split + epilogue: this wave's rows of the bins k = 0..N2 ----
        // (row addresses are rebuilt per frame from opaque copies of w and N2: hoisted out of the loop they are five registers per row
        //  that the transforms need)
        double bsum = 0.0;
        int w_d = w, n2_d = n2;
        asm volatile("" : "+s"(w_d), "+s"(n2_d));
#pragma unroll
        for (int cc = 0; cc < C::kRowsD; ++cc) {
            const int rho = w_d + W * cc;
            if (64 * rho <= n2) {                            // wave-uniform
                const int k = lane + 64 * rho;
                const int kk = k <= n2_d ? k : n2_d;         // lanes beyond the last bin read a valid entry and store nothing
                const cd A = region.get(kk), B = region.get(n2_d - kk);
                const cd tw = cmul(lane_tw, tab.get(C::kSrow + 64 + rho));
                const cd S = {A.x + B.x, A.y - B.y};
                const cd D = {A.x - B.x, A.y + B.y};
                const cd X = {S.x + fma(tw.x, D.y, tw.y * D.x), S.y + fma(tw.y, D.y, -tw.x * D.x)};
                double pk = fma(X.x, X.x, X.y * X.y);
                if (MODE != 1 && (k == 0 || k == n2)) pk *= 0.5;
                if (MODE == 1) pk = sqrt(pk);
                if (MODE == 2) {
                    if (k <= n2 && k >= p.k_lo && k <= p.k_hi) bsum += pk;
                } else if (k <= n2 && active) {
                    orow[k] = pk;
                }
            }
        }
        if (MODE == 2) {
            bsum = wave_sum(bsum);
            if (lane == 0) bpart[w] = bsum;
        }
        __syncthreads();                                     // (4) the region is free for the next frame's transforms
        if (MODE == 2 && w == 0 && active && lane == 0) {
            double s = bpart[0];
#pragma unroll
            for (int v = 1; v < W; ++v) s += bpart[v];
            orow[0] = s;
        }
        clip = clip_n;
        f = f_n;
    }
}

template <int W, bool DETREND, bool EX>
int launch_wd(const WideDParams& prm, hipStream_t s, int mode, bool band, int n_cu) {
    using C = WideDCfg<W, EX>;
    auto k0 = stft_rbluew_f64_kernel<W, DETREND, 0, EX>;
    auto k1 = stft_rbluew_f64_kernel<W, DETREND, 1, EX>;
    auto k2 = stft_rbluew_f64_kernel<W, DETREND, 2, EX>;
    auto kern = band ? k2 : mode == SG_MODE_PSD ? k0 : k1;
    WideDParams p = prm;
    int64_t n_groups = static_cast<int64_t>(n_cu) * C::kGroups;                 // one workgroup per CU
    if (n_groups > p.total_frames) n_groups = p.total_frames;
    p.n_groups = static_cast<int>(n_groups);
    p.iters = static_cast<int>((p.total_frames + n_groups - 1) / n_groups);
    const int n_wg = static_cast<int>((n_groups + C::kGroups - 1) / C::kGroups);
    SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(C::kLdsBytes)));
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * C::kWaves), C::kLdsBytes, s, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft_rbluew_f64 launch");
}

template <int W, bool EX>
int launch_w(const sg_plan& p, const StftArgs& a) {
    WideDParams prm{};
    prm.x = static_cast<const double*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.total_frames = a.n_frames * a.n_clips;
    prm.out = static_cast<double*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.n2 = p.nfft / 2;
    prm.mp = p.nfft / 2 / W;
    prm.aligned = (p.hop % 2 == 0) && (a.clip_stride % 2 == 0 || a.n_clips == 1) && (reinterpret_cast<uintptr_t>(a.x) % 16 == 0);
    prm.tabs = static_cast<const double2*>(p.rb_wc_dev);
    prm.scale = p.scale;
    prm.k_lo = a.k_lo; prm.k_hi = a.k_hi;
    const bool band = a.band_mode != 0;                    // run_stft has checked: psd plan, 0 <= k_lo <= k_hi < n_bins
    return p.detrend == SG_DETREND_CONSTANT ? launch_wd<W, true, EX>(prm, a.stream, p.mode, band, p.n_cu)
                                            : launch_wd<W, false, EX>(prm, a.stream, p.mode, band, p.n_cu);
}

void host_fft_ld(std::vector<long double>& re, std::vector<long double>& im) {      // radix-2, once per plan, for the filter spectrum
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

template <int W, bool EX>
void fill_tables(std::vector<double>& t, const std::vector<double>& window, int n) {
    using C = WideDCfg<W, EX>;
    constexpr int R = C::R, M = C::M;
    const int n2 = n / 2, mp = n2 / W;
    const long double pi = 3.14159265358979323846264338327950288L;
    t.assign(2 * (static_cast<size_t>(C::kTabs) + W * C::kWinStride), 0.0);
    auto put = [&](int i, long double re, long double im) { t[2 * static_cast<size_t>(i)] = static_cast<double>(re); t[2 * static_cast<size_t>(i) + 1] = static_cast<double>(im); };
    for (int w = 0; w < W; ++w)
        for (int a = 0; a < mp; ++a) put(C::kWinDev + w * C::kWinStride + a, window[2 * (W * a + w)], window[2 * (W * a + w) + 1]);
    if (!EX) {
        std::vector<long double> br(mp), bi(mp);             // b[j] = exp(+i pi j^2 / mp); j^2 mod 2 mp keeps the angle small
        for (int j = 0; j < mp; ++j) {
            const long long q = (static_cast<long long>(j) * j) % (2LL * mp);
            const long double ang = pi * static_cast<long double>(q) / static_cast<long double>(mp);
            br[j] = cosl(ang); bi[j] = sinl(ang);
            put(C::kChirp + j, br[j], -bi[j]);               // c[j] = conj b[j]
        }
        std::vector<long double> hr(M, 0.0L), hi(M, 0.0L);
        hr[0] = br[0]; hi[0] = bi[0];
        for (int j = 1; j < mp; ++j) { hr[j] = hr[M - j] = br[j]; hi[j] = hi[M - j] = bi[j]; }
        host_fft_ld(hr, hi);
        for (int k = 0; k < M; ++k) put(C::kFilt + k, hr[k] / M, hi[k] / M);
        for (int k0 = 0; k0 < mp; ++k0) {
            const long double ang = -2.0L * pi * static_cast<long double>(k0) / static_cast<long double>(n2);
            put(C::kCtw + k0, cosl(ang), sinl(ang));
        }
    } else {                                                 // W_N2^k0, k0 = lane + 64 c, as a per-lane times a per-row factor
        for (int l = 0; l < 64; ++l) {
            const long double ang = -2.0L * pi * static_cast<long double>(l) / static_cast<long double>(n2);
            put(C::kCtw + l, cosl(ang), sinl(ang));
        }
        for (int c = 0; c < C::kRows; ++c) {
            const long double ang = -2.0L * pi * static_cast<long double>(64 * c) / static_cast<long double>(n2);
            put(C::kCtw + 64 + c, cosl(ang), sinl(ang));
        }
    }
    for (int l = 0; l < 64; ++l) {
        for (int r = 1; r < R; ++r) {
            const long double ang = -2.0L * pi * static_cast<long double>((static_cast<long long>(l) * r) % M) / M;
            put(C::kTw1 + (r - 1) * 64 + l, cosl(ang), sinl(ang));
        }
        for (int s = 1; s < 8; ++s) {
            const long double ang = -2.0L * pi * static_cast<long double>(((l & 7) * s) % 64) / 64.0L;
            put(C::kTw2 + (s - 1) * 64 + l, cosl(ang), sinl(ang));
        }
        const long double ang = -2.0L * pi * static_cast<long double>(l) / static_cast<long double>(n);
        put(C::kSrow + l, cosl(ang), sinl(ang));
    }
    for (int rho = 0; 64 * rho <= n2; ++rho) {
        const long double ang = -2.0L * pi * static_cast<long double>(64 * rho) / static_cast<long double>(n);
        put(C::kSrow + 64 + rho, cosl(ang), sinl(ang));
    }
}

}  // namespace

// wavefronts per frame of a plan rbluewd_ok() accepts (spectro_api.hip); 8192 itself: four waves, each ONE 1024-point transform (EX)
int rbluew_f64_size(int nfft) { return nfft <= 2048 ? 2 : nfft <= 4096 || nfft == 8192 ? 4 : 8; }

bool rbluew_f64_can_run(const sg_plan& p, const StftArgs& a) {
    return p.dtype == SG_F64 && !a.in_i16 && !a.db_mode && a.mel_ipl == 0 && (reinterpret_cast<uintptr_t>(a.x) % 8 == 0) && a.n_frames <= INT32_MAX;
}

int launch_rbluew_f64(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    if (p.nfft == 8192) return launch_w<4, true>(p, a);
    const int W = rbluew_f64_size(p.nfft);
    return W == 2 ? launch_w<2, false>(p, a) : W == 4 ? launch_w<4, false>(p, a) : launch_w<8, false>(p, a);
}

// one table of (re, im) pairs in the order of WideDCfg (LDS part, then the window rows per wave), computed in long double
int build_rbluew_f64_tables(sg_plan& p, const std::vector<double>& window) {
    std::vector<double> t;
    const int W = rbluew_f64_size(p.nfft);
    if (p.nfft == 8192) fill_tables<4, true>(t, window, p.nfft);
    else if (W == 2) fill_tables<2, false>(t, window, p.nfft); else if (W == 4) fill_tables<4, false>(t, window, p.nfft); else fill_tables<8, false>(t, window, p.nfft);
    SG_HIP(hipMalloc(&p.rb_wc_dev, t.size() * sizeof(double)));
    SG_HIP(hipMemcpy(p.rb_wc_dev, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    return SG_OK;
}

}  // namespace sg
