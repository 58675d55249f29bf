// stft_rbluew.hip -- "wide" register chirp-z kernel: f32 STFT for transform lengths 2048 < nperseg = nfft <= 8192 that are not a power of
// two, W = 2 (n <= 4096, n % 4 == 0) or W = 4 (n <= 8192, n % 8 == 0) wavefronts per frame.  (The GUI's nperseg spin box runs to 8192 in
// steps of 32, /root/reference/GUI.py:87-89; scipy takes these sizes through pocketfft's mixed radix / Bluestein code,
// scipy/signal/_spectral_py.py:2200-2202.)  Until round 4 they ran on the LDS chirp-z kernel (stft_bluestein.hip) at 2-9 M frames/s.
//
// The real frame is packed into N2 = n/2 complex points z[j] = x[2j] + i x[2j+1]; their DFT is split by decimation in time over the
// W wavefronts of a frame group (mp = N2 / W <= 1024 points each):
//     Z[k0 + r mp] = sum_w  W_W^(w r) * ( W_N2^(w k0) * F_w[k0] ),      F_w = DFT_mp( z[W a + w], a < mp ),   k0 < mp, r < W
// Wave w computes F_w exactly as stft_rblue.hip computes its whole frame: a chirp-z transform of size mp on TWO passes of the 2048-point
// register FFT (cfft_wave.h), with the tables of size mp -- chirp, filter spectrum and FFT twiddles are common to the W waves, only the
// window rows differ.  It multiplies by W_N2^(w k0) (one table of W_N2^k0, raised to the power w in registers) and leaves G_w in LDS;
// after a workgroup barrier wave r adds the W terms for its quarter / half of the spectrum, Z goes to LDS, and after another barrier the
// waves share the real-input split X[k] = 1/2 [(Z[k] + conj Z[N2-k]) - i w_n^k (Z[k] - conj Z[N2-k])], k = 0..N2, row by row (the split
// twiddle is the product of a per-lane and a per-row factor: a table of N2 entries would not fit).  The frame mean (detrend) needs every
// wave's samples: each wave sums the samples it has PREFETCHED for the next frame and the partial sums travel with G, so the mean costs
// no barrier of its own.  Four workgroup barriers per frame; a workgroup = 8 waves = 4 (W = 2) or 2 (W = 4) frame groups in lockstep,
// tables once per workgroup in LDS (145 / 161 KB), one workgroup per CU, two waves per SIMD.
// nperseg 8192 itself runs here too, without the chirp: see WideCfg<W, EX>.
// Algorithmic HBM bytes per frame: hop*4 + (n/2+1)*4.
#include "spectro_internal.h"
#include "cfft_wave.h"

#include <cmath>
#include <vector>

namespace sg {
namespace {

using namespace wavefft;

// EX ("exact", nperseg 8192 on W = 2 waves): mp = 2048 IS the transform length, so the wave's DFT is one plain 2048-point transform -- no chirp,
// no filter, no second pass, every bin used; 32 rows per wave instead of 16, G slots of 2048 entries, the window rows fetched with the samples
// (no LDS left for them), W_N2^k0 as a per-lane times a per-row factor
template <int W, bool EX = false> struct WideCfg {
    static constexpr int T = 4, R = 8 * T, M = 64 * R;                   // the sub-transform: L = 2048
    static constexpr int kWaves = 8, kGroups = kWaves / W;               // per workgroup
    static constexpr int kRows = EX ? 32 : 16;                           // rows of 64 points a wave fills: mp <= 1024 (EX: mp = 2048)
    static constexpr int kRowsD = EX ? 33 : 17;                          // rows of output bins per wave: (N2 + 1) / 64 / W, rounded up
    static constexpr bool kPrefetch = !EX;                               // EX: no registers for the next frame's samples next to the transform
    static constexpr int kSlabW = EX ? 2048 : 2 * 8 * kS1;               // float2 per wave: the transform's exchange slab (2 * 8 * kS1); then G_w[k0] (mp entries)
    static constexpr int kRegion = W * kSlabW + 8;                       // per frame group; after the G exchange it holds Z[0..N2]
    // LDS tables, float2 units -- the device table has exactly this layout (EX: then the window rows [W][2048])
    static constexpr int kWin = 0;                                       // [W][1024] (w[2j], w[2j+1]), j = W a + w; zero for a >= mp           (EX: not in LDS)
    static constexpr int kChirp = kWin + (EX ? 0 : W * 1024);            // [1024] c[a] = exp(-i pi a^2 / mp)                                     (EX: none)
    static constexpr int kFilt = kChirp + (EX ? 0 : 1024);               // [16][64][2]: FFT_M(b) / M, rows 2i, 2i+1 of a lane side by side       (EX: none)
    static constexpr int kTw1 = kFilt + (EX ? 0 : M);                    // [16][64][2] (31 rows, padded)
    static constexpr int kTw2 = kTw1 + R * 64;                           // [7][64]
    static constexpr int kCtw = kTw2 + 7 * 64;                           // [1024] exp(-2 pi i k0 / N2)   (EX: [64] exp(-2 pi i lane / N2), then [32] exp(-2 pi i 64 c / N2), padded)
    static constexpr int kSrow = kCtw + (EX ? 128 : 1024);               // [64] exp(-2 pi i lane / n), then [.] exp(-2 pi i 64 rho / n), rho <= 67
    static constexpr int kTabs = kSrow + 192;
    static constexpr int kWinDev = kTabs;                                // EX, device table only: [W][2048] (w[2j], w[2j+1]), j = W a + w
    static constexpr int kMisc = 8;                                      // float2: partial sums [kGroups][W] and band partials [kGroups][W] (floats)
    static constexpr size_t kLdsBytes = (static_cast<size_t>(kTabs) + kGroups * kRegion + kMisc) * sizeof(float2);
};
static_assert(WideCfg<4>::kLdsBytes <= 160 * 1024 && WideCfg<2>::kLdsBytes <= 160 * 1024 && WideCfg<2, true>::kLdsBytes <= 160 * 1024, "LDS of a CU");

struct WideParams {
    const float* x;
    int64_t clip_stride;
    int n_frames, hop;
    int64_t total_frames;
    int n_groups, iters;       // frame groups of the launch (<= total_frames), frames per group rounded up
    float* out;
    int64_t out_clip_stride;
    int n2, mp;                // nperseg / 2, n2 / W
    int aligned;               // every frame starts on an 8-byte boundary: one 8-byte load per point, else two 4-byte loads
    const float2* tab;         // [kTabs] (EX: + [W][2048] window rows)
    float scale;
    int k_lo, k_hi;            // MODE 2: bins of the band
};

// MODE: 0 psd, 1 magnitude, 2 band power (A11)
template <int W, bool DETREND, int MODE, bool EX>
__global__ __launch_bounds__((64 * WideCfg<W, EX>::kWaves), 2) void stft_rbluew_kernel(const WideParams p) {
    using C = WideCfg<W, EX>;
    constexpr int T = C::T, R = C::R, kWaves = C::kWaves;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave / W, w = wave % W;                  // frame group of the workgroup, this wave's residue class / spectrum part
    float2* const region = lds + C::kTabs + grp * C::kRegion;
    float2* const slab = region + w * C::kSlabW;
    float* const psum = reinterpret_cast<float*>(lds + C::kTabs + C::kGroups * C::kRegion) + grp * W;
    float* const bpart = psum + kWaves;

    // ---- tables -> LDS (once per workgroup); sqrt of the PSD scale rides on the window ---------------------------------------------------
    const float sq = sqrtf(MODE != 1 ? p.scale * 0.5f : p.scale * 0.25f);      // bins 0 and N2 get 1/2 below
    for (int i = threadIdx.x; i < C::kTabs; i += 64 * kWaves) {
        float2 v = p.tab[i];
        if (i < C::kChirp) { v.x *= sq; v.y *= sq; }
        lds[i] = v;
    }
    __syncthreads();

    const int n2 = p.n2, mp = p.mp;
    const float n_f = static_cast<float>(2 * n2);
    const CfftLds fl = cfft_lds(lds + C::kTw1, lds + C::kTw2, slab, lane);
    const float2* const win = lds + C::kWin + w * 1024 + lane;               // + 64 a
    const float2* const chirp = lds + C::kChirp + lane;                      // + 64 a
    const float2 lane_tw = lds[C::kSrow + lane];
    // W_W^(w' r), r = this wave's part of the spectrum: exact 0 / +-1 coefficients
    float bw_re[W], bw_im[W];
#pragma unroll
    for (int v = 0; v < W; ++v) {
        const int q = ((v * w) % W) * (4 / W);                               // quarter turns clockwise
        bw_re[v] = q == 0 ? 1.f : q == 2 ? -1.f : 0.f;
        bw_im[v] = q == 3 ? 1.f : q == 1 ? -1.f : 0.f;
    }

    const int lg = xcd_remap(blockIdx.x, gridDim.x) * C::kGroups + grp;
    int64_t g = lg < p.n_groups ? p.total_frames * lg / p.n_groups : 0;
    const int64_t g_end = lg < p.n_groups ? p.total_frames * (lg + 1) / p.n_groups : 0;
    int clip = static_cast<int>(g / p.n_frames);
    int f = static_cast<int>(g - static_cast<int64_t>(clip) * p.n_frames);

    auto load_frame = [&](int cl, int fr, float2 (&dst)[C::kRows]) {
        const float* const src = p.x + static_cast<int64_t>(cl) * p.clip_stride + static_cast<int64_t>(fr) * p.hop + 2 * w + 2 * W * lane;
        if (p.aligned) {                                     // wave-uniform
#pragma unroll
            for (int a = 0; a < C::kRows; ++a)
                dst[a] = lane + 64 * a < mp ? *reinterpret_cast<const float2*>(src + 128 * W * a) : make_float2(0.f, 0.f);
        } else {
#pragma unroll
            for (int a = 0; a < C::kRows; ++a)
                dst[a] = lane + 64 * a < mp ? make_float2(src[128 * W * a], src[128 * W * a + 1]) : make_float2(0.f, 0.f);
        }
    };
    auto part_sum = [&](const float2 (&v)[C::kRows]) {
        float s = v[0].x + v[0].y;
#pragma unroll
        for (int a = 1; a < C::kRows; ++a) s += v[a].x + v[a].y;
        return wave_sum(s);
    };
    auto group_mean = [&]() {
        float s = psum[0];
#pragma unroll
        for (int v = 1; v < W; ++v) s += psum[v];
        return s / n_f;                                      // a true division: a constant clip must detrend to 0 exactly, as in scipy
    };

    float2 winr[EX ? C::kRows : 1];                          // EX: this wave's window rows, fetched with every frame's samples (L2 hits)
    auto load_window = [&]() {
#pragma unroll
        for (int a = 0; a < (EX ? C::kRows : 0); ++a) {
            const float2 v = p.tab[C::kWinDev + w * 2048 + lane + 64 * a];
            winr[a] = make_float2(v.x * sq, v.y * sq);
        }
    };
    float2 nxt[C::kRows];
    if (C::kPrefetch) load_frame(clip, f, nxt);              // (a group without frames reads frame 0 of clip 0 and stores nothing)
    float mean = 0.f;
    if (DETREND && C::kPrefetch) {
        const float s = part_sum(nxt);
        if (lane == 0) psum[w] = s;
        __syncthreads();
        mean = group_mean();
        __syncthreads();                                     // the loop's first partial sums land after every wave has read these
    }

    for (int it = 0; it < p.iters; ++it, ++g) {
        const bool active = g < g_end;                       // uniform over the frame group
        float* const orow = p.out + static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(f) * (MODE == 2 ? 1 : n2 + 1);
        const bool more = g + 1 < g_end;
        const int clip_n = !more ? clip : f + 1 == p.n_frames ? clip + 1 : clip, f_n = !more ? f : f + 1 == p.n_frames ? 0 : f + 1;
        if (!C::kPrefetch) {                                 // samples and window rows fetched here; the mean costs a fifth barrier
            load_window();
            load_frame(clip, f, nxt);
            if (DETREND) {
                const float s = part_sum(nxt);
                if (lane == 0) psum[w] = s;
                __syncthreads();
                mean = group_mean();
            }
        }
        float2 d[T][8], e[T][8];
        // ---- a[m] = (x[2j] w[2j] + i x[2j+1] w[2j+1]) * c[a], j = W a + w; rows beyond mp are zero (their window entries are) ----------------
#pragma unroll
        for (int a = 0; a < R; ++a) {
            if (a < C::kRows) {
                const float2 wn = EX ? winr[EX ? a : 0] : lds_get(win + 64 * a);
                const float x0 = (nxt[a].x - mean) * wn.x, x1 = (nxt[a].y - mean) * wn.y;
                if (EX) {
                    d[a % T][a / T] = make_float2(x0, x1);
                } else {
                    const float2 c = lds_get(chirp + 64 * a);
                    d[a % T][a / T] = make_float2(fmaf(x0, c.x, -x1 * c.y), fmaf(x0, c.y, x1 * c.x));
                }
            } else {
                d[a % T][a / T] = make_float2(0.f, 0.f);
            }
        }
        if (C::kPrefetch) load_frame(clip_n, f_n, nxt);      // the group's last frame fetches itself again
        cfft_wave<T>(d, e, fl);
        // ---- Y = conj(A * B) -----------------------------------------------------------------------------------------------------------
        if (!EX) {
#pragma unroll
        for (int c = 0; c < R; c += 2) {
            const v4f fb = lds_get2(lds + C::kFilt + (c >> 1) * 128 + 2 * lane);
            const float2 y0 = cmul(e[c % T][c / T], make_float2(fb.x, fb.y));
            const float2 y1 = cmul(e[(c + 1) % T][(c + 1) / T], make_float2(fb.z, fb.w));
            d[c % T][c / T] = make_float2(y0.x, -y0.y);
            d[(c + 1) % T][(c + 1) / T] = make_float2(y1.x, -y1.y);
        }
        cfft_wave<T>(d, e, fl);                              // e = V; the convolution is conj(V) (1 / L is in B)
        }
        // ---- G_w[k0] = W_N2^(w k0) * c[k0] * conj(V[k0]) -> this wave's slab ---------------------------------------------------------------
        if (EX) {                                            // F_w[k0] is the transform itself
#pragma unroll
            for (int c = 0; c < C::kRows; ++c) {
                float2 z = e[c % T][c / T];
                if (w != 0) {                                // wave-uniform
                    const float2 t = cmul(lds_get(lds + C::kCtw + lane), lds_get(lds + C::kCtw + 64 + c));
                    float2 pw = t;
                    if (w >= 2) pw = cmul(t, t);
                    if (w == 3) pw = cmul(pw, t);
                    z = cmul(z, pw);
                }
                lds_put(slab + lane + 64 * c, z);
            }
        } else {
#pragma unroll
        for (int c = 0; c < C::kRows; ++c) {
            const float2 ch = lds_get(chirp + 64 * c);
            const float2 v = e[c % T][c / T];
            float2 z = make_float2(fmaf(ch.x, v.x, ch.y * v.y), fmaf(ch.y, v.x, -ch.x * v.y));      // (ch.x + i ch.y) * (v.x - i v.y)
            if (w != 0) {                                    // wave-uniform
                const float2 t = lds_get(lds + C::kCtw + lane + 64 * c);
                float2 pw = t;
                if (w >= 2) pw = cmul(t, t);
                if (w == 3) pw = cmul(pw, t);
                z = cmul(z, pw);
            }
            lds_put(slab + lane + 64 * c, z);
        }
        }
        if (DETREND && C::kPrefetch) {                       // the next frame's samples have arrived by now
            const float s = part_sum(nxt);
            if (lane == 0) psum[w] = s;
        }
        __syncthreads();                                     // (1) every G_w and partial sum of the workgroup is in LDS
        if (DETREND && C::kPrefetch) mean = group_mean();
        // ---- Z[k0 + w mp] = sum_v W_W^(v w) G_v[k0] ------------------------------------------------------------------------------------------
        float2 z[C::kRows];
#pragma unroll
        for (int c = 0; c < C::kRows; ++c) {
            float2 acc = lds_get(region + lane + 64 * c);                        // v = 0: coefficient 1
#pragma unroll
            for (int v = 1; v < W; ++v) {
                const float2 gv = lds_get(region + v * C::kSlabW + lane + 64 * c);
                acc.x += gv.x * bw_re[v] - gv.y * bw_im[v];
                acc.y += gv.x * bw_im[v] + gv.y * bw_re[v];
            }
            z[c] = acc;
        }
        __syncthreads();                                     // (2) every wave has taken its G values: the region becomes Z[0..N2]
#pragma unroll
        for (int c = 0; c < C::kRows; ++c)
            if (lane + 64 * c < mp) lds_put(region + w * mp + lane + 64 * c, z[c]);
        if (w == 0 && lane == 0) lds_put(region + n2, z[0]);                     // Z[N2] := Z[0]
        __syncthreads();                                     // (3)
        // ---- split + epilogue: this wave's rows of the bins k = 0..N2 ----------------------------------------------------------------------
        // (row addresses are rebuilt per frame from opaque copies of w and N2: hoisted out of the loop they are three registers per row
        //  held across the transforms)
        float bsum = 0.f;
        int w_d = w, n2_d = n2;
        asm volatile("" : "+s"(w_d), "+s"(n2_d));
#pragma unroll
        for (int cc = 0; cc < C::kRowsD; ++cc) {
            const int rho = w_d + W * cc;
            if (64 * rho <= n2) {                            // wave-uniform
                const int k = lane + 64 * rho;
                const int kk = k <= n2_d ? k : n2_d;         // lanes beyond the last bin read a valid entry and store nothing
                const float2 A = lds_get(region + kk);
                const float2 B = lds_get(region + (n2_d - kk));
                const float2 tw = cmul(lane_tw, lds_get(lds + C::kSrow + 64 + rho));
                const float2 S = make_float2(A.x + B.x, A.y - B.y);
                const float2 D = make_float2(A.x - B.x, A.y + B.y);
                const float2 X = make_float2(S.x + fmaf(tw.x, D.y, tw.y * D.x), S.y + fmaf(tw.y, D.y, -tw.x * D.x));
                float pk = fmaf(X.x, X.x, X.y * X.y);
                if (MODE != 1 && (k == 0 || k == n2)) pk *= 0.5f;
                if (MODE == 1) pk = sqrtf(pk);
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
            float s = bpart[0];
#pragma unroll
            for (int v = 1; v < W; ++v) s += bpart[v];
            orow[0] = s;
        }
        clip = clip_n;
        f = f_n;
    }
}

template <int W, bool DETREND, bool EX>
int launch_wd(const WideParams& prm, hipStream_t s, int mode, bool band, int n_cu) {
    using C = WideCfg<W, EX>;
    auto k0 = stft_rbluew_kernel<W, DETREND, 0, EX>;
    auto k1 = stft_rbluew_kernel<W, DETREND, 1, EX>;
    auto k2 = stft_rbluew_kernel<W, DETREND, 2, EX>;
    auto kern = band ? k2 : mode == SG_MODE_PSD ? k0 : k1;
    WideParams p = prm;
    int64_t n_groups = static_cast<int64_t>(n_cu) * C::kGroups;                 // one workgroup per CU (its tables fill the LDS)
    if (n_groups > p.total_frames) n_groups = p.total_frames;
    p.n_groups = static_cast<int>(n_groups);
    p.iters = static_cast<int>((p.total_frames + n_groups - 1) / n_groups);
    const int n_wg = static_cast<int>((n_groups + C::kGroups - 1) / C::kGroups);
    SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(C::kLdsBytes)));
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * C::kWaves), C::kLdsBytes, s, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft_rbluew launch");
}

template <int W, bool EX>
int launch_w(const sg_plan& p, const StftArgs& a) {
    WideParams prm{};
    prm.x = static_cast<const float*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.total_frames = a.n_frames * a.n_clips;
    prm.out = static_cast<float*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.n2 = p.nfft / 2;
    prm.mp = p.nfft / 2 / W;
    prm.aligned = (p.hop % 2 == 0) && (a.clip_stride % 2 == 0 || a.n_clips == 1) && (reinterpret_cast<uintptr_t>(a.x) % 8 == 0);
    prm.tab = static_cast<const float2*>(p.rb_wc_dev);
    prm.scale = static_cast<float>(p.scale);
    prm.k_lo = a.k_lo; prm.k_hi = a.k_hi;
    const bool band = a.band_mode != 0;                    // run_stft has checked: psd plan, 0 <= k_lo <= k_hi < n_bins
    return p.detrend == SG_DETREND_CONSTANT ? launch_wd<W, true, EX>(prm, a.stream, p.mode, band, p.n_cu)
                                            : launch_wd<W, false, EX>(prm, a.stream, p.mode, band, p.n_cu);
}

template <int W, bool EX>
void fill_tables(std::vector<float2>& tab, const std::vector<double>& window, int n) {
    using C = WideCfg<W, EX>;
    constexpr int R = C::R, M = C::M;
    const int n2 = n / 2, mp = n2 / W;
    const long double pi = 3.14159265358979323846264338327950288L;
    auto unit = [&](long double ang) { return make_float2(static_cast<float>(cosl(ang)), static_cast<float>(sinl(ang))); };
    tab.assign(C::kTabs + (EX ? W * 2048 : 0), make_float2(0.f, 0.f));
    for (int w = 0; w < W; ++w)
        for (int a = 0; a < mp; ++a) {
            const int j = W * a + w;
            tab[(EX ? C::kWinDev + w * 2048 : C::kWin + w * 1024) + a] = make_float2(static_cast<float>(window[2 * j]), static_cast<float>(window[2 * j + 1]));
        }
    if (!EX) {
        std::vector<double> br(mp), bi(mp);                  // b[j] = exp(+i pi j^2 / mp); j^2 mod 2 mp keeps the angle small
        for (int j = 0; j < mp; ++j) {
            const long long q = (static_cast<long long>(j) * j) % (2LL * mp);
            const long double ang = pi * static_cast<long double>(q) / static_cast<long double>(mp);
            br[j] = static_cast<double>(cosl(ang));
            bi[j] = static_cast<double>(sinl(ang));
            tab[C::kChirp + j] = make_float2(static_cast<float>(br[j]), static_cast<float>(-bi[j]));      // c[j] = conj b[j]
        }
        std::vector<double> hr(M, 0.0), hi(M, 0.0);
        hr[0] = br[0]; hi[0] = bi[0];
        for (int j = 1; j < mp; ++j) { hr[j] = hr[M - j] = br[j]; hi[j] = hi[M - j] = bi[j]; }
        host_fft_pow2(hr, hi);
        for (int k = 0; k < M; ++k) {
            const int r = k >> 6, l = k & 63;
            tab[C::kFilt + (r >> 1) * 128 + 2 * l + (r & 1)] = make_float2(static_cast<float>(hr[k] / M), static_cast<float>(hi[k] / M));
        }
        for (int k0 = 0; k0 < mp; ++k0) tab[C::kCtw + k0] = unit(-2.0L * pi * static_cast<long double>(k0) / static_cast<long double>(n2));
    } else {                                                 // W_N2^k0, k0 = lane + 64 c, as a per-lane times a per-row factor
        for (int l = 0; l < 64; ++l) tab[C::kCtw + l] = unit(-2.0L * pi * static_cast<long double>(l) / static_cast<long double>(n2));
        for (int c = 0; c < C::kRows; ++c) tab[C::kCtw + 64 + c] = unit(-2.0L * pi * static_cast<long double>(64 * c) / static_cast<long double>(n2));
    }
    for (int l = 0; l < 64; ++l) {
        for (int r = 1; r < R; ++r)
            tab[C::kTw1 + ((r - 1) >> 1) * 128 + 2 * l + ((r - 1) & 1)] = unit(-2.0L * pi * static_cast<long double>((static_cast<long long>(l) * r) % M) / M);
        for (int s = 1; s < 8; ++s) tab[C::kTw2 + (s - 1) * 64 + l] = unit(-2.0L * pi * static_cast<long double>(((l & 7) * s) % 64) / 64.0L);
        tab[C::kSrow + l] = unit(-2.0L * pi * static_cast<long double>(l) / static_cast<long double>(n));
    }
    for (int rho = 0; 64 * rho <= n2; ++rho) tab[C::kSrow + 64 + rho] = unit(-2.0L * pi * static_cast<long double>(64 * rho) / static_cast<long double>(n));
}

}  // namespace

// wavefronts per frame of a plan rbluew_ok() accepts (spectro_api.hip)
// (8192 itself: two waves, each ONE 2048-point transform, EX)
int rbluew_size(int nfft) { return nfft <= 4096 || nfft == 8192 ? 2 : 4; }

// (odd hops and clips at odd strides run here too, with 4-byte loads; int16 input is converted first, spectro_api.hip)
bool rbluew_can_run(const sg_plan&, const StftArgs& a) {
    return !a.in_i16 && (reinterpret_cast<uintptr_t>(a.x) % 4 == 0) && a.n_frames <= INT32_MAX;
}

int launch_rbluew(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    if (p.nfft == 8192) return launch_w<2, true>(p, a);
    return rbluew_size(p.nfft) == 2 ? launch_w<2, false>(p, a) : launch_w<4, false>(p, a);
}

// One device table in the kernel's LDS order (computed in double): window pairs per wave, chirp, filter spectrum, FFT twiddles,
// W_N2^k0, split-twiddle factors.
int build_rbluew_tables(sg_plan& p, const std::vector<double>& window) {
    std::vector<float2> tab;
    if (p.nfft == 8192) fill_tables<2, true>(tab, window, p.nfft);
    else if (rbluew_size(p.nfft) == 2) fill_tables<2, false>(tab, window, p.nfft); else fill_tables<4, false>(tab, window, p.nfft);
    SG_HIP(hipMalloc(&p.rb_wc_dev, tab.size() * sizeof(float2)));
    SG_HIP(hipMemcpy(p.rb_wc_dev, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
    return SG_OK;
}

}  // namespace sg
