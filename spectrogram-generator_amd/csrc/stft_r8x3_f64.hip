// stft_r8x3_f64.hip -- the register kernels in double precision: nperseg = nfft = 1024 ("r8x3d"), 512, 256 and (round 4) 128 ("rsmalld").
//
// Why they exist: the reference's default nperseg is 1024 (GUI.py:214, spin box 32...8192) and its recordings arrive as float64
// (neo magnitudes, SweepManager.py:135-136), and scipy computes in the input's precision (scipy/signal/_spectral_py.py:1976-1981)
// -- a batch of f64 sweeps at the reference's own settings is exactly these plans.  Until round 2 they ran on the LDS Stockham
// kernel (stft_stockham.hip: a workgroup per frame, a barrier per pass, 0.10 G frames/s at 1024).
//
// Same machine mapping as stft_r8x3.hip / stft_rsmall.hip (read those files for the index maps): one wavefront carries G = 8/R
// frames (R = nfft/128), 8 complex values per lane, pass 1 = G R-point DFTs, passes 2/3 = register radix-8, two padded LDS
// transposes (strides 72 / 66 elements), a split pass in which only the upper half crosses lanes, no s_barrier in the loop.
// Differences from the f32 kernels:
//   * a complex double is 16 bytes: the slab is kept as separate real and imaginary planes of 8-byte elements, so every exchange
//     is two ds_write_b64 / ds_read_b64 with the index maps of the f32 kernels -- conflict-free by the same argument
//     (tools/sim_r8x3.py, sim_rsmall.py) instead of a new analysis for 16-byte accesses;
//   * 8 values + 8 prefetched + window + twiddle doubles per lane: 208-233 VGPRs at R = 8, two waves per SIMD;
//   * no register sliding window (every group reloads its samples from L1 / L2; the next group's loads are issued before this
//     group's FFT).  A sliding variant (hops 128 / 256 / 512 and interleaved 64 / 32 / 16, as in stft_r8x3.hip) was built and measured
//     in round 2: 202 / 864 / 443 us against 201 / 862 / 434 us per 64-clip batch at hops 256 / 64 / 128 -- the f64 arithmetic, not the
//     re-read samples, is what this kernel's joules go to; not kept; of the fused products of the f32 kernel only the band power (A11: the HMM feature path on f64 recordings,
//     `sg_stft_band_power`) is replicated here.
//   * the rows of a group are stored straight from the split pass (G segments of 8L = 128 / 256 bytes per instruction at R = 2 / 4): sending
//     them through the slab to leave as one contiguous run, what gave stft_rsmall.hip 13 % in round 3 (its segments are 64 bytes), costs
//     this kernel 4-6 % (nfft 256 hop 64: 216 vs 208 us, 512 hop 128: 209 vs 197 us per 64-clip batch; same box, two rounds); not kept.
// Algorithmic HBM bytes per frame: hop*8 + (nfft/2+1)*8 (band power: hop*8 + 8).
#include "spectro_internal.h"

#include <cmath>
#include <cstdlib>

namespace sg {
namespace {

constexpr int kS1 = 72, kS2 = 66, kSlab = 8 * kS1;       // 576 elements per plane
constexpr int kWaves = 4;                                // per workgroup
#ifndef SG_R8D_TW_LDS
#define SG_R8D_TW_LDS 0        // 1: the t1 / t2 twiddles in a workgroup-shared LDS table instead of up to 56 VGPRs -> 142-152 VGPRs, three waves per
                               // SIMD.  Measured (profiles/r03_f64_tw_lds.txt): nfft 1024 192 -> 210 us (the extra LDS reads cost more than the third
                               // wave hides), 512 equal, 256 204 -> 200 us.  Off.
#endif
constexpr int kOcc = SG_R8D_TW_LDS ? 3 : 2;             // waves per SIMD
#ifndef SG_R8D_PRIO
// wave priority rises along a group (pass 1 -> stores), as in stft_r8x3.hip -- for the kernels that carry several frames per
// wave only: same-box A/B per 64-clip batch, nfft 512 hop 128: 192 us with, 200-203 us without; nfft 1024 hop 256: 195-196 us
// with, 190-192 us without (so R = 8 runs without)
#define SG_R8D_PRIO 1
#endif

struct cd { double x, y; };
__device__ __forceinline__ cd cadd(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd csub(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd w) { return {fma(a.x, w.x, -a.y * w.y), fma(a.x, w.y, a.y * w.x)}; }
__device__ __forceinline__ cd mul_mi(cd a) { return {a.y, -a.x}; }

// forward 8-point DFT in registers, natural order in and out (see fft_wave.h: the two 1/sqrt(2) rotations are folded into the
// FMAs of the last stage)
__device__ __forceinline__ void radix8(cd (&a)[8]) {
    constexpr double h = 0.70710678118654752440;
    const cd b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]);
    const cd b1 = cadd(a[1], a[5]), b5 = csub(a[1], a[5]);
    const cd b2 = cadd(a[2], a[6]), b6 = csub(a[2], a[6]);
    const cd b3 = cadd(a[3], a[7]), b7 = csub(a[3], a[7]);
    const cd t5 = {b5.x + b5.y, b5.y - b5.x};
    const cd t6 = mul_mi(b6);
    const cd t7 = {b7.y - b7.x, -(b7.x + b7.y)};
    const cd c0 = cadd(b0, b2), c2 = csub(b0, b2);
    const cd c1 = cadd(b1, b3), c3 = mul_mi(csub(b1, b3));
    const cd c4 = cadd(b4, t6), c6 = csub(b4, t6);
    const cd c5 = cadd(t5, t7), c7 = mul_mi(csub(t5, t7));
    a[0] = cadd(c0, c1); a[4] = csub(c0, c1);
    a[2] = cadd(c2, c3); a[6] = csub(c2, c3);
    a[1] = {fma(h, c5.x, c4.x), fma(h, c5.y, c4.y)};
    a[5] = {fma(-h, c5.x, c4.x), fma(-h, c5.y, c4.y)};
    a[3] = {fma(h, c7.x, c6.x), fma(h, c7.y, c6.y)};
    a[7] = {fma(-h, c7.x, c6.x), fma(-h, c7.y, c6.y)};
}

typedef __attribute__((address_space(3))) volatile double lds_f64;
struct Slab {                       // a wave's exchange area: real plane, imaginary plane
    double* re; double* im;
    __device__ __forceinline__ void put(int i, cd v) const { *(lds_f64*)(re + i) = v.x; *(lds_f64*)(im + i) = v.y; }
    __device__ __forceinline__ cd get(int i) const { return {*(lds_f64*)(re + i), *(lds_f64*)(im + i)}; }
};
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

struct R8DParams {
    const double* x;
    int64_t clip_stride;
    int n_frames, hop;
    int groups_per_clip;     // ceil(n_frames / G), G = 8/R frames per wave step
    int64_t total_groups;
    int n_waves;
    double* out;
    int64_t out_clip_stride;
    const double2* win2;     // [M]  (w[2n], w[2n+1]),  M = nfft/2 = 64R
    const double2* tw;       // [(R-1) + 7 + 4][64]: t1[r-1][j] = exp(-2 pi i j r/M), t2[s-1][j] = exp(-2 pi i (j&7) s/64),
                             //                      t3[t][j] = (cos, sin)(2 pi ((j % 8R) + 8R t)/(2M))
    double scale;
    int k_lo, k_hi;          // MODE 2: bins of the band
};

template <int R> __device__ __forceinline__ void radix_first(cd* a);
template <> __device__ __forceinline__ void radix_first<1>(cd*) {}
template <> __device__ __forceinline__ void radix_first<2>(cd* a) {
    const cd s = cadd(a[0], a[1]), d = csub(a[0], a[1]);
    a[0] = s; a[1] = d;
}
template <> __device__ __forceinline__ void radix_first<4>(cd* a) {
    const cd s02 = cadd(a[0], a[2]), d02 = csub(a[0], a[2]);
    const cd s13 = cadd(a[1], a[3]), d13 = mul_mi(csub(a[1], a[3]));
    a[0] = cadd(s02, s13); a[2] = csub(s02, s13);
    a[1] = cadd(d02, d13); a[3] = csub(d02, d13);
}
template <> __device__ __forceinline__ void radix_first<8>(cd* a) { radix8(*reinterpret_cast<cd(*)[8]>(a)); }

template <int L> __device__ __forceinline__ double group_sum(double v) {     // over the L = 8R lanes that share a frame
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// R = nfft/128: 1 (128: no pass 1), 2 (256), 4 (512), 8 (1024).  A wave carries G = 8/R frames per step (stft_rsmall.hip; G = 1 is the r8x3 mapping).
// MODE 0 psd, 1 magnitude, 2 band power: out[clip][frame] = sum of PSD bins [k_lo, k_hi] (A11)
template <int R, bool DETREND, int MODE>
__global__ __launch_bounds__(64 * kWaves, kOcc) void stft_reg_f64_kernel(const R8DParams p) {
    constexpr int G = 8 / R, M = 64 * R, L = 8 * R, NB = M + 1, RS = M + 8;
    constexpr bool kPrio = SG_R8D_PRIO && R < 8;
    static_assert(G * RS <= kSlab, "split regions must fit the slab");
    constexpr int kTwRows = SG_R8D_TW_LDS ? (R - 1 + 7) : 0;
    __shared__ __attribute__((aligned(16))) double lds[kWaves * 2 * kSlab + 2 * kTwRows * 64 + 2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Slab sl{lds + wave * 2 * kSlab, lds + wave * 2 * kSlab + kSlab};
    const Slab twl{lds + kWaves * 2 * kSlab, lds + kWaves * 2 * kSlab + kTwRows * 64};      // rows 0..R-2: t1, R-1..R+5: t2 (lane-linear)
    if (SG_R8D_TW_LDS) {
        for (int i = threadIdx.x; i < kTwRows * 64; i += 64 * kWaves) { const double2 v = p.tw[i]; twl.put(i, {v.x, v.y}); }
        __syncthreads();
    }

    const int lw = xcd_remap(blockIdx.x, gridDim.x) * kWaves + wave;
    if (lw >= p.n_waves) return;

    cd w[R], t3[4];
#if !SG_R8D_TW_LDS
    cd t1[R > 1 ? R - 1 : 1], t2[7];
#endif
    const double sq = sqrt(MODE != 1 ? p.scale * 0.5 : p.scale * 0.25);      // PSD scale rides on the window (stft_r8x3.hip)
#pragma unroll
    for (int a = 0; a < R; ++a) { const double2 v = p.win2[lane + 64 * a]; w[a] = {v.x * sq, v.y * sq}; }
#if SG_R8D_TW_LDS
#define SG_T1(r) twl.get((r) * 64 + lane)
#define SG_T2(s) twl.get((R - 1 + (s)) * 64 + lane)
#else
#pragma unroll
    for (int r = 0; r < R - 1; ++r) { const double2 v = p.tw[r * 64 + lane]; t1[r] = {v.x, v.y}; }
#pragma unroll
    for (int s = 0; s < 7; ++s) { const double2 v = p.tw[(R - 1 + s) * 64 + lane]; t2[s] = {v.x, v.y}; }
#define SG_T1(r) t1[r]
#define SG_T2(s) t2[s]
#endif
#pragma unroll
    for (int t = 0; t < 4; ++t) { const double2 v = p.tw[(R - 1 + 7 + t) * 64 + lane]; t3[t] = {v.x, v.y}; }

    const int j0 = lane & 7, hi = lane >> 3;
    const int x1w = hi * kS1 + j0, x1r = lane;                          // + 8 v   | + b kS1
    const int x2w = j0 * kS2 + (hi % R) + L * (hi / R), x2r = lane;     // + R s   | + j kS2     (hi = g R + r here)
    const int g3 = lane / L, lu = lane - g3 * L;
    const int x3w = g3 * RS + lu, x3b = g3 * RS + (M - lu);             // + L t   | - L t
    const double r0 = (MODE != 1 && lu == 0) ? 0.5 : 1.0;

    int64_t q = p.total_groups * lw / p.n_waves;
    const int64_t q_end = p.total_groups * (lw + 1) / p.n_waves;

    // the loads of group q+1 are issued before the FFT of group q
    auto load_group = [&](int clip, int gi, cd (&dst)[8]) {
        const double* const xclip = p.x + static_cast<int64_t>(clip) * p.clip_stride + 2 * lane;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = min(gi * G + g, p.n_frames - 1);               // partial last group: recompute the last frame
            const double* const src = xclip + static_cast<int64_t>(f) * p.hop;
#pragma unroll
            for (int k = 0; k < R; ++k) { const double2 v = *reinterpret_cast<const double2*>(src + 128 * k); dst[g * R + k] = {v.x, v.y}; }
        }
    };
    int clip = static_cast<int>(q / p.groups_per_clip);
    int gi = static_cast<int>(q - static_cast<int64_t>(clip) * p.groups_per_clip);
    cd nxt[8];
    if (q < q_end) load_group(clip, gi, nxt);

    for (; q < q_end; ++q) {
        const int clip_n = gi + 1 == p.groups_per_clip ? clip + 1 : clip, gi_n = gi + 1 == p.groups_per_clip ? 0 : gi + 1;
        const bool more = q + 1 < q_end;                                 // the run's last group fetches itself again (an
        cd a[8];                                                         // unconditional load: no wait parked behind it)
#pragma unroll
        for (int v = 0; v < 8; ++v) a[v] = nxt[v];
        load_group(more ? clip_n : clip, more ? gi_n : gi, nxt);
        if (kPrio) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (DETREND) {                                               // A3 (scipy:2191, detrend 'constant')
                double s = a[g * R].x + a[g * R].y;
#pragma unroll
                for (int k = 1; k < R; ++k) s += a[g * R + k].x + a[g * R + k].y;
                const double mean = wave_sum(s) * (1.0 / (2 * M));
#pragma unroll
                for (int k = 0; k < R; ++k) { a[g * R + k].x -= mean; a[g * R + k].y -= mean; }
            }
#pragma unroll
            for (int k = 0; k < R; ++k) { a[g * R + k].x *= w[k].x; a[g * R + k].y *= w[k].y; }       // A4
            // ---- pass 1: G independent R-point DFTs ----
            radix_first<R>(a + g * R);
#pragma unroll
            for (int r = 1; r < R; ++r) a[g * R + r] = cmul(a[g * R + r], SG_T1(r - 1));
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) sl.put(x1w + 8 * v, a[v]);
        wave_lds_fence();
#pragma unroll
        for (int b = 0; b < 8; ++b) a[b] = sl.get(x1r + b * kS1);
        wave_lds_fence();
        // ---- pass 2 ----
        if (kPrio) __builtin_amdgcn_s_setprio(1);
        radix8(a);
#pragma unroll
        for (int s = 1; s < 8; ++s) a[s] = cmul(a[s], SG_T2(s - 1));
#pragma unroll
        for (int s = 0; s < 8; ++s) sl.put(x2w + R * s, a[s]);
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = sl.get(x2r + j * kS2);
        wave_lds_fence();
        // ---- pass 3: lane lu of group g3 holds Z_g[lu + L t] ----
        if (kPrio) __builtin_amdgcn_s_setprio(2);
        radix8(a);
#pragma unroll
        for (int t = 4; t < 8; ++t) sl.put(x3w + L * t, a[t]);
        if (lu == 0) sl.put(g3 * RS + M, a[0]);                          // Z_g[M] := Z_g[0]
        wave_lds_fence();
        // ---- split pass + |X|^2 (A5 tail, A6) ----
        if (kPrio) __builtin_amdgcn_s_setprio(3);
        const int f = gi * G + g3;
        const bool live = f < p.n_frames;
        double* const orow = p.out + static_cast<int64_t>(clip) * p.out_clip_stride +
                             static_cast<int64_t>(min(f, p.n_frames - 1)) * (MODE == 2 ? 1 : NB);
        double bsum = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const cd A = a[t];
            const cd B = sl.get(x3b - L * t);
            const cd cs = t3[t];
            const cd S = {A.x + B.x, A.y - B.y};
            const cd D = {A.x - B.x, A.y + B.y};
            const cd T = {fma(cs.y, D.x, -cs.x * D.y), fma(cs.x, D.x, cs.y * D.y)};
            const cd Xk = csub(S, T), Xm = cadd(S, T);
            double pk = fma(Xk.x, Xk.x, Xk.y * Xk.y), pm = fma(Xm.x, Xm.x, Xm.y * Xm.y);
            if (MODE != 1 && t == 0) { pk *= r0; pm *= r0; }
            if (MODE == 1) { pk = sqrt(pk); pm = sqrt(pm); }
            const int k = lu + L * t;
            if (MODE == 2) {
                if (k >= p.k_lo && k <= p.k_hi) bsum += pk;
                if (M - k >= p.k_lo && M - k <= p.k_hi) bsum += pm;
            } else if (live) {
                orow[k] = pk;
                orow[M - k] = pm;
            }
        }
        {
            double pq = fma(a[4].x, a[4].x, a[4].y * a[4].y) * 4.0;      // k = M/2 pairs with itself: lane lu == 0 holds Z[M/2]
            if (MODE == 1) pq = sqrt(pq);
            if (MODE == 2) {
                if (lu == 0 && M / 2 >= p.k_lo && M / 2 <= p.k_hi) bsum += pq;
                bsum = group_sum<L>(bsum);
                if (live && lu == 0) orow[0] = bsum;
            } else if (live && lu == 0) {
                orow[M / 2] = pq;
            }
        }
        wave_lds_fence();
        clip = clip_n;
        gi = gi_n;
    }
}

template <int R, bool DETREND>
int launch_mode(const R8DParams& prm, int n_wg, hipStream_t s, int mode, bool band) {
    if (band) hipLaunchKernelGGL((stft_reg_f64_kernel<R, DETREND, 2>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    else if (mode == SG_MODE_PSD) hipLaunchKernelGGL((stft_reg_f64_kernel<R, DETREND, 0>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    else hipLaunchKernelGGL((stft_reg_f64_kernel<R, DETREND, 1>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft_reg_f64 launch");
}

template <int R>
int launch_r(const sg_plan& p, const StftArgs& a) {
    constexpr int G = 8 / R;
    R8DParams prm{};
    prm.x = static_cast<const double*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.groups_per_clip = static_cast<int>((a.n_frames + G - 1) / G);
    prm.total_groups = static_cast<int64_t>(prm.groups_per_clip) * a.n_clips;
    int64_t n_waves = static_cast<int64_t>(p.n_cu) * 4 * kOcc;
    if (n_waves > prm.total_groups) n_waves = prm.total_groups;      // GUI-sized calls: one group per wave, latency before efficiency
    prm.n_waves = static_cast<int>(n_waves);
    prm.out = static_cast<double*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.win2 = static_cast<const double2*>(p.win_dev);
    prm.tw = static_cast<const double2*>(p.r8_tw_dev);
    prm.scale = p.scale;
    prm.k_lo = a.k_lo; prm.k_hi = a.k_hi;
    const int n_wg = static_cast<int>((n_waves + kWaves - 1) / kWaves);
    const bool band = a.band_mode != 0;                     // run_stft has checked: psd plan, 0 <= k_lo <= k_hi < n_bins
    return p.detrend == SG_DETREND_CONSTANT ? launch_mode<R, true>(prm, n_wg, a.stream, p.mode, band)
                                            : launch_mode<R, false>(prm, n_wg, a.stream, p.mode, band);
}

}  // namespace

bool r8x3_f64_can_run(const sg_plan& p, const StftArgs& a) {
    return p.dtype == SG_F64 && !a.in_i16 && !a.db_mode && a.mel_ipl == 0 && (p.hop % 2 == 0) &&
           (a.clip_stride % 2 == 0 || a.n_clips == 1) && (reinterpret_cast<uintptr_t>(a.x) % 16 == 0) && a.n_frames <= INT32_MAX;
}

int launch_r8x3_f64(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    switch (p.nfft) {
        case 128: return launch_r<1>(p, a);
        case 256: return launch_r<2>(p, a);
        case 512: return launch_r<4>(p, a);
        default: return launch_r<8>(p, a);
    }
}

// the per-lane twiddle table [(R-1) + 7 + 4][64] of stft_rsmall.hip / stft_r8x3.hip (R = nfft/128) in double
int build_r8x3_f64_tables(sg_plan& p) {
    const int R = p.nfft / 128, M = 64 * R, L = 8 * R;
    std::vector<double> tw(static_cast<size_t>(R - 1 + 7 + 4) * 64 * 2);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    auto put = [&](int row, int j, long double ang) {
        tw[2 * (static_cast<size_t>(row) * 64 + j)] = static_cast<double>(cosl(ang));
        tw[2 * (static_cast<size_t>(row) * 64 + j) + 1] = static_cast<double>(sinl(ang));
    };
    for (int j = 0; j < 64; ++j) {
        for (int r = 1; r < R; ++r) put(r - 1, j, -two_pi * static_cast<long double>((j * r) % M) / M);
        for (int s = 1; s < 8; ++s) put(R - 1 + s - 1, j, -two_pi * static_cast<long double>(((j & 7) * s) % 64) / 64.0L);
        for (int t = 0; t < 4; ++t) put(R - 1 + 7 + t, j, two_pi * static_cast<long double>((j % L) + L * t) / (2.0L * M));
    }
    SG_HIP(hipMalloc(&p.r8_tw_dev, tw.size() * sizeof(double)));
    SG_HIP(hipMemcpy(p.r8_tw_dev, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    return SG_OK;
}

}  // namespace sg
