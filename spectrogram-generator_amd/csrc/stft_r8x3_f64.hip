// stft_r8x3_f64.hip -- the nperseg = nfft = 1024 register kernel in double precision.
//
// Why it exists: the reference's default nperseg is 1024 (GUI.py:214) and its recordings arrive as float64 (neo magnitudes,
// SweepManager.py:135-136), and scipy computes in the input's precision (scipy/signal/_spectral_py.py:1976-1981) -- a batch
// of f64 sweeps at the reference's own default is exactly this plan.  Until round 2 it ran on the LDS Stockham kernel
// (stft_stockham.hip: one 256-thread workgroup per frame, a barrier per pass, 0.10 G frames/s).
//
// Same machine mapping as stft_r8x3.hip (read that file for the index maps): one wavefront = one frame, 512 complex points of
// the even/odd packed signal as 8 complex values per lane, three register radix-8 passes, two padded LDS transposes (strides
// 72 / 66 elements), a split pass in which only the upper half crosses lanes, no s_barrier in the frame loop.  Differences:
//   * a complex double is 16 bytes: the slab is kept as separate real and imaginary planes of 8-byte elements, so every exchange
//     is two ds_write_b64 / ds_read_b64 with the index maps of the f32 kernel -- conflict-free by the same argument
//     (tools/sim_r8x3.py) instead of a new analysis for 16-byte accesses;
//   * 8 values + 8 prefetched + 16 window + 36 twiddle doubles per lane: ~230 VGPRs, two waves per SIMD;
//   * no register sliding window (every frame reloads its 1024 samples from L1 / L2; the next frame's loads are issued before
//     this frame's FFT); of the fused products of the f32 kernel only the band power (A11: the HMM feature path on f64
//     recordings, `sg_stft_band_power`) is replicated here.
// Algorithmic HBM bytes per frame: hop*8 + 513*8 (band power: hop*8 + 8).
#include "spectro_internal.h"

#include <cmath>
#include <cstdlib>

namespace sg {
namespace {

constexpr int kN = 1024, kM = 512, kBins = 513;
constexpr int kS1 = 72, kS2 = 66, kSlab = 8 * kS1;       // 576 elements per plane
constexpr int kWaves = 4;                                // per workgroup
constexpr int kOcc = 2;                                  // waves per SIMD

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
    int64_t total_frames;
    int n_waves;
    double* out;
    int64_t out_clip_stride;
    const double2* win2;     // [512]  (w[2n], w[2n+1])
    const double2* tw;       // [18][64]: t1[r-1][j] = exp(-2 pi i j r/512), t2[s-1][j] = exp(-2 pi i (j&7) s/64), t3[m][j] = (cos, sin)(2 pi (j+64m)/1024)
    double scale;
    int k_lo, k_hi;          // MODE 2: bins of the band
};

template <bool DETREND, int MODE>    // MODE 0 psd, 1 magnitude, 2 band power: out[clip][frame] = sum of PSD bins [k_lo, k_hi] (A11)
__global__ __launch_bounds__(64 * kWaves, kOcc) void stft1024_r8x3_f64_kernel(const R8DParams p) {
    __shared__ __attribute__((aligned(16))) double lds[kWaves * 2 * kSlab];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Slab sl{lds + wave * 2 * kSlab, lds + wave * 2 * kSlab + kSlab};

    const int lw = xcd_remap(blockIdx.x, gridDim.x) * kWaves + wave;
    if (lw >= p.n_waves) return;

    cd w[8], t1[7], t2[7], t3[4];
    const double sq = sqrt(MODE != 1 ? p.scale * 0.5 : p.scale * 0.25);      // PSD scale rides on the window (stft_r8x3.hip)
#pragma unroll
    for (int a = 0; a < 8; ++a) { const double2 v = p.win2[lane + 64 * a]; w[a] = {v.x * sq, v.y * sq}; }
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const double2 u = p.tw[r * 64 + lane], v = p.tw[(7 + r) * 64 + lane];
        t1[r] = {u.x, u.y}; t2[r] = {v.x, v.y};
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) { const double2 v = p.tw[(14 + m) * 64 + lane]; t3[m] = {v.x, v.y}; }
    const double r0 = (MODE != 1 && lane == 0) ? 0.5 : 1.0;

    const int j0 = lane & 7, hi = lane >> 3;
    const int x1w = hi * kS1 + j0, x1r = lane, x2w = j0 * kS2 + hi, x2r = lane, x3w = lane, x3b = kM - lane;

    int64_t g = p.total_frames * lw / p.n_waves;
    const int64_t g_end = p.total_frames * (lw + 1) / p.n_waves;
    while (g < g_end) {
        const int clip = static_cast<int>(g / p.n_frames);
        const int f0 = static_cast<int>(g - static_cast<int64_t>(clip) * p.n_frames);
        const int f1 = static_cast<int>(min(static_cast<int64_t>(p.n_frames), f0 + (g_end - g)));
        g += f1 - f0;
        const double* src = p.x + static_cast<int64_t>(clip) * p.clip_stride + 2 * lane + static_cast<int64_t>(f0) * p.hop;
        double* orow = p.out + static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(f0) * (MODE == 2 ? 1 : kBins);

        cd raw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double2 v = *reinterpret_cast<const double2*>(src + 128 * k); raw[k] = {v.x, v.y}; }
        for (int f = f0; f < f1; ++f) {
            cd a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = raw[k];
            {   // prefetch the next frame (the run's last frame fetches itself again: an unconditional load keeps the compiler
                // from parking a wait right behind it)
                const double* const nxt = f + 1 < f1 ? src + p.hop : src;
                src = nxt;
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 v = *reinterpret_cast<const double2*>(nxt + 128 * k); raw[k] = {v.x, v.y}; }
            }
            if (DETREND) {                                        // A3 (scipy:2191, detrend 'constant')
                double s = a[0].x + a[0].y;
#pragma unroll
                for (int k = 1; k < 8; ++k) s += a[k].x + a[k].y;
                const double mean = wave_sum(s) * (1.0 / kN);
#pragma unroll
                for (int k = 0; k < 8; ++k) { a[k].x -= mean; a[k].y -= mean; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { a[k].x *= w[k].x; a[k].y *= w[k].y; }       // A4

            // ---- pass 1 ----
            radix8(a);
#pragma unroll
            for (int r = 1; r < 8; ++r) a[r] = cmul(a[r], t1[r - 1]);
#pragma unroll
            for (int r = 0; r < 8; ++r) sl.put(x1w + 8 * r, a[r]);
            wave_lds_fence();
#pragma unroll
            for (int b = 0; b < 8; ++b) a[b] = sl.get(x1r + b * kS1);
            wave_lds_fence();
            // ---- pass 2 ----
            radix8(a);
#pragma unroll
            for (int s = 1; s < 8; ++s) a[s] = cmul(a[s], t2[s - 1]);
#pragma unroll
            for (int s = 0; s < 8; ++s) sl.put(x2w + 8 * s, a[s]);
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = sl.get(x2r + j * kS2);
            wave_lds_fence();
            // ---- pass 3: Z[lane + 64 t] ----
            radix8(a);
#pragma unroll
            for (int t = 4; t < 8; ++t) sl.put(x3w + 64 * t, a[t]);
            sl.put(kM + lane, a[0]);                              // lane 0: Z[512] := Z[0]
            wave_lds_fence();
            // ---- split pass + |X|^2 (A5 tail, A6) ----
            double bsum = 0.0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const cd A = a[m];
                const cd B = sl.get(x3b - 64 * m);
                const cd cs = t3[m];
                const cd S = {A.x + B.x, A.y - B.y};
                const cd D = {A.x - B.x, A.y + B.y};
                const cd T = {fma(cs.y, D.x, -cs.x * D.y), fma(cs.x, D.x, cs.y * D.y)};
                const cd Xk = csub(S, T), Xm = cadd(S, T);
                double pk = fma(Xk.x, Xk.x, Xk.y * Xk.y), pm = fma(Xm.x, Xm.x, Xm.y * Xm.y);
                if (MODE != 1 && m == 0) { pk *= r0; pm *= r0; }
                if (MODE == 1) { pk = sqrt(pk); pm = sqrt(pm); }
                const int k = lane + 64 * m;
                if (MODE == 2) {
                    if (k >= p.k_lo && k <= p.k_hi) bsum += pk;
                    if (kM - k >= p.k_lo && kM - k <= p.k_hi) bsum += pm;
                } else {
                    orow[k] = pk;
                    orow[kM - k] = pm;
                }
            }
            {
                const double zx = __shfl(a[4].x, 0), zy = __shfl(a[4].y, 0);     // k = 256 pairs with itself: lane 0's a[4]
                double pq = fma(zx, zx, zy * zy) * 4.0;
                if (MODE == 1) pq = sqrt(pq);
                if (MODE == 2) {
                    if (lane == 0 && 256 >= p.k_lo && 256 <= p.k_hi) bsum += pq;
                    bsum = wave_sum(bsum);
                    if (lane == 0) orow[0] = bsum;
                } else {
                    orow[256] = pq;
                }
            }
            orow += MODE == 2 ? 1 : kBins;
            wave_lds_fence();
        }
    }
}

template <bool DETREND>
int launch_mode(const R8DParams& prm, int n_wg, hipStream_t s, int mode, bool band) {
    if (band) hipLaunchKernelGGL((stft1024_r8x3_f64_kernel<DETREND, 2>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    else if (mode == SG_MODE_PSD) hipLaunchKernelGGL((stft1024_r8x3_f64_kernel<DETREND, 0>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    else hipLaunchKernelGGL((stft1024_r8x3_f64_kernel<DETREND, 1>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft1024_r8x3_f64 launch");
}

}  // namespace

bool r8x3_f64_can_run(const sg_plan& p, const StftArgs& a) {
    return p.dtype == SG_F64 && !a.in_i16 && !a.db_mode && a.mel_ipl == 0 && (p.hop % 2 == 0) &&
           (a.clip_stride % 2 == 0 || a.n_clips == 1) && (reinterpret_cast<uintptr_t>(a.x) % 16 == 0) && a.n_frames <= INT32_MAX;
}

int launch_r8x3_f64(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    R8DParams prm{};
    prm.x = static_cast<const double*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.total_frames = a.n_frames * a.n_clips;
    int64_t n_waves = static_cast<int64_t>(p.n_cu) * 4 * kOcc;
    const int64_t by_work = (prm.total_frames + 3) / 4;
    if (n_waves > by_work) n_waves = by_work;
    prm.n_waves = static_cast<int>(n_waves);
    prm.out = static_cast<double*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.win2 = static_cast<const double2*>(p.win_dev);
    prm.tw = static_cast<const double2*>(p.r8_tw_dev);
    prm.scale = p.scale;
    prm.k_lo = a.k_lo; prm.k_hi = a.k_hi;
    const int n_wg = static_cast<int>((n_waves + kWaves - 1) / kWaves);
    const bool band = a.band_mode != 0;                     // run_stft has checked: psd plan, 0 <= k_lo <= k_hi < 513
    return p.detrend == SG_DETREND_CONSTANT ? launch_mode<true>(prm, n_wg, a.stream, p.mode, band) : launch_mode<false>(prm, n_wg, a.stream, p.mode, band);
}

// the [18][64] per-lane twiddle table of stft_r8x3.hip in double
int build_r8x3_f64_tables(sg_plan& p) {
    std::vector<double> tw(18 * 64 * 2);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int j = 0; j < 64; ++j) {
        for (int r = 1; r < 8; ++r) {
            const long double a1 = -two_pi * static_cast<long double>((j * r) % 512) / 512.0L;
            tw[2 * ((r - 1) * 64 + j)] = static_cast<double>(cosl(a1));
            tw[2 * ((r - 1) * 64 + j) + 1] = static_cast<double>(sinl(a1));
            const long double a2 = -two_pi * static_cast<long double>(((j & 7) * r) % 64) / 64.0L;
            tw[2 * ((7 + r - 1) * 64 + j)] = static_cast<double>(cosl(a2));
            tw[2 * ((7 + r - 1) * 64 + j) + 1] = static_cast<double>(sinl(a2));
        }
        for (int m = 0; m < 4; ++m) {
            const long double a3 = two_pi * static_cast<long double>(j + 64 * m) / 1024.0L;
            tw[2 * ((14 + m) * 64 + j)] = static_cast<double>(cosl(a3));
            tw[2 * ((14 + m) * 64 + j) + 1] = static_cast<double>(sinl(a3));
        }
    }
    SG_HIP(hipMalloc(&p.r8_tw_dev, tw.size() * sizeof(double)));
    SG_HIP(hipMemcpy(p.r8_tw_dev, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    return SG_OK;
}

}  // namespace sg
