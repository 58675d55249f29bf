// stft_rbig_f64.hip -- the register kernel for nperseg = nfft = 2048 / 4096 in double precision ("rbigd").
//
// Why: the reference's recordings arrive as float64 (SweepManager.py:135-136), scipy computes in the input's precision
// (scipy/signal/_spectral_py.py:1976-1981), and the GUI's nperseg spin box reaches 8192 (GUI.py:87-89): a batch of f64 sweeps at
// nperseg 2048 / 4096 ran on the LDS Stockham kernel until round 3 (a workgroup per frame, a barrier per pass).
//
// The machine mapping is stft_rbig.hip's (read that file for the index maps; tools/sim_rbig.py replays them): one wavefront = one
// frame, M = 512 T complex points, lane j keeps z[j + 64 a] for a < R = 8 T in registers (d[a0][a1], a = a0 + T a1), pass 1 = an
// R-point DFT in registers (T radix-8 over a1, constant twiddles, 8 radix-T over a0) and the lane twiddle, passes 2 / 3 = T radix-8
// per lane, every exchange 8 registers at a time through one slab per wave, split pass with only the upper half crossing lanes.
// Differences, as in stft_r8x3_f64.hip: a complex double is 16 bytes, so the slab and the tables are kept as separate real and
// imaginary planes of 8-byte elements -- every LDS access is a ds_*_b64 with the f32 kernel's index maps, conflict-free by the same
// argument; no sample prefetch and no sliding window (T = 4 holds 2 x 128 value registers as it is: one wave per SIMD).
// Algorithmic HBM bytes per frame: hop*8 + (512T+1)*8.
#include "spectro_internal.h"

#include <cmath>
#include <cstdlib>

namespace sg {
namespace {

constexpr int kS1 = 72, kS2 = 66, kSlab = 8 * kS1;       // 576 elements per plane

struct cd { double x, y; };
__device__ __forceinline__ cd cadd(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd csub(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd w) { return {fma(a.x, w.x, -a.y * w.y), fma(a.x, w.y, a.y * w.x)}; }
__device__ __forceinline__ cd mul_mi(cd a) { return {a.y, -a.x}; }

__device__ __forceinline__ void radix8(cd (&a)[8]) {      // forward 8-point DFT in registers (fft_wave.h, in double)
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
template <int T> __device__ __forceinline__ void radix_t(cd (&v)[T]);
template <> __device__ __forceinline__ void radix_t<2>(cd (&v)[2]) {
    const cd s = cadd(v[0], v[1]), d = csub(v[0], v[1]);
    v[0] = s; v[1] = d;
}
template <> __device__ __forceinline__ void radix_t<4>(cd (&v)[4]) {
    const cd s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
    const cd s13 = cadd(v[1], v[3]), d13 = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(s02, s13); v[2] = csub(s02, s13);
    v[1] = cadd(d02, d13); v[3] = csub(d02, d13);
}

// exp(-2*pi*i*n/R) for the in-register twiddles of pass 1 (compile-time indices after unrolling)
__device__ constexpr double kW16[16][2] = {{1.00000000000000000e+00, -0.00000000000000000e+00}, {9.23879532511286738e-01, -3.82683432365089782e-01}, {7.07106781186547573e-01, -7.07106781186547462e-01}, {3.82683432365089837e-01, -9.23879532511286738e-01}, {6.12323399573676604e-17, -1.00000000000000000e+00}, {-3.82683432365089726e-01, -9.23879532511286738e-01}, {-7.07106781186547462e-01, -7.07106781186547573e-01}, {-9.23879532511286738e-01, -3.82683432365089893e-01}, {-1.00000000000000000e+00, -1.22464679914735321e-16}, {-9.23879532511286850e-01, 3.82683432365089671e-01}, {-7.07106781186547684e-01, 7.07106781186547462e-01}, {-3.82683432365090337e-01, 9.23879532511286516e-01}, {-1.83697019872102969e-16, 1.00000000000000000e+00}, {3.82683432365090004e-01, 9.23879532511286627e-01}, {7.07106781186547351e-01, 7.07106781186547684e-01}, {9.23879532511286516e-01, 3.82683432365090392e-01}};
__device__ constexpr double kW32[32][2] = {{1.00000000000000000e+00, -0.00000000000000000e+00}, {9.80785280403230431e-01, -1.95090322016128248e-01}, {9.23879532511286738e-01, -3.82683432365089782e-01}, {8.31469612302545236e-01, -5.55570233019602178e-01}, {7.07106781186547573e-01, -7.07106781186547462e-01}, {5.55570233019602289e-01, -8.31469612302545236e-01}, {3.82683432365089837e-01, -9.23879532511286738e-01}, {1.95090322016128331e-01, -9.80785280403230431e-01}, {6.12323399573676604e-17, -1.00000000000000000e+00}, {-1.95090322016128193e-01, -9.80785280403230431e-01}, {-3.82683432365089726e-01, -9.23879532511286738e-01}, {-5.55570233019601956e-01, -8.31469612302545458e-01}, {-7.07106781186547462e-01, -7.07106781186547573e-01}, {-8.31469612302545347e-01, -5.55570233019602178e-01}, {-9.23879532511286738e-01, -3.82683432365089893e-01}, {-9.80785280403230431e-01, -1.95090322016128609e-01}, {-1.00000000000000000e+00, -1.22464679914735321e-16}, {-9.80785280403230431e-01, 1.95090322016128359e-01}, {-9.23879532511286850e-01, 3.82683432365089671e-01}, {-8.31469612302545458e-01, 5.55570233019601956e-01}, {-7.07106781186547684e-01, 7.07106781186547462e-01}, {-5.55570233019602178e-01, 8.31469612302545236e-01}, {-3.82683432365090337e-01, 9.23879532511286516e-01}, {-1.95090322016128664e-01, 9.80785280403230320e-01}, {-1.83697019872102969e-16, 1.00000000000000000e+00}, {1.95090322016128304e-01, 9.80785280403230431e-01}, {3.82683432365090004e-01, 9.23879532511286627e-01}, {5.55570233019601845e-01, 8.31469612302545458e-01}, {7.07106781186547351e-01, 7.07106781186547684e-01}, {8.31469612302545236e-01, 5.55570233019602178e-01}, {9.23879532511286516e-01, 3.82683432365090392e-01}, {9.80785280403230320e-01, 1.95090322016128720e-01}};
template <int R> __device__ __forceinline__ cd const_tw(int n) {
    return R == 16 ? cd{kW16[n & 15][0], kW16[n & 15][1]} : cd{kW32[n & 31][0], kW32[n & 31][1]};
}

typedef __attribute__((address_space(3))) volatile double lds_f64;
struct Planes {                     // complex values in LDS: real plane, imaginary plane
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

struct BigDParams {
    const double* x;
    int64_t clip_stride;
    int n_frames, hop;
    int64_t total_frames;
    int n_waves;
    double* out;
    int64_t out_clip_stride;
    const double2* win2;      // [M]
    const double2* tw;        // [(R-1) + 7 + R/2][64]
    double scale;
    int k_lo, k_hi;           // MODE 2: bins of the band
};

template <int T> struct WavesD { static constexpr int value = T == 4 ? 4 : 8; };      // per workgroup = per CU (LDS: 125 / 121 KiB)

// MODE 0 psd, 1 magnitude, 2 band power (A11): out[clip][frame] = sum of PSD bins [k_lo, k_hi]
template <int T, bool DETREND, int MODE>
__global__ __launch_bounds__((64 * WavesD<T>::value), (WavesD<T>::value / 4)) void stft_rbig_f64_kernel(const BigDParams p) {
    constexpr int R = 8 * T, M = 64 * R, NB = M + 1, kWaves = WavesD<T>::value;
    constexpr int kTw1 = M, kTw2 = kTw1 + (R - 1) * 64, kTw3 = kTw2 + 7 * 64, kTabs = kTw3 + (R / 2) * 64;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Planes tab{lds, lds + kTabs};                                  // window, t1, t2, t3
    double* const slab = lds + 2 * kTabs + wave * 2 * kSlab;
    const Planes sl{slab, slab + kSlab};

    {   // tables into LDS, once per workgroup; sqrt of the PSD scale rides on the window (stft_r8x3.hip)
        const double sq = sqrt(MODE != 1 ? p.scale * 0.5 : p.scale * 0.25);
        for (int i = threadIdx.x; i < M; i += 64 * kWaves) { const double2 v = p.win2[i]; tab.put(i, {v.x * sq, v.y * sq}); }
        for (int i = threadIdx.x; i < kTabs - M; i += 64 * kWaves) { const double2 v = p.tw[i]; tab.put(M + i, {v.x, v.y}); }
    }
    __syncthreads();

    const int lw = xcd_remap(blockIdx.x, gridDim.x) * kWaves + wave;
    if (lw >= p.n_waves) return;
    int64_t g = p.total_frames * lw / p.n_waves;
    const int64_t g_end = p.total_frames * (lw + 1) / p.n_waves;

    const int wtab = lane, t1 = kTw1 + lane, t2 = kTw2 + lane, t3 = kTw3 + lane;     // + 64 * row
    const int j0 = lane & 7, hi = lane >> 3;
    const int x1w = hi * kS1 + j0, x1r = lane;            // + 8 r1 | + b kS1
    const int x2w = j0 * kS2 + hi, x2r = lane;            // + ((8q + R s) % 64) | + j kS2
    const int x3w = lane, x3b = 256 - lane;               // + 64 slot | - 64 (c - 4i)
    const double r0 = (MODE != 1 && lane == 0) ? 0.5 : 1.0;

    // (clip, frame) of the run's first frame by one division; after that they advance incrementally
    int clip = static_cast<int>(g / p.n_frames);
    int f = static_cast<int>(g - static_cast<int64_t>(clip) * p.n_frames);
    for (; g < g_end; ++g) {
        double* const orow = p.out + static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(f) * (MODE == 2 ? 1 : NB);
        cd d[T][8];
        {
            const double* const src = p.x + static_cast<int64_t>(clip) * p.clip_stride + static_cast<int64_t>(f) * p.hop + 2 * lane;
#pragma unroll
            for (int a0 = 0; a0 < T; ++a0)
#pragma unroll
                for (int a1 = 0; a1 < 8; ++a1) {
                    const double2 v = *reinterpret_cast<const double2*>(src + 128 * (a0 + T * a1));
                    d[a0][a1] = {v.x, v.y};
                }
        }
        double bsum = 0.0;                                   // MODE 2: this lane's share of the band sum
        if (DETREND) {                                       // A3 (scipy:2191, detrend 'constant')
            double s = 0.0;
#pragma unroll
            for (int a0 = 0; a0 < T; ++a0)
#pragma unroll
                for (int a1 = 0; a1 < 8; ++a1) s += d[a0][a1].x + d[a0][a1].y;
            const double mean = wave_sum(s) * (1.0 / (2 * M));
#pragma unroll
            for (int a0 = 0; a0 < T; ++a0)
#pragma unroll
                for (int a1 = 0; a1 < 8; ++a1) { d[a0][a1].x -= mean; d[a0][a1].y -= mean; }
        }
#pragma unroll
        for (int a0 = 0; a0 < T; ++a0)
#pragma unroll
            for (int a1 = 0; a1 < 8; ++a1) {                 // A4
                const cd w = tab.get(wtab + 64 * (a0 + T * a1));
                d[a0][a1].x *= w.x; d[a0][a1].y *= w.y;
            }
        // ---- pass 1: R-point DFT over a = a0 + T*a1 ----
#pragma unroll
        for (int a0 = 0; a0 < T; ++a0) {
            radix8(d[a0]);                                   // over a1 -> r1
            if (a0 > 0) {
#pragma unroll
                for (int r1 = 1; r1 < 8; ++r1) d[a0][r1] = cmul(d[a0][r1], const_tw<R>(a0 * r1));
            }
        }
#pragma unroll
        for (int r1 = 0; r1 < 8; ++r1) {                     // over a0 -> r0 ; r = r1 + 8*r0
            cd v[T];
#pragma unroll
            for (int a0 = 0; a0 < T; ++a0) v[a0] = d[a0][r1];
            radix_t<T>(v);
#pragma unroll
            for (int q = 0; q < T; ++q) d[q][r1] = v[q];
        }
#pragma unroll
        for (int q = 0; q < T; ++q)
#pragma unroll
            for (int r1 = 0; r1 < 8; ++r1)
                if (q + r1 > 0) d[q][r1] = cmul(d[q][r1], tab.get(t1 + 64 * (r1 + 8 * q - 1)));
#pragma unroll
        for (int q = 0; q < T; ++q) {                        // exchange 1, one group of 8 at a time through the slab
#pragma unroll
            for (int r1 = 0; r1 < 8; ++r1) sl.put(x1w + 8 * r1, d[q][r1]);
            wave_lds_fence();
#pragma unroll
            for (int b = 0; b < 8; ++b) d[q][b] = sl.get(x1r + b * kS1);
            wave_lds_fence();
        }
        // ---- pass 2 ----
#pragma unroll
        for (int q = 0; q < T; ++q) {
            radix8(d[q]);
#pragma unroll
            for (int s = 1; s < 8; ++s) d[q][s] = cmul(d[q][s], tab.get(t2 + 64 * (s - 1)));
        }
        cd e[T][8];                                          // pass-3 operands: e[q3][j]
#pragma unroll
        for (int q3 = 0; q3 < T; ++q3) {                     // exchange 2: group q3 collects the (q, s) with (8q + R*s) / 64 == q3
#pragma unroll
            for (int q = 0; q < T; ++q)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int uu = 8 * q + R * s;            // + r1 (= hi) < 8 never carries into the next 64
                    if (uu / 64 == q3) sl.put(x2w + (uu % 64), d[q][s]);
                }
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 8; ++j) e[q3][j] = sl.get(x2r + j * kS2);
            wave_lds_fence();
        }
        // ---- pass 3: e[q3][t] = Z[lane + 64*(q3 + T*t)] ----
#pragma unroll
        for (int q3 = 0; q3 < T; ++q3) radix8(e[q3]);
#define SG_Z(c) e[(c) % T][(c) / T]                          // Z[lane + 64*c]
        // ---- split pass + epilogue: lower blocks c (registers) pair with upper blocks R-1-c (and element 0 of block R-c) of the
        //      mirrored lane; four blocks per trip through the slab ----
#pragma unroll
        for (int i = 0; i < R / 8; ++i) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) sl.put(x3w + 64 * s4, SG_Z(R - 4 * i - 4 + s4));
            if (lane == 0) sl.put(256, i == 0 ? SG_Z(0) : SG_Z(R - 4 * i));   // i = 0: Z[M] := Z[0]
            wave_lds_fence();
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = 4 * i + cc;
                const cd A = SG_Z(c);
                const cd B = sl.get(x3b - 64 * cc);
                const cd cs = tab.get(t3 + 64 * c);
                const cd S = {A.x + B.x, A.y - B.y};
                const cd D = {A.x - B.x, A.y + B.y};
                const cd Tt = {fma(cs.y, D.x, -cs.x * D.y), fma(cs.x, D.x, cs.y * D.y)};
                const cd Xk = csub(S, Tt), Xm = cadd(S, Tt);
                double pk = fma(Xk.x, Xk.x, Xk.y * Xk.y), pm = fma(Xm.x, Xm.x, Xm.y * Xm.y);
                if (MODE != 1 && c == 0) { pk *= r0; pm *= r0; }
                if (MODE == 1) { pk = sqrt(pk); pm = sqrt(pm); }
                const int k = lane + 64 * c;
                if (MODE == 2) {
                    if (k >= p.k_lo && k <= p.k_hi) bsum += pk;
                    if (M - k >= p.k_lo && M - k <= p.k_hi) bsum += pm;
                } else {
                    orow[k] = pk;
                    orow[M - k] = pm;
                }
            }
            wave_lds_fence();
        }
        {   // k = M/2: lane 0, block c = R/2
            const cd z = SG_Z(R / 2);
            const double zx = __shfl(z.x, 0), zy = __shfl(z.y, 0);
            double pq = fma(zx, zx, zy * zy) * 4.0;
            if (MODE == 1) pq = sqrt(pq);
            if (MODE == 2) {
                if (lane == 0 && M / 2 >= p.k_lo && M / 2 <= p.k_hi) bsum += pq;
                bsum = wave_sum(bsum);
                if (lane == 0) orow[0] = bsum;
            } else {
                orow[M / 2] = pq;                            // wave-uniform store
            }
        }
#undef SG_Z
        if (++f == p.n_frames) { f = 0; ++clip; }
    }
}

template <int T, bool DETREND>
int launch_td(const BigDParams& prm, hipStream_t s, int mode, bool band, int n_cu) {
    constexpr int R = 8 * T, M = 64 * R, kWaves = WavesD<T>::value;
    auto k0 = stft_rbig_f64_kernel<T, DETREND, 0>;
    auto k1 = stft_rbig_f64_kernel<T, DETREND, 1>;
    auto k2 = stft_rbig_f64_kernel<T, DETREND, 2>;
    auto kern = band ? k2 : mode == SG_MODE_PSD ? k0 : k1;
    const size_t lds = (2 * (static_cast<size_t>(M) + (R - 1 + 7 + R / 2) * 64) + static_cast<size_t>(kWaves) * 2 * kSlab) * sizeof(double);
    BigDParams p = prm;
    int64_t n_waves = static_cast<int64_t>(n_cu) * kWaves;              // one workgroup per CU
    if (n_waves > p.total_frames) n_waves = p.total_frames;
    p.n_waves = static_cast<int>(n_waves);
    const int n_wg = static_cast<int>((n_waves + kWaves - 1) / kWaves);
    SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * kWaves), lds, s, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft_rbig_f64 launch");
}

template <int T>
int launch_t(const sg_plan& p, const StftArgs& a) {
    BigDParams prm{};
    prm.x = static_cast<const double*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.total_frames = a.n_frames * a.n_clips;
    prm.out = static_cast<double*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.win2 = static_cast<const double2*>(p.win_dev);
    prm.tw = static_cast<const double2*>(p.r8_tw_dev);
    prm.scale = p.scale;
    prm.k_lo = a.k_lo; prm.k_hi = a.k_hi;
    const bool band = a.band_mode != 0;                     // run_stft has checked: psd plan, 0 <= k_lo <= k_hi < n_bins
    return p.detrend == SG_DETREND_CONSTANT ? launch_td<T, true>(prm, a.stream, p.mode, band, p.n_cu)
                                            : launch_td<T, false>(prm, a.stream, p.mode, band, p.n_cu);
}

}  // namespace

// 16-byte aligned double2 loads; anything else of the plan (odd hops, unaligned clips) is served by the Stockham kernel
bool rbig_f64_can_run(const sg_plan& p, const StftArgs& a) {
    return p.dtype == SG_F64 && !a.in_i16 && !a.db_mode && a.mel_ipl == 0 && (p.hop % 2 == 0) &&
           (a.clip_stride % 2 == 0 || a.n_clips == 1) && (reinterpret_cast<uintptr_t>(a.x) % 16 == 0) && a.n_frames <= INT32_MAX;
}

int launch_rbig_f64(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    return p.nfft == 2048 ? launch_t<2>(p, a) : launch_t<4>(p, a);
}

// the per-lane twiddle table [(R-1) + 7 + R/2][64] of stft_rbig.hip (R = nfft/128) in double
int build_rbig_f64_tables(sg_plan& p) {
    const int R = p.nfft / 128, M = 64 * R;
    std::vector<double> tw(static_cast<size_t>(R - 1 + 7 + R / 2) * 64 * 2);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    auto put = [&](int row, int j, long double ang) {
        tw[2 * (static_cast<size_t>(row) * 64 + j)] = static_cast<double>(cosl(ang));
        tw[2 * (static_cast<size_t>(row) * 64 + j) + 1] = static_cast<double>(sinl(ang));
    };
    for (int j = 0; j < 64; ++j) {
        for (int r = 1; r < R; ++r) put(r - 1, j, -two_pi * static_cast<long double>((static_cast<long long>(j) * r) % M) / M);
        for (int s = 1; s < 8; ++s) put(R - 1 + s - 1, j, -two_pi * static_cast<long double>(((j & 7) * s) % 64) / 64.0L);
        for (int c = 0; c < R / 2; ++c) put(R - 1 + 7 + c, j, two_pi * static_cast<long double>(j + 64 * c) / (2.0L * M));
    }
    SG_HIP(hipMalloc(&p.r8_tw_dev, tw.size() * sizeof(double)));
    SG_HIP(hipMemcpy(p.r8_tw_dev, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    return SG_OK;
}

}  // namespace sg
