// cfft_wave_f64.h -- the L-point complex FFT of one wavefront in registers in double precision, L = 64 * 8T (T = 1 / 2: 512 / 1024 points),
// shared by the double-precision register chirp-z kernels (stft_rblue_f64.hip: one wavefront per frame; stft_rbluew_f64.hip: two, four or
// eight wavefronts per frame).  Passes and exchanges are those of stft_rbig_f64.hip (index maps: tools/sim_rbig.py); a complex double is
// 16 bytes, so slab and tables are separate real / imaginary planes of 8-byte elements and every LDS access is a ds_*_b64.
#pragma once
#include <hip/hip_runtime.h>

namespace sg {
namespace wavefft64 {

constexpr int kS1 = 72, kS2 = 66, kSlab = 8 * kS1;       // 576 elements per plane: one exchange group, and the N2 + 1 <= 513 split entries

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
template <> __device__ __forceinline__ void radix_t<1>(cd (&)[1]) {}
template <> __device__ __forceinline__ void radix_t<2>(cd (&v)[2]) {
    const cd s = cadd(v[0], v[1]), d = csub(v[0], v[1]);
    v[0] = s; v[1] = d;
}
// exp(-2*pi*i*n/16) for the in-register twiddles of pass 1 at T = 2 (compile-time indices after unrolling)
__device__ constexpr double kW16[16][2] = {{1.00000000000000000e+00, -0.00000000000000000e+00}, {9.23879532511286738e-01, -3.82683432365089782e-01}, {7.07106781186547573e-01, -7.07106781186547462e-01}, {3.82683432365089837e-01, -9.23879532511286738e-01}, {6.12323399573676604e-17, -1.00000000000000000e+00}, {-3.82683432365089726e-01, -9.23879532511286738e-01}, {-7.07106781186547462e-01, -7.07106781186547573e-01}, {-9.23879532511286738e-01, -3.82683432365089893e-01}, {-1.00000000000000000e+00, -1.22464679914735321e-16}, {-9.23879532511286849e-01, 3.82683432365089671e-01}, {-7.07106781186547684e-01, 7.07106781186547351e-01}, {-3.82683432365090004e-01, 9.23879532511286627e-01}, {-1.83697019872102977e-16, 1.00000000000000000e+00}, {3.82683432365089615e-01, 9.23879532511286849e-01}, {7.07106781186547351e-01, 7.07106781186547684e-01}, {9.23879532511286627e-01, 3.82683432365090060e-01}};
__device__ __forceinline__ cd const_tw16(int n) { return cd{kW16[n & 15][0], kW16[n & 15][1]}; }

typedef __attribute__((address_space(3))) volatile double lds_f64;
struct Planes {                     // complex values in LDS: real plane, imaginary plane
    double* re; double* im;
    __device__ __forceinline__ void put(int i, cd v) const { *(lds_f64*)(re + i) = v.x; *(lds_f64*)(im + i) = v.y; }
    __device__ __forceinline__ cd get(int i) const { return {*(lds_f64*)(re + i), *(lds_f64*)(im + i)}; }
    // the planes seen from element i: a per-lane base whose constant offsets fold into the ds instructions' immediates
    __device__ __forceinline__ Planes at(int i) const { return {re + i, im + i}; }
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

// in d[a0][a1] = y[lane + 64 (a0 + T a1)] (destroyed), out e[q3][t] = Y[lane + 64 (q3 + T t)] -- the same index form, so a second transform
// takes e as its d.  tab: the table planes, with rows [R - 1][64] of exp(-2 pi i lane r / L) at tw1 and [7][64] of exp(-2 pi i (lane & 7) s / 64)
// at tw2; sl: this wave's exchange slab (kSlab elements per plane).
template <int T>
__device__ __forceinline__ void cfft_wave_f64(cd (&d)[T][8], cd (&e)[T][8], const Planes& tab, int tw1, int tw2, const Planes& sl, int lane) {
    constexpr int R = 8 * T;
    const int j0 = lane & 7, hi = lane >> 3;
    const int x1w = hi * kS1 + j0, x1r = lane;            // + 8 r1 | + b kS1
    const int x2w = j0 * kS2 + hi, x2r = lane;            // + ((8q + R s) % 64) | + j kS2
#pragma unroll
    for (int a0 = 0; a0 < T; ++a0) {
        radix8(d[a0]);
        if (a0 > 0) {
#pragma unroll
            for (int r1 = 1; r1 < 8; ++r1) d[a0][r1] = cmul(d[a0][r1], const_tw16(a0 * r1));
        }
    }
#pragma unroll
    for (int r1 = 0; r1 < 8; ++r1) {
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
            if (q + r1 > 0) d[q][r1] = cmul(d[q][r1], tab.get(tw1 + lane + 64 * (r1 + 8 * q - 1)));
#pragma unroll
    for (int q = 0; q < T; ++q) {                        // exchange 1
#pragma unroll
        for (int r1 = 0; r1 < 8; ++r1) sl.put(x1w + 8 * r1, d[q][r1]);
        wave_lds_fence();
#pragma unroll
        for (int b = 0; b < 8; ++b) d[q][b] = sl.get(x1r + b * kS1);
        wave_lds_fence();
    }
#pragma unroll
    for (int q = 0; q < T; ++q) {                        // pass 2
        radix8(d[q]);
#pragma unroll
        for (int s = 1; s < 8; ++s) d[q][s] = cmul(d[q][s], tab.get(tw2 + lane + 64 * (s - 1)));
    }
#pragma unroll
    for (int q3 = 0; q3 < T; ++q3) {                     // exchange 2
#pragma unroll
        for (int q = 0; q < T; ++q)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int uu = 8 * q + R * s;
                if (uu / 64 == q3) sl.put(x2w + (uu % 64), d[q][s]);
            }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 8; ++j) e[q3][j] = sl.get(x2r + j * kS2);
        wave_lds_fence();
    }
#pragma unroll
    for (int q3 = 0; q3 < T; ++q3) radix8(e[q3]);        // pass 3
}

}  // namespace wavefft64
}  // namespace sg
