// cfft_wave.h -- the L-point complex FFT of one wavefront in registers, L = 64 * 8T (T = 1 / 2 / 4: 512 / 1024 / 2048 points), shared by the
// register chirp-z kernels (stft_rblue.hip: one wavefront per frame; stft_rbluew.hip: two or four wavefronts per frame).  Passes and
// exchanges are those of stft_rbig.hip (index maps: tools/sim_rbig.py): radix-8 over the lane's rows, a radix-T pass, per-lane twiddles,
// exchange 1 through the wave's padded LDS slab, radix-8, twiddles, exchange 2, radix-8.
#pragma once
#include "fft_wave.h"

namespace sg {
namespace wavefft {

constexpr int kS1 = 72, kS2 = 66;                            // padded row strides of the two exchanges (float2 units)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f lds_get2(const float2* p) { return *(__attribute__((address_space(3))) volatile v4f*)(p); }

template <int T> __device__ __forceinline__ void radix_t(float2 (&v)[T]);
template <> __device__ __forceinline__ void radix_t<1>(float2 (&)[1]) {}
template <> __device__ __forceinline__ void radix_t<2>(float2 (&v)[2]) {
    const float2 s = cadd(v[0], v[1]), d = csub(v[0], v[1]);
    v[0] = s; v[1] = d;
}
template <> __device__ __forceinline__ void radix_t<4>(float2 (&v)[4]) {
    const float2 s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
    const float2 s13 = cadd(v[1], v[3]), d13 = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(s02, s13); v[2] = csub(s02, s13);
    v[1] = cadd(d02, d13); v[3] = csub(d02, d13);
}

// exp(-2*pi*i*n/R), R = 16 / 32: compile-time indices after unrolling
template <int R> __device__ __forceinline__ float2 const_tw(int n) {
    constexpr float kPi = 3.14159265358979323846f;
    const int m = n & (R - 1);                               // (R = 8: never called with a0 > 0)
    // (cosf / sinf of a constant fold at compile time)
    return make_float2(__builtin_cosf(-2.0f * kPi * m / R), __builtin_sinf(-2.0f * kPi * m / R));
}

// LDS addresses of one wave's transform: its twiddle rows and its exchange slab (8 * kS1 float2 for T < 4, 2 * 8 * kS1 for T = 4)
struct CfftLds {
    const float2* tw1;       // table [R/2][64][2] + 2 * lane: rows r = 2i+1, 2i+2 of exp(-2 pi i lane r / L) side by side (R - 1 rows, padded)
    const float2* tw2;       // table [7][64] + lane: exp(-2 pi i (lane & 7) s / 64), s = 1..7
    float2* x1w;             // slab + (lane >> 3) * kS1 + (lane & 7)     (+ 8 r1)
    float2* x1r;             // slab + lane                               (+ b kS1)
    float2* x2w;             // slab + (lane & 7) * kS2 + (lane >> 3)     (+ (8q + R s) % 64)
    float2* x2r;             // slab + lane                               (+ j kS2)
};
__device__ __forceinline__ CfftLds cfft_lds(const float2* tw1_table, const float2* tw2_table, float2* slab, int lane) {
    const int j0 = lane & 7, hi = lane >> 3;
    return CfftLds{tw1_table + 2 * lane, tw2_table + lane, slab + hi * kS1 + j0, slab + lane, slab + j0 * kS2 + hi, slab + lane};
}

// in d[a0][a1] = y[lane + 64*(a0 + T*a1)] (destroyed), out e[q3][t] = Y[lane + 64*(q3 + T*t)] -- the same index form, so a second
// transform takes e as its d
template <int T>
__device__ __forceinline__ void cfft_wave(float2 (&d)[T][8], float2 (&e)[T][8], const CfftLds& c) {
    constexpr int R = 8 * T;
#pragma unroll
    for (int a0 = 0; a0 < T; ++a0) {
        radix8(d[a0]);                                 // over a1 -> r1
        if (a0 > 0) {
#pragma unroll
            for (int r1 = 1; r1 < 8; ++r1) d[a0][r1] = cmul(d[a0][r1], const_tw<R>(a0 * r1));
        }
    }
#pragma unroll
    for (int r1 = 0; r1 < 8; ++r1) {                     // over a0 -> r0 ; r = r1 + 8*r0
        float2 v[T];
#pragma unroll
        for (int a0 = 0; a0 < T; ++a0) v[a0] = d[a0][r1];
        radix_t<T>(v);
#pragma unroll
        for (int r0 = 0; r0 < T; ++r0) d[r0][r1] = v[r0];
    }
#pragma unroll
    for (int i = 0; i < R - 1; i += 2) {                 // table rows i, i + 1 <-> r = i + 1, i + 2
        const v4f w = lds_get2(c.tw1 + (i >> 1) * 128);
        d[(i + 1) / 8][(i + 1) % 8] = cmul(d[(i + 1) / 8][(i + 1) % 8], make_float2(w.x, w.y));
        if (i + 2 < R) d[(i + 2) / 8][(i + 2) % 8] = cmul(d[(i + 2) / 8][(i + 2) % 8], make_float2(w.z, w.w));
    }
#pragma unroll
    for (int q = 0; q < T; ++q) {                        // exchange 1, one group of 8 at a time through the slab
#pragma unroll
        for (int r1 = 0; r1 < 8; ++r1) lds_put(c.x1w + 8 * r1, d[q][r1]);
        wave_lds_fence();
#pragma unroll
        for (int b = 0; b < 8; ++b) d[q][b] = lds_get(c.x1r + b * kS1);
        wave_lds_fence();
    }
#pragma unroll
    for (int q = 0; q < T; ++q) {                        // pass 2
        radix8(d[q]);
#pragma unroll
        for (int s = 1; s < 8; ++s) d[q][s] = cmul(d[q][s], lds_get(c.tw2 + 64 * (s - 1)));
    }
#pragma unroll
    for (int q3 = 0; q3 < T; ++q3) {                     // exchange 2: group q3 collects the (q, s) with (8q + R*s) / 64 == q3
#pragma unroll
        for (int q = 0; q < T; ++q)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int uu = 8 * q + R * s;
                if (uu / 64 == q3) lds_put(c.x2w + (uu % 64), d[q][s]);
            }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 8; ++j) e[q3][j] = lds_get(c.x2r + j * kS2);
        wave_lds_fence();
    }
#pragma unroll
    for (int q3 = 0; q3 < T; ++q3) radix8(e[q3]);        // pass 3: e[q3][t] = Y[lane + 64*(q3 + T*t)]
}

}  // namespace wavefft
}  // namespace sg
