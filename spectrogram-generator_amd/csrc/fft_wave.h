// fft_wave.h -- device helpers shared by the per-wavefront register FFT kernels (stft_r8x3.hip, stft_rsmall.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace sg {
namespace wavefft {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    return make_float2(fmaf(a.x, w.x, -a.y * w.y), fmaf(a.x, w.y, a.y * w.x));
}
// multiply by -i
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// In-register 8-point DFT, forward sign, natural order in and out:
// a[r] <- sum_k a[k] * exp(-2*pi*i*k*r/8).
// The two 1/sqrt(2) rotations (w8, w8^3) are kept unscaled and the factor is applied inside the last
// butterfly stage as an FMA (a[1] = c4 + h*c5 ...): 4 multiplies fewer per butterfly.
__device__ __forceinline__ void radix8(float2 (&a)[8]) {
    constexpr float h = 0.70710678118654752440f;
    const float2 b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]);
    const float2 b1 = cadd(a[1], a[5]), b5 = csub(a[1], a[5]);
    const float2 b2 = cadd(a[2], a[6]), b6 = csub(a[2], a[6]);
    const float2 b3 = cadd(a[3], a[7]), b7 = csub(a[3], a[7]);
    const float2 t5 = make_float2(b5.x + b5.y, b5.y - b5.x);       // b5 * w8   * sqrt(2)
    const float2 t6 = mul_mi(b6);                                  // b6 * w8^2
    const float2 t7 = make_float2(b7.y - b7.x, -(b7.x + b7.y));    // b7 * w8^3 * sqrt(2)
    const float2 c0 = cadd(b0, b2), c2 = csub(b0, b2);
    const float2 c1 = cadd(b1, b3), c3 = mul_mi(csub(b1, b3));
    const float2 c4 = cadd(b4, t6), c6 = csub(b4, t6);
    const float2 c5 = cadd(t5, t7), c7 = mul_mi(csub(t5, t7));     // both still carry sqrt(2)
    a[0] = cadd(c0, c1); a[4] = csub(c0, c1);
    a[2] = cadd(c2, c3); a[6] = csub(c2, c3);
    a[1] = make_float2(fmaf(h, c5.x, c4.x), fmaf(h, c5.y, c4.y));
    a[5] = make_float2(fmaf(-h, c5.x, c4.x), fmaf(-h, c5.y, c4.y));
    a[3] = make_float2(fmaf(h, c7.x, c6.x), fmaf(h, c7.y, c6.y));
    a[7] = make_float2(fmaf(-h, c7.x, c6.x), fmaf(-h, c7.y, c6.y));
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// Sum over the 64 lanes as a wave-uniform value (lands in an SGPR, so it can be a scalar operand of what follows).
// DPP butterflies inside a row of 16, then row_bcast15 / row_bcast31 carry the row sums into row 3: 6 VALU + 1 readlane
// (reading the four row sums with v_readlane and adding them cost twice that).  The two row_bcast adds are written
// out: with a row mask the instruction leaves the other rows untouched, which is the "+ 0" wanted there -- through
// the update_dpp builtin the compiler materialises that zero with two extra moves per step.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);              // quad_perm [1,0,3,2]  (lane ^ 1)
    v += dpp_mov<0x4E>(v);              // quad_perm [2,3,0,1]  (lane ^ 2)
    v += dpp_mov<0x141>(v);             // row_half_mirror      (the other quad of each 8)
    v += dpp_mov<0x140>(v);             // row_mirror           (the other 8 of each 16): every lane holds its row's sum
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"     // rows 1, 3: r0+r1, r2+r3
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"       // row 3: r0+r1+r2+r3
                 "s_nop 0"
                 : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Orders this wave's LDS writes before its following LDS reads for the compiler; the hardware
// executes one wave's DS operations in issue order, so no s_barrier / s_waitcnt is needed.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroups b and b+8 share an XCD (round-robin dispatch); hand each XCD a contiguous run of
// chunk groups so that the (nperseg - hop)-sample halo between neighbouring chunks is an L2 hit.
// Bijective for any grid size (see guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

typedef float v2f __attribute__((ext_vector_type(2)));
// LDS accesses go through volatile 64-bit vectors: hipcc otherwise fuses neighbouring ds_read_b64 /
// ds_write_b64 into ds_read2_b64 / ds_write2_b64, which run at half the LDS rate on gfx950
// (MI355X_MICROARCH.md LDS table: ds_read2_b64 128 B/clk vs ds_read_b64 256 B/clk).
#ifndef SG_LDS_VOLATILE
#define SG_LDS_VOLATILE 1
#endif
#if SG_LDS_VOLATILE
typedef __attribute__((address_space(3))) volatile v2f lds_v2f;
#else
typedef __attribute__((address_space(3))) v2f lds_v2f;
#endif
__device__ __forceinline__ void lds_put(float2* p, float2 v) { *(lds_v2f*)(p) = v2f{v.x, v.y}; }
__device__ __forceinline__ float2 lds_get(const float2* p) {
    const v2f v = *(lds_v2f*)(p);
    return make_float2(v.x, v.y);
}

}  // namespace wavefft
}  // namespace sg
