// Micro-benchmark (design aid): issue cost of the cross-lane VALU forms an in-register transpose would use on gfx950:
// v_permlane32_swap_b32, v_permlane16_swap_b32, v_cndmask_b32 with a DPP source (row_ror:8, quad_perm), v_mov_b32_dpp with a
// bank mask, next to plain v_fma_f32.  8 independent chains, 1 / 2 / 4 waves per SIMD; ns per wave-instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/xlane_rate.hip -o tools/ubench/xlane_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define REP8(OP) OP(0, 1) OP(2, 3) OP(4, 5) OP(6, 7) OP(1, 2) OP(3, 4) OP(5, 6) OP(7, 0)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* st) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 1) {   // 8 swaps on 4 register pairs (each pair swapped twice: data returns)
                asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                             "v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 2) {
                asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                             "v_permlane16_swap_b32 %1, %2\n v_permlane16_swap_b32 %3, %4\n v_permlane16_swap_b32 %5, %6\n v_permlane16_swap_b32 %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 3) {   // v_cndmask with a DPP source, row_ror:8
                asm volatile("v_cndmask_b32_dpp %0, %1, %0, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %2, %3, %2, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n"
                             "v_cndmask_b32_dpp %4, %5, %4, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %6, %7, %6, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n"
                             "v_cndmask_b32_dpp %1, %2, %1, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %3, %4, %3, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n"
                             "v_cndmask_b32_dpp %5, %6, %5, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %7, %0, %7, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 4) {   // quad_perm [1,0,3,2]
                asm volatile("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %2, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_cndmask_b32_dpp %4, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %6, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_cndmask_b32_dpp %1, %2, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %3, %4, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_cndmask_b32_dpp %5, %6, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %7, %0, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 5) {   // v_mov_b32_dpp with a bank mask (writes half the lanes)
                asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                             "v_mov_b32_dpp %4, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %6, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                             "v_mov_b32_dpp %1, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n v_mov_b32_dpp %3, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                             "v_mov_b32_dpp %5, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n v_mov_b32_dpp %7, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 6) {   // plain v_cndmask_b32 (no DPP)
                asm volatile("v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %3, %2, vcc\n v_cndmask_b32 %4, %5, %4, vcc\n v_cndmask_b32 %6, %7, %6, vcc\n"
                             "v_cndmask_b32 %1, %2, %1, vcc\n v_cndmask_b32 %3, %4, %3, vcc\n v_cndmask_b32 %5, %6, %5, vcc\n v_cndmask_b32 %7, %0, %7, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 8) {   // v_fmamk_f32: literal constant K (what `fmaf(h, x, y)` with a float literal compiles to)
                asm volatile("v_fmamk_f32 %0, %0, 0x3f3504f3, %8\n v_fmamk_f32 %1, %1, 0x3f3504f3, %8\n v_fmamk_f32 %2, %2, 0x3f3504f3, %8\n v_fmamk_f32 %3, %3, 0x3f3504f3, %8\n"
                             "v_fmamk_f32 %4, %4, 0x3f3504f3, %8\n v_fmamk_f32 %5, %5, 0x3f3504f3, %8\n v_fmamk_f32 %6, %6, 0x3f3504f3, %8\n v_fmamk_f32 %7, %7, 0x3f3504f3, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            } else if (KIND == 9) {   // v_mul_f32 with a literal
                asm volatile("v_mul_f32 %0, 0x3f3504f3, %0\n v_mul_f32 %1, 0x3f3504f3, %1\n v_mul_f32 %2, 0x3f3504f3, %2\n v_mul_f32 %3, 0x3f3504f3, %3\n"
                             "v_mul_f32 %4, 0x3f3504f3, %4\n v_mul_f32 %5, 0x3f3504f3, %5\n v_mul_f32 %6, 0x3f3504f3, %6\n v_mul_f32 %7, 0x3f3504f3, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 10) {  // v_mul_f32 with an SGPR source
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b));
            } else if (KIND == 11) {  // v_mul_f32 with an inline constant (0.5)
                asm volatile("v_mul_f32 %0, 0.5, %0\n v_mul_f32 %1, 0.5, %1\n v_mul_f32 %2, 0.5, %2\n v_mul_f32 %3, 0.5, %3\n"
                             "v_mul_f32 %4, 0.5, %4\n v_mul_f32 %5, 0.5, %5\n v_mul_f32 %6, 0.5, %6\n v_mul_f32 %7, 0.5, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 12) {  // v_mul_f32 with a VGPR constant
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if (KIND == 13) {  // v_fma_f32 (VOP3) with an SGPR multiplier
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));
            } else if (KIND == 14) {  // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(*reinterpret_cast<double*>(&a0)), "+v"(*reinterpret_cast<double*>(&a2)), "+v"(*reinterpret_cast<double*>(&a4)), "+v"(*reinterpret_cast<double*>(&a6)) : "v"(*reinterpret_cast<const double*>(&b)));
            } else if (KIND == 7) {   // v_mov_b32 plain
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n v_mov_b32 %1, %2\n v_mov_b32 %3, %4\n v_mov_b32 %5, %6\n v_mov_b32 %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { st[0] = __builtin_amdgcn_s_memtime() - t0; st[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char* name, float* out) {
    for (int occ : {1, 2, 4}) {
        const int iters = 2000, nwg = 256 * occ;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        unsigned long long* st; CHECK(hipHostMalloc(&st, 16)); st[0] = st[1] = 1;
        hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, out, 100, st);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, out, iters, st);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double inst_per_simd = (double)occ * iters * 8 * 16, ghz = st[0] / (st[1] * 10.0);
        printf("%-34s waves/SIMD=%d  %.3f ms  %.2f ns = %.2f cycles per wave-instr per SIMD (last block's clock %.2f GHz)\n",
               name, occ, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * ghz, ghz);
        CHECK(hipHostFree(st));
    }
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>("v_fma_f32", out);
    run<1>("v_permlane32_swap_b32", out);
    run<2>("v_permlane16_swap_b32", out);
    run<3>("v_cndmask_b32_dpp row_ror:8", out);
    run<4>("v_cndmask_b32_dpp quad_perm", out);
    run<5>("v_mov_b32_dpp row_ror:8 bank_mask", out);
    run<6>("v_cndmask_b32", out);
    run<7>("v_mov_b32", out);
    run<8>("v_fmamk_f32 (literal K)", out);
    run<9>("v_mul_f32 literal", out);
    run<10>("v_mul_f32 SGPR source", out);
    run<11>("v_mul_f32 inline constant 0.5", out);
    run<12>("v_mul_f32 VGPR constant", out);
    run<13>("v_fma_f32 SGPR multiplier", out);
    return 0;
}
