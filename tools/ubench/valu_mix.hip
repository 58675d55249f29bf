// Micro-benchmark: issue cost of the instruction forms the register-FFT loop uses, at 4 waves/SIMD (design aid).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_mix.hip -o tools/ubench/valu_mix.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float v2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float sc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    v2 q0 = {a1, a0}, q1 = {a3, a2}, q2 = {a5, a4}, q3 = {a7, a6};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (KIND == 0)        // baseline: 8 independent v_fmac_f32 (VOP2)
            asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                         "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        else if (KIND == 1)   // v_fmamk_f32 with a 32-bit literal
            asm volatile("v_fmamk_f32 %0, %0, 0x3f3504f3, %8\n v_fmamk_f32 %1, %1, 0x3f3504f3, %8\n v_fmamk_f32 %2, %2, 0x3f3504f3, %8\n v_fmamk_f32 %3, %3, 0x3f3504f3, %8\n"
                         "v_fmamk_f32 %4, %4, 0x3f3504f3, %8\n v_fmamk_f32 %5, %5, 0x3f3504f3, %8\n v_fmamk_f32 %6, %6, 0x3f3504f3, %8\n v_fmamk_f32 %7, %7, 0x3f3504f3, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        else if (KIND == 2)   // v_mul_f32_e64 with a neg modifier (VOP3)
            asm volatile("v_mul_f32_e64 %0, %0, -%8\n v_mul_f32_e64 %1, %1, -%8\n v_mul_f32_e64 %2, %2, -%8\n v_mul_f32_e64 %3, %3, -%8\n"
                         "v_mul_f32_e64 %4, %4, -%8\n v_mul_f32_e64 %5, %5, -%8\n v_mul_f32_e64 %6, %6, -%8\n v_mul_f32_e64 %7, %7, -%8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        else if (KIND == 3)   // v_fma_f32 with an SGPR operand
            asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                         "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc), "v"(c));
        else if (KIND == 4)   // v_pk_add_f32 with neg modifiers (the complex subtract the compiler emits)
            asm volatile("v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %5 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %6 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %7 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %0, %0, %5\n v_pk_add_f32 %1, %1, %6\n v_pk_add_f32 %2, %2, %7\n v_pk_add_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0), "v"(q1), "v"(q2), "v"(q3));
        else if (KIND == 5)   // v_mov_b64
            asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %5\n v_mov_b64 %2, %6\n v_mov_b64 %3, %7\n"
                         "v_mov_b64 %0, %5\n v_mov_b64 %1, %6\n v_mov_b64 %2, %7\n v_mov_b64 %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0), "v"(q1), "v"(q2), "v"(q3));
        else if (KIND == 6)   // dependent chain: each fmac needs the previous one (ILP 1)
            asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n"
                         "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n"
                         : "+v"(a0) : "v"(b), "v"(c));
        else if (KIND == 7)   // two interleaved chains (ILP 2)
            asm volatile("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3\n"
                         "v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3\n"
                         : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
        else if (KIND == 8)   // butterfly-like: outputs feed the next pair (add/sub of two registers, ILP 2 with swaps)
            asm volatile("v_add_f32 %2, %0, %1\n v_sub_f32 %3, %0, %1\n v_add_f32 %0, %2, %3\n v_sub_f32 %1, %2, %3\n"
                         "v_add_f32 %2, %0, %1\n v_sub_f32 %3, %0, %1\n v_add_f32 %0, %2, %3\n v_sub_f32 %1, %2, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + q0.x;
}

template <int KIND>
int run(const char* name, int wg_per_cu, float* out) {
    const int iters = 2000, nwg = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, out, 100, 1.0001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, out, iters, 1.0001f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_simd = (double)wg_per_cu * iters * 8 * 16;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instruction per SIMD\n", name, wg_per_cu, ms, ms * 1e6 / inst_per_simd);
    return 0;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int w : {4, 2}) {
#define R(K, N) if (w == 4) run<K>(N, 4, out); else run<K>(N, 2, out);
        R(0, "v_fmac_f32 x8 independent") R(1, "v_fmamk_f32 literal") R(2, "v_mul_f32_e64 neg") R(3, "v_fma_f32 sgpr operand")
        R(4, "v_pk_add_f32 (+neg)") R(5, "v_mov_b64") R(6, "dependent chain ILP1") R(7, "two chains ILP2") R(8, "add/sub butterfly ILP2")
    }
    return 0;
}
