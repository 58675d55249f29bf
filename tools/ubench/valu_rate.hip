// Micro-benchmark: issue rate of plain vs packed f32 VALU on gfx950 (design aid, not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    const long long c0 = clock64();
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    v2 pb = {b, b}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (KIND == 0) {   // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 1) {   // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
        } else if (KIND == 2) {   // 8 independent v_add_f32
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else {                  // 8 independent v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
        }
      }
    }
    if (cyc && blockIdx.x == 0 && threadIdx.x == 0) *cyc = clock64() - c0;     // shader-clock cycles of one wave
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int KIND>
int run(const char* name, int wg_per_cu, float* out) {
    const int iters = 2000, nwg = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, out, 100, (long long*)nullptr);
    long long* cyc; CHECK(hipHostMalloc(&cyc, 8)); *cyc = 0;
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, out, iters, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // instructions per SIMD = wg_per_cu waves * iters * 8
    double inst_per_simd = (double)wg_per_cu * iters * 8 * 16;
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD; wave 0 ran %lld shader cycles = %.2f cycles per SIMD instr, clock %.2f GHz\n",
           name, wg_per_cu, ms, ms * 1e6 / inst_per_simd, *cyc, (double)*cyc / inst_per_simd, *cyc / (ms * 1e6));
    return 0;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int w : {1, 2, 4, 8}) {
        if (w == 1) { run<0>("v_fma_f32", 1, out); run<1>("v_pk_fma_f32", 1, out); run<2>("v_add_f32", 1, out); run<3>("v_pk_add_f32", 1, out); }
        if (w == 2) { run<0>("v_fma_f32", 2, out); run<1>("v_pk_fma_f32", 2, out); run<2>("v_add_f32", 2, out); run<3>("v_pk_add_f32", 2, out); }
        if (w == 4) { run<0>("v_fma_f32", 4, out); run<1>("v_pk_fma_f32", 4, out); run<2>("v_add_f32", 4, out); run<3>("v_pk_add_f32", 4, out); }
        if (w == 8) { run<0>("v_fma_f32", 8, out); run<1>("v_pk_fma_f32", 8, out); run<2>("v_add_f32", 8, out); run<3>("v_pk_add_f32", 8, out); }
    }
    return 0;
}
