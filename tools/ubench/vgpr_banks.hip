// Micro-benchmark: VGPR bank conflicts of VALU source operands on gfx950 (design aid).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/vgpr_banks.hip -o tools/ubench/vgpr_banks.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(s) s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s "\n"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119"

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 0.5\n v_mov_b32 v102, 0.25\n v_mov_b32 v103, 2.0\n v_mov_b32 v104, 1.0\n v_mov_b32 v105, 0.5\n"
                 "v_mov_b32 v106, 0.25\n v_mov_b32 v107, 2.0\n v_mov_b32 v108, 1.0\n v_mov_b32 v109, 0.5\n v_mov_b32 v110, 0.25\n v_mov_b32 v111, 2.0\n"
                 "v_mov_b32 v112, 1.0\n v_mov_b32 v113, 0.5\n v_mov_b32 v114, 0.25\n v_mov_b32 v115, 2.0\n v_mov_b32 v116, 1.0\n v_mov_b32 v117, 0.5\n v_mov_b32 v118, 0.25\n v_mov_b32 v119, 2.0" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) asm volatile(REP8("v_fmac_f32 v100, v101, v102") ::: CLOB);    // dst b0, src b1, b2
            if (KIND == 1) asm volatile(REP8("v_fmac_f32 v100, v104, v102") ::: CLOB);    // src0 in dst's bank
            if (KIND == 2) asm volatile(REP8("v_fmac_f32 v100, v101, v105") ::: CLOB);    // src0 and src1 same bank
            if (KIND == 3) asm volatile(REP8("v_fmac_f32 v100, v104, v108") ::: CLOB);    // all three bank 0
            if (KIND == 4) asm volatile(REP8("v_add_f32 v100, v101, v102") ::: CLOB);     // two sources, banks 1, 2
            if (KIND == 5) asm volatile(REP8("v_add_f32 v100, v101, v105") ::: CLOB);     // two sources, same bank
            if (KIND == 6) asm volatile(REP8("v_fma_f32 v100, v101, v102, v103") ::: CLOB);   // 3 sources, banks 1, 2, 3
            if (KIND == 7) asm volatile(REP8("v_fma_f32 v100, v101, v102, v105") ::: CLOB);   // src0 and src2 same bank
            if (KIND == 8) asm volatile(REP8("v_fma_f32 v100, v101, v105, v109") ::: CLOB);   // all sources bank 1
            if (KIND == 9) asm volatile("v_fmac_f32 v100, v101, v102\n v_fmac_f32 v104, v105, v106\n v_fmac_f32 v108, v109, v110\n v_fmac_f32 v112, v113, v114\n"
                                        "v_fmac_f32 v101, v102, v103\n v_fmac_f32 v105, v106, v107\n v_fmac_f32 v109, v110, v111\n v_fmac_f32 v113, v114, v115" ::: CLOB);  // independent, conflict-free
            if (KIND == 10) asm volatile(REP8("v_fma_f32 v100, s4, v102, v103") ::: CLOB);    // SGPR source
            if (KIND == 11) asm volatile(REP8("v_pk_add_f32 v[100:101], v[102:103], v[104:105]") ::: CLOB);
            if (KIND == 12) asm volatile(REP8("v_pk_add_f32 v[100:101], v[102:103], v[106:107]") ::: CLOB);
            if (KIND == 13) asm volatile(REP8("v_mov_b64 v[100:101], v[102:103]") ::: CLOB);
            if (KIND == 14) asm volatile(REP8("v_fmamk_f32 v100, v101, 0x3f3504f3, v102") ::: CLOB);
            if (KIND == 15) asm volatile(REP8("v_fmamk_f32 v100, v101, 0x3f3504f3, v105") ::: CLOB);   // src0 and src2 same bank
        }
    }
    float r;
    asm volatile("v_mov_b32 %0, v100" : "=v"(r) :: CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
int run(const char* name, float* out) {
    const int iters = 2000, wg_per_cu = 4, nwg = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, out, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s %.3f ms  %.2f ns per wave-instruction per SIMD (4 waves/SIMD)\n", name, ms, ms * 1e6 / ((double)wg_per_cu * iters * 8 * 16));
    return 0;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>("fmac dst b0, src b1 b2", out); run<1>("fmac src0 in dst's bank", out); run<2>("fmac src0 src1 same bank", out);
    run<3>("fmac all bank 0", out); run<4>("add src b1 b2", out); run<5>("add srcs same bank", out);
    run<6>("fma srcs b1 b2 b3", out); run<7>("fma src0 src2 same bank", out); run<8>("fma all srcs bank 1", out);
    run<9>("fmac x8 independent conflict-free", out); run<10>("fma sgpr, v, v", out);
    run<11>("pk_add srcs pairs (2,3) (0,1)", out); run<12>("pk_add srcs pairs (2,3) (2,3)", out); run<13>("mov_b64", out);
    run<14>("fmamk srcs b1 b2", out); run<15>("fmamk srcs same bank", out);
    return 0;
}
