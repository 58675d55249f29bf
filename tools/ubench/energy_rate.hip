// Energy price list.  The register kernels draw the board's full 1400 W on random data (sclk 1.6-2.1 GHz: tools/telemetry.py); round 3 measured
// how much of their time follows that clock (profiles/r03_limiter_families.txt: little for the headline kernel, most for rbig 4096 and the f64
// kernels) -- for those what an instruction costs in joules matters as much as its issue slots.
// Each variant runs one instruction form back to back on every SIMD (4 waves/SIMD, register operands changing every
// iteration -- not zeros) for `secs` seconds while tools/telemetry.py reads board power; it prints wave-instructions/s.
// energy per wave-instruction = (P_variant - P_idle_loop) / rate.
//
//   ./energy_rate.bin <variant> [secs]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096;

#define REP8(X) X X X X X X X X

template <int V>
__global__ __launch_bounds__(256, 4) void k(float* out, float seed) {
    __shared__ v2f lds[8 * 288];
    const int tid = threadIdx.x;
    float a0 = seed + tid * 0.37f, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float c = 0.99991f, d = 0.00013f * (1 + (tid & 7));
    const v2f pc = {c, 0.99987f}, pd = {d, d * 1.5f};
    v2f* slot = lds + tid;                              // consecutive 8-B slots: conflict-free ds_*_b64
    if (V == 8 || V == 9) slot[0] = p0;
    for (int it = 0; it < kIters; ++it) {
        if (V == 0) {          // v_fma_f32, 8 independent chains x 4
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (V == 1) {   // v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));)
        } else if (V == 2) {   // v_add_f32 (two-register form: a += d; a -= d' keeps values bounded and changing)
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_add_f32 %2, %2, %9\n v_sub_f32 %3, %3, %9\n"
                              "v_add_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_add_f32 %6, %6, %9\n v_sub_f32 %7, %7, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (V == 3) {   // v_pk_add_f32
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));)
        } else if (V == 4) {   // v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (V == 5) {   // v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));)
        } else if (V == 6) {   // v_mov_b32 (register copies)
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (V == 7) {   // s_nop loop: what a resident, clocked, idle-issuing chip draws
            REP8(asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n");)
        } else if (V == 8) {   // ds_write_b64 + ds_read_b64 pairs (8 + 8 per group), conflict-free
            for (int g = 0; g < 4; ++g) {
                typedef __attribute__((address_space(3))) volatile v2f lv;
                lv* s = (lv*)slot;
                s[0] = p0; s[288 * 1] = p1; s[288 * 2] = p2; s[288 * 3] = p3; s[288 * 4] = p4; s[288 * 5] = p5; s[288 * 6] = p6; s[288 * 7] = p7;
                p0 = s[288 * 7]; p1 = s[0]; p2 = s[288 * 1]; p3 = s[288 * 2]; p4 = s[288 * 3]; p5 = s[288 * 4]; p6 = s[288 * 5]; p7 = s[288 * 6];
                p0 += pd;
            }
        } else if (V == 9) {   // ds_read_b64 only
            for (int g = 0; g < 8; ++g) {
                typedef __attribute__((address_space(3))) volatile v2f lv;
                lv* s = (lv*)slot;
                p0 += s[0]; p1 += s[288 * 1]; p2 += s[288 * 2]; p3 += s[288 * 3]; p4 += s[288 * 4]; p5 += s[288 * 5]; p6 += s[288 * 6]; p7 += s[288 * 7];
            }
        } else if (V == 10) {  // v_mov_b32 with a DPP operand (quad_perm), the form the in-row reductions and transposes use
            REP8(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (V == 11) {  // v_permlane32_swap pairs (gfx950)
            REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %4, %6\n v_permlane16_swap_b32 %5, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
    }
    if (V == 1 || V == 3 || V == 5 || V == 8 || V == 9) { a0 = p0.x + p0.y + p1.x + p2.y + p3.x + p4.y + p5.x + p6.y + p7.x; }
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 123.456f) out[tid] = s;
}

template <int V>
double run(float* out, double secs, int per_iter) {
    const int grid = 256 * 4;   // 4 workgroups of 4 waves per CU = 4 waves/SIMD
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    while (el < secs) {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, 1.0f + launches + i);
        CHECK(hipDeviceSynchronize());
        launches += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const double winstr = (double)launches * grid * 4 * kIters * per_iter;
    printf("variant %d: %.3e wave-instr/s over %.2f s (%.2f ns per wave-instr per SIMD)\n", V, winstr / el, el, el / (winstr / 1024) * 1e9);
    return winstr / el;
}

int main(int argc, char** argv) {
    const int v = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 4.0;
    float* out; CHECK(hipMalloc(&out, 4096));
    switch (v) {
        case 0: run<0>(out, secs, 64); break;
        case 1: run<1>(out, secs, 64); break;
        case 2: run<2>(out, secs, 64); break;
        case 3: run<3>(out, secs, 64); break;
        case 4: run<4>(out, secs, 64); break;
        case 5: run<5>(out, secs, 64); break;
        case 6: run<6>(out, secs, 64); break;
        case 7: run<7>(out, secs, 64); break;
        case 8: run<8>(out, secs, 64); break;     // 32 writes + 32 reads
        case 9: run<9>(out, secs, 64); break;     // 64 reads
        case 10: run<10>(out, secs, 64); break;
        case 11: run<11>(out, secs, 64); break;
        default: printf("unknown variant\n"); return 1;
    }
    return 0;
}
