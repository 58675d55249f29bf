// Model of the r8x3 kernel's memory/VALU interplay (design aid): per frame 2 dwordx2 loads, NV independent FMAs,
// 2052 B of stores in several shapes, at a chosen occupancy.  Prints us/launch for the cfg2 shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NF = 119808, NB = 513, HOP = 256;

template <int NV>
__device__ __forceinline__ void valu_block(float (&a)[8], float b, float c) {
#pragma unroll
    for (int i = 0; i < NV / 8; ++i)
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));
}

// SHAPE 0: 9 dword stores in a burst; 1: 9 dword stores spread between VALU blocks; 2: 2 dwordx4 + 1 dword burst; 3: no stores
template <int NV, int SHAPE>
__global__ __launch_bounds__(256) void model(const float* x, float* out, int n_waves, int lds_pad) {
    extern __shared__ float pad[];
    if (lds_pad < 0) pad[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const long g0 = (long)NF * w / n_waves, g1 = (long)NF * (w + 1) / n_waves;
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = lane + i;
    float2 v = *reinterpret_cast<const float2*>(x + g0 * HOP + 2 * lane);
    float2 u = *reinterpret_cast<const float2*>(x + g0 * HOP + 128 + 2 * lane);
    for (long f = g0; f < g1; ++f) {
        const float s = v.x + v.y + u.x + u.y;
        if (f + 1 < g1) {
            v = *reinterpret_cast<const float2*>(x + (f + 1) * HOP + 2 * lane);
            u = *reinterpret_cast<const float2*>(x + (f + 1) * HOP + 128 + 2 * lane);
        }
        a[0] += s;
        float* row = out + f * NB;
        if (SHAPE == 1) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                valu_block<NV / 4>(a, 1.0001f, 0.5f);
                __builtin_nontemporal_store(a[m], row + lane + 64 * m);
                __builtin_nontemporal_store(a[m + 4], row + 512 - lane - 64 * m);
            }
            __builtin_nontemporal_store(a[0], row + 256);
        } else {
            valu_block<NV>(a, 1.0001f, 0.5f);
            if (SHAPE == 0) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    __builtin_nontemporal_store(a[m], row + lane + 64 * m);
                    __builtin_nontemporal_store(a[m + 4], row + 512 - lane - 64 * m);
                }
                __builtin_nontemporal_store(a[0], row + 256);
            } else if (SHAPE == 2) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(v4f{a[0], a[1], a[2], a[3]}, reinterpret_cast<v4f*>(row + 1 + 4 * lane));
                __builtin_nontemporal_store(v4f{a[4], a[5], a[6], a[7]}, reinterpret_cast<v4f*>(row + 257 + 4 * lane));
                __builtin_nontemporal_store(a[0], row);
            } else {
                if (a[0] == 123.456f) row[lane] = a[1];
            }
        }
    }
}

// Sweep: wave w takes frames w*R.., then jumps by n_waves*R: all waves write one contiguous window at any time
template <int NV, int R>
__global__ __launch_bounds__(256) void model_sweep(const float* x, float* out, int n_waves, int lds_pad) {
    extern __shared__ float pad[];
    if (lds_pad < 0) pad[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = lane + i;
    for (long fb = (long)w * R; fb < NF; fb += (long)n_waves * R) {
        float2 v = *reinterpret_cast<const float2*>(x + fb * HOP + 2 * lane);
        float2 u = *reinterpret_cast<const float2*>(x + fb * HOP + 128 + 2 * lane);
#pragma unroll 1
        for (int j = 0; j < R; ++j) {
            const long f = fb + j;
            if (f >= NF) break;
            const float s = v.x + v.y + u.x + u.y;
            if (j + 1 < R && f + 1 < NF) {
                v = *reinterpret_cast<const float2*>(x + (f + 1) * HOP + 2 * lane);
                u = *reinterpret_cast<const float2*>(x + (f + 1) * HOP + 128 + 2 * lane);
            }
            a[0] += s;
            valu_block<NV>(a, 1.0001f, 0.5f);
            float* row = out + f * NB;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                __builtin_nontemporal_store(a[m], row + lane + 64 * m);
                __builtin_nontemporal_store(a[m + 4], row + 512 - lane - 64 * m);
            }
            __builtin_nontemporal_store(a[0], row + 256);
        }
    }
}

template <int NV, int R>
int timeit_sweep(const char* name, int occ, float** xs, float** outs) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int n_waves = 256 * 4 * occ, nwg = n_waves / 4;
    const int lds = 160 * 1024 / occ - 1024;
    auto kern = model_sweep<NV, R>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, xs[i], outs[i], n_waves, 0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, xs[i % 4], outs[i % 4], n_waves, 0);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s occ=%d  %.1f us/launch\n", name, occ, ms * 1e3 / iters);
    return 0;
}

// Deferred bursts: keep G frames of results in registers, then store G rows back to back (G*2052 contiguous bytes)
template <int NV, int G>
__global__ __launch_bounds__(256) void model_burst(const float* x, float* out, int n_waves, int lds_pad) {
    extern __shared__ float pad[];
    if (lds_pad < 0) pad[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const long g0 = (long)NF * w / n_waves, g1 = (long)NF * (w + 1) / n_waves;
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = lane + i;
    float2 v = *reinterpret_cast<const float2*>(x + g0 * HOP + 2 * lane);
    float2 u = *reinterpret_cast<const float2*>(x + g0 * HOP + 128 + 2 * lane);
    for (long fb = g0; fb < g1; fb += G) {
        float keep[G][9];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const long f = fb + j;
            const float s = v.x + v.y + u.x + u.y;
            if (f + 1 < g1) {
                v = *reinterpret_cast<const float2*>(x + (f + 1) * HOP + 2 * lane);
                u = *reinterpret_cast<const float2*>(x + (f + 1) * HOP + 128 + 2 * lane);
            }
            a[0] += s;
            valu_block<NV>(a, 1.0001f, 0.5f);
#pragma unroll
            for (int m = 0; m < 8; ++m) keep[j][m] = a[m];
            keep[j][8] = a[0] + a[1];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const long f = fb + j;
            if (f < g1) {
                float* row = out + f * NB;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    __builtin_nontemporal_store(keep[j][m], row + lane + 64 * m);
                    __builtin_nontemporal_store(keep[j][m + 4], row + 512 - lane - 64 * m);
                }
                __builtin_nontemporal_store(keep[j][8], row + 256);
            }
        }
    }
}

template <int NV, int G>
int timeit_burst(const char* name, int occ, float** xs, float** outs) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int n_waves = 256 * 4 * occ, nwg = n_waves / 4;
    const int lds = 160 * 1024 / occ - 1024;
    auto kern = model_burst<NV, G>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, xs[i], outs[i], n_waves, 0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, xs[i % 4], outs[i % 4], n_waves, 0);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s occ=%d  %.1f us/launch\n", name, occ, ms * 1e3 / iters);
    return 0;
}

// WG-interleaved: the 4 waves of a workgroup take consecutive frames (wave i -> frame base + 4*s + i), so the
// workgroup as a whole streams 4 KB in / 8 KB out per step.  SYNC: s_barrier per step keeps them together.
template <int NV, int SYNC, int NLOAD>
__global__ __launch_bounds__(256) void model_wg(const float* x, float* out, int n_wgs, int lds_pad) {
    extern __shared__ float pad[];
    if (lds_pad < 0) pad[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wg = blockIdx.x;
    const long g0 = (long)NF * wg / n_wgs, g1 = (long)NF * (wg + 1) / n_wgs;
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = lane + i;
    for (long fb = g0; fb < g1; fb += 4) {
        const long f = fb + wv;
        const bool act = f < g1;
        if (act) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) { const float2 v = *reinterpret_cast<const float2*>(x + f * HOP + 128 * k + 2 * lane); s += v.x + v.y; }
            a[0] += s;
            valu_block<NV>(a, 1.0001f, 0.5f);
            float* row = out + f * NB;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                __builtin_nontemporal_store(a[m], row + lane + 64 * m);
                __builtin_nontemporal_store(a[m + 4], row + 512 - lane - 64 * m);
            }
            __builtin_nontemporal_store(a[0], row + 256);
        }
        if (SYNC) __syncthreads();
    }
}

template <int NV, int SYNC, int NLOAD>
int timeit_wg(const char* name, int occ, float** xs, float** outs) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nwg = 256 * occ;
    const int lds = 160 * 1024 / occ - 1024;
    auto kern = model_wg<NV, SYNC, NLOAD>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, xs[i], outs[i], nwg, 0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, xs[i % 4], outs[i % 4], nwg, 0);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s occ=%d  %.1f us/launch\n", name, occ, ms * 1e3 / iters);
    return 0;
}

template <int NV, int SHAPE>
int timeit(const char* name, int occ, float** xs, float** outs) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int n_waves = 256 * 4 * occ, nwg = n_waves / 4;
    const int lds = 160 * 1024 / occ - 1024;    // forces exactly `occ` workgroups per CU
    auto kern = model<NV, SHAPE>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, xs[i], outs[i], n_waves, 0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, xs[i % 4], outs[i % 4], n_waves, 0);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s occ=%d  %.1f us/launch\n", name, occ, ms * 1e3 / iters);
    return 0;
}

int main() {
    float *xs[4], *outs[4];
    for (int i = 0; i < 4; ++i) { CHECK(hipMalloc(&xs[i], (size_t)64 * 480000 * 4)); CHECK(hipMalloc(&outs[i], (size_t)NF * NB * 4 + 8192)); CHECK(hipMemset(xs[i], 0, (size_t)64 * 480000 * 4)); }
    for (int occ : {5, 8}) {
        timeit<0, 0>("NV=0   burst 9xdword", occ, xs, outs);
        timeit<344, 3>("NV=344 no stores", occ, xs, outs);
        timeit<344, 0>("NV=344 burst 9xdword", occ, xs, outs);
        timeit<344, 1>("NV=344 spread 9xdword", occ, xs, outs);
        timeit<344, 2>("NV=344 burst 2xdwordx4+1", occ, xs, outs);
        timeit_sweep<0, 1>("NV=0 sweep R=1", occ, xs, outs);
        timeit_sweep<0, 2>("NV=0 sweep R=2", occ, xs, outs);
        timeit_sweep<0, 4>("NV=0 sweep R=4", occ, xs, outs);
        timeit_sweep<0, 8>("NV=0 sweep R=8", occ, xs, outs);
        timeit_sweep<344, 4>("NV=344 sweep R=4", occ, xs, outs);
        timeit_sweep<344, 8>("NV=344 sweep R=8", occ, xs, outs);
        timeit_burst<0, 2>("NV=0 deferred burst G=2", occ, xs, outs);
        timeit_burst<0, 4>("NV=0 deferred burst G=4", occ, xs, outs);
        timeit_burst<0, 8>("NV=0 deferred burst G=8", occ, xs, outs);
        timeit_burst<344, 2>("NV=344 deferred burst G=2", occ, xs, outs);
        timeit_burst<344, 4>("NV=344 deferred burst G=4", occ, xs, outs);
        timeit_burst<344, 8>("NV=344 deferred burst G=8", occ, xs, outs);
    }
    return 0;
}
