// Micro-benchmark: cost of the r8x3 output pattern (513-float rows, dword stores) vs aligned dwordx4 streaming.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NF = 119808, NB = 513, HOP = 256;

// A: one wave per 16 consecutive rows; per row 2 dwordx2-ish loads (256 floats) + 9 dword stores like r8x3
template <int NT>
__global__ __launch_bounds__(256) void rows_dword(const float* x, float* out) {
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk * 16 >= NF) return;
    for (int f = chunk * 16; f < chunk * 16 + 16; ++f) {
        const float2 v = *reinterpret_cast<const float2*>(x + (size_t)f * HOP + 2 * lane);
        const float2 w = *reinterpret_cast<const float2*>(x + (size_t)f * HOP + 128 + 2 * lane);
        float* row = out + (size_t)f * NB;
        const float s = v.x + v.y + w.x + w.y;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (NT) { __builtin_nontemporal_store(s + m, row + lane + 64 * m); __builtin_nontemporal_store(s - m, row + 512 - lane - 64 * m); }
            else { row[lane + 64 * m] = s + m; row[512 - lane - 64 * m] = s - m; }
        }
        if (lane == 0) row[256] = s;
    }
}

// B: same bytes, flat aligned float4 streaming (each wave: 16 rows = 8208 floats = 2052 float4)
template <int NT>
__global__ __launch_bounds__(256) void flat_x4(const float* x, float* out) {
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk * 16 >= NF) return;
    float acc = 0.f;
    for (int i = 0; i < 16; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)(chunk * 16 + i) * HOP + 4 * lane);
        acc += v.x + v.y + v.z + v.w;
    }
    float4* o = reinterpret_cast<float4*>(out + (size_t)chunk * 16 * NB);
    for (int i = lane; i < 2052; i += 64) {
        float4 v = make_float4(acc, acc + 1, acc + 2, acc + i);
        if (NT) __builtin_nontemporal_store(v.x, &o[i].x), __builtin_nontemporal_store(v.y, &o[i].y), __builtin_nontemporal_store(v.z, &o[i].z), __builtin_nontemporal_store(v.w, &o[i].w);
        else o[i] = v;
    }
}

template <typename K>
int timeit(const char* name, K kern, float** xs, float** outs) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nwg = (NF / 16 + 3) / 4;
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, xs[i], outs[i]);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int iters = 40;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, xs[i % 4], outs[i % 4]);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)NF * (HOP + NB) * 4;
    printf("%-22s %.1f us/launch  %.0f GB/s\n", name, ms * 1e3 / iters, bytes / (ms / iters * 1e-3) / 1e9);
    return 0;
}

int main() {
    float *xs[4], *outs[4];
    for (int i = 0; i < 4; ++i) { CHECK(hipMalloc(&xs[i], (size_t)64 * 480000 * 4)); CHECK(hipMalloc(&outs[i], (size_t)NF * NB * 4 + 4096)); CHECK(hipMemset(xs[i], 0, (size_t)64 * 480000 * 4)); }
    timeit("rows dword", rows_dword<0>, xs, outs);
    timeit("rows dword nt", rows_dword<1>, xs, outs);
    timeit("flat float4", flat_x4<0>, xs, outs);
    timeit("flat float4 nt(scalar)", flat_x4<1>, xs, outs);
    return 0;
}
