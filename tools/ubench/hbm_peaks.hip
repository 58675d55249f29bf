// HBM practical peaks on this device: read-only, write-only, copy (float4 per lane, grid-stride).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, int NT>   // 0 read, 1 write, 2 copy, 3 = read 1 : write 2 (like the STFT stream)
__global__ __launch_bounds__(256) void k(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n, float* sink) {
    v4f acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (MODE == 0) { acc += in[i]; }
        else if (MODE == 1) { v4f v = {1.f, 2.f, 3.f, (float)i}; if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
        else if (MODE == 2) { v4f v = in[i]; if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
        else { v4f v = in[i]; if (NT) { __builtin_nontemporal_store(v, out + 2 * i); __builtin_nontemporal_store(v, out + 2 * i + 1); } else { out[2 * i] = v; out[2 * i + 1] = v; } }
    }
    if (MODE == 0 && acc.x == 123.456f) *sink = acc.y;
}

template <int MODE, int NT>
int run(const char* name, v4f* a, v4f* b, size_t n, double bytes, float* sink) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(256), 0, 0, a, b, n, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(256), 0, 0, a, b, n, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %.2f TB/s\n", name, bytes * 5 / (ms * 1e-3) / 1e12);
    return 0;
}

// 4-byte-per-lane streaming (what a row-per-wave epilogue emits) for comparison with the 16-byte forms above
template <int MODE>
__global__ __launch_bounds__(256) void k1(const float* __restrict__ in, float* __restrict__ out, size_t n, float* sink) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (MODE == 0) acc += in[i];
        else if (MODE == 1) out[i] = (float)i;
        else { const float v = in[i]; out[2 * i] = v; out[2 * i + 1] = v; }
    }
    if (MODE == 0 && acc == 123.456f) *sink = acc;
}
template <int MODE>
int run1(const char* name, float* a, float* b, size_t n, double bytes, float* sink) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    hipLaunchKernelGGL((k1<MODE>), dim3(grid), dim3(256), 0, 0, a, b, n, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k1<MODE>), dim3(grid), dim3(256), 0, 0, a, b, n, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %.2f TB/s\n", name, bytes * 5 / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    const size_t bytes = (size_t)2 << 30;      // 2 GiB per buffer
    v4f *a, *b; float* sink;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, 2 * bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, 2 * bytes));
    const size_t n = bytes / 16;
    run<0, 0>("read", a, b, n, (double)bytes, sink);
    run<1, 0>("write", a, b, n, (double)bytes, sink);
    run<1, 1>("write nt", a, b, n, (double)bytes, sink);
    run<2, 0>("copy (r+w bytes)", a, b, n, 2.0 * bytes, sink);
    run<2, 1>("copy nt", a, b, n, 2.0 * bytes, sink);
    run<3, 0>("read1:write2", a, b, n, 3.0 * bytes, sink);
    run<3, 1>("read1:write2 nt", a, b, n, 3.0 * bytes, sink);
    run1<0>("read dword", (float*)a, (float*)b, bytes / 4, (double)bytes, sink);
    run1<1>("write dword", (float*)a, (float*)b, bytes / 4, (double)bytes, sink);
    run1<2>("read1:write2 dword", (float*)a, (float*)b, bytes / 4, 3.0 * bytes, sink);
    return 0;
}
