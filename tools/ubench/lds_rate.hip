// Micro-benchmark: LDS issue rate of ds_read/ds_write b32/b64/b128 on gfx950 (design aid, not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_rate.hip -o tools/ubench/lds_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// KIND: 0 read b32, 1 read b64, 2 read b128, 3 write b32, 4 write b64, 5 write b128, 6 read b64 stride-72 rows (the FFT transpose read)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    const unsigned base = wave * 4096 * 0 + (KIND == 0 || KIND == 3 ? lane * 4 : KIND == 1 || KIND == 4 || KIND == 6 ? lane * 8 : lane * 16);
    float acc = 0.f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    typedef float v4 __attribute__((ext_vector_type(4)));
    v2 d2 = {1.f, 2.f}; v4 d4 = {1.f, 2.f, 3.f, 4.f}; float d1 = 1.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2\n s_waitcnt lgkmcnt(4)" : "=v"(r) : "v"(base), "n"(u * 256)); acc += r; }
            if (KIND == 1) { v2 r; asm volatile("ds_read_b64 %0, %1 offset:%2\n s_waitcnt lgkmcnt(4)" : "=v"(r) : "v"(base), "n"(u * 512)); acc += r.x; }
            if (KIND == 6) { v2 r; asm volatile("ds_read_b64 %0, %1 offset:%2\n s_waitcnt lgkmcnt(4)" : "=v"(r) : "v"(base), "n"(u * 576)); acc += r.x; }
            if (KIND == 2) { v4 r; asm volatile("ds_read_b128 %0, %1 offset:%2\n s_waitcnt lgkmcnt(4)" : "=v"(r) : "v"(base), "n"(u * 1024)); acc += r.x; }
            if (KIND == 3) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(base), "v"(d1), "n"(u * 256));
            if (KIND == 4) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(base), "v"(d2), "n"(u * 512));
            if (KIND == 5) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(base), "v"(d4), "n"(u * 1024));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND>
int run(const char* name, int bytes_per_lane, int wg_per_cu, float* out) {
    const int iters = 4000, nwg = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 16384 * 2, 0, out, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 16384 * 2, 0, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops_per_cu = (double)wg_per_cu * 4 * iters * 8;          // wave-level DS instructions per CU
    const double ns = ms * 1e6 / ops_per_cu;
    printf("%-22s waves/CU=%2d  %.3f ms  %.2f ns per wave-op per CU  -> %.0f B/ns/CU (%.1f B/clk @2.4GHz)\n", name, wg_per_cu * 4, ms, ns,
           64.0 * bytes_per_lane / ns, 64.0 * bytes_per_lane / ns / 2.4);
    return 0;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int w : {1, 2, 4}) {
        run<0>("ds_read_b32", 4, w, out); run<1>("ds_read_b64", 8, w, out); run<6>("ds_read_b64 stride 576B", 8, w, out); run<2>("ds_read_b128", 16, w, out);
        run<3>("ds_write_b32", 4, w, out); run<4>("ds_write_b64", 8, w, out); run<5>("ds_write_b128", 16, w, out);
    }
    return 0;
}
