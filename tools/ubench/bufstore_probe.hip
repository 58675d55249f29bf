// Probe (design aid; the data operand of __builtin_amdgcn_raw_buffer_store_b32 is an INTEGER: a float must be bit-cast, or its value is converted): does buffer_store_dword through __builtin_amdgcn_make_buffer_rsrc write what a global store writes on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/bufstore_probe.hip -o tools/ubench/bufstore_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int FLAGS, bool UNIFORM>
__global__ void k(float* out, int row_len, int num_records) {
    const int lane = threadIdx.x & 63;
    float* row = out + (long)blockIdx.x * row_len;
    if (UNIFORM) {
        const long off = row - out;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)off), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)off >> 32));
        row = out + (long)(((unsigned long long)hi << 32) | lo);
    }
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(row, 0, num_records, FLAGS);
    for (int c = 0; c < 4; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 1000.0f * blockIdx.x + lane + 64 * c), r, 4 * lane + 256 * c, 0, 0);
}
template <int FLAGS, bool UNIFORM>
int run(const char* name, float* d, int row_len, int num_records) {
    CHECK(hipMemset(d, 0, 8 * row_len * 4));
    hipLaunchKernelGGL((k<FLAGS, UNIFORM>), dim3(8), dim3(64), 0, 0, d, row_len, num_records);
    CHECK(hipDeviceSynchronize());
    std::vector<float> h(8 * row_len);
    CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    int ok = 0, zero = 0;
    for (int b = 0; b < 8; ++b) for (int i = 0; i < 256; ++i) { const float v = h[b * row_len + i]; ok += v == 1000.0f * b + i; zero += v == 0.0f; }
    printf("%-46s num_records %10d: %4d of 2048 correct, %4d zero;  row 1: [0] %g [1] %g [2] %g [63] %g [64] %g [255] %g [256] %g\n", name, num_records, ok, zero,
           h[row_len], h[row_len + 1], h[row_len + 2], h[row_len + 63], h[row_len + 64], h[row_len + 255], h[row_len + 256]);
    return 0;
}
int main() {
    float* d; CHECK(hipMalloc(&d, 8 * 300 * 4));
    run<0x00020000, false>("flags 0x00020000, divergent pointer", d, 300, 256 * 4);
    run<0x00020000, true>("flags 0x00020000, readfirstlane'd pointer", d, 300, 256 * 4);
    run<0x00020000, true>("flags 0x00020000, uniform, records 200*4", d, 300, 200 * 4);
    run<0x00020000, true>("flags 0x00020000, uniform, records -1", d, 300, -1);
    run<0x00027000, true>("flags 0x00027000, uniform", d, 300, 256 * 4);
    run<0, true>("flags 0, uniform", d, 300, 256 * 4);
    return 0;
}
