// satomic.hip -- do scalar-memory atomics (s_atomic_add, lgkmcnt-tracked: no wait behind a wave's vector stores) work on gfx950,
// are their tickets unique across XCDs, and what do they cost?   hipcc --offload-arch=gfx950 -O3 satomic.hip -o satomic.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// mode 0: scalar atomic; mode 1: vector atomic (agent scope) from lane 0
template <int MODE>
__global__ __launch_bounds__(256) void draw(unsigned* ctr, int n_ctr, int stride, int per_wave, unsigned* tickets, unsigned long long* ticks,
                                            float* sink, int store_rows) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    unsigned* my = ctr + (n_ctr > 1 ? (blockIdx.x % n_ctr) * stride : 0);
    unsigned long long t_acc = 0;
    for (int i = 0; i < per_wave; ++i) {
        // background: a few vector stores, as the STFT frame loop would have in flight
        for (int r = 0; r < store_rows; ++r) sink[(static_cast<size_t>(wave) * per_wave + i) * 64 * store_rows + r * 64 + lane] = 1.0f;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned got;
        if (MODE == 0) {
            unsigned v = 1;
            asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(my) : "memory");
            got = v;
        } else {
            unsigned v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            got = __builtin_amdgcn_readfirstlane(v);
        }
        t_acc += __builtin_amdgcn_s_memrealtime() - t0 + (got & 0);
        if (lane == 0) tickets[static_cast<size_t>(wave) * per_wave + i] = got;
    }
    if (lane == 0) ticks[wave] = t_acc;
}

template <int MODE>
void run(const char* name, unsigned* ctr, int n_ctr, int stride, int n_wg, int per_wave, int store_rows) {
    const int n_waves = n_wg * 4;
    unsigned* tickets; unsigned long long* ticks; float* sink;
    CK(hipMalloc(&tickets, sizeof(unsigned) * n_waves * per_wave));
    CK(hipMalloc(&ticks, sizeof(unsigned long long) * n_waves));
    CK(hipMalloc(&sink, sizeof(float) * 64 * std::max(store_rows, 1) * n_waves * per_wave));
    CK(hipMemset(ctr, 0, sizeof(unsigned) * stride * std::max(n_ctr, 1)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(draw<MODE>, dim3(n_wg), dim3(256), 0, 0, ctr, n_ctr, stride, per_wave, tickets, ticks, sink, store_rows);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> h(static_cast<size_t>(n_waves) * per_wave); std::vector<unsigned long long> ht(n_waves); std::vector<unsigned> hc(stride * std::max(n_ctr, 1));
    CK(hipMemcpy(h.data(), tickets, h.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ht.data(), ticks, ht.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), ctr, hc.size() * 4, hipMemcpyDeviceToHost));
    // uniqueness: per counter, the tickets must be exactly 0..count-1
    long long total = 0; for (int c = 0; c < std::max(n_ctr, 1); ++c) total += hc[c * stride];
    bool unique = true;
    if (n_ctr <= 1) { std::sort(h.begin(), h.end()); for (size_t i = 0; i < h.size(); ++i) if (h[i] != i) { unique = false; break; } }
    double tsum = 0; for (auto v : ht) tsum += v;
    printf("%-44s waves %5d x %3d draws, %2d counters: kernel %8.1f us, wait per draw %7.3f us, final sum %lld (want %lld)%s\n", name, n_waves, per_wave,
           std::max(n_ctr, 1), ms * 1e3, tsum * 0.01 / (static_cast<double>(n_waves) * per_wave), total, static_cast<long long>(n_waves) * per_wave,
           n_ctr <= 1 ? (unique ? ", tickets unique" : ", TICKETS NOT UNIQUE") : "");
    CK(hipFree(tickets)); CK(hipFree(ticks)); CK(hipFree(sink));
}

int main(int argc, char** argv) {
    const bool scalar = argc > 1 && atoi(argv[1]) == 1;
    unsigned *plain, *uncached;
    const int stride = 1088;
    CK(hipMalloc(&plain, sizeof(unsigned) * stride * 256));
    CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&uncached), sizeof(unsigned) * stride * 256, hipDeviceMallocUncached));
    run<1>("vector, 1 counter, idle", plain, 1, stride, 64, 4, 0);
    run<1>("vector, 1 counter, 3072 waves", plain, 1, stride, 768, 8, 0);
    run<1>("vector, 1 counter, 3072 waves, 9 stores", plain, 1, stride, 768, 8, 9);
    run<1>("vector, 256 counters, 3072 waves, 9 stores", plain, 256, stride, 768, 8, 9);
    run<1>("vector, 32 counters, 3072 waves, 9 stores", plain, 32, stride, 768, 8, 9);
    if (scalar) {
        run<0>("scalar, 1 counter, idle", plain, 1, stride, 64, 4, 0);
        run<0>("scalar, 1 counter, 3072 waves", plain, 1, stride, 768, 8, 0);
        run<0>("scalar, 1 counter UNCACHED mem, 3072 waves", uncached, 1, stride, 768, 8, 0);
        run<0>("scalar, 1 counter, 3072 waves, 9 stores", plain, 1, stride, 768, 8, 9);
        run<0>("scalar, 256 counters, 3072 waves, 9 stores", plain, 256, stride, 768, 8, 9);
        run<0>("scalar, 8 counters (b%8), 3072 waves, 9 stores", plain, 8, stride, 768, 8, 9);
    }
    return 0;
}
