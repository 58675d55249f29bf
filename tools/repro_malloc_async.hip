// repro_malloc_async.hip -- HIP only, no libspectro: does a block from hipMallocAsync carry another allocation's data while it is in use?
// (round-2 finding: ~8 % of int16 batch calls read foreign data when their float workspace came from hipMallocAsync / hipFreeAsync;
//  a per-stream hipMalloc block: none.  VERDICT r02 item 8 / ADVICE r02: settle whether that was the runtime or this library.)
//
// What the library did per call, on ONE stream (the default one in the failing runs):
//     hipMallocAsync(work) -> convert kernel (int16 src -> float work) -> transform kernel (reads work, writes out) -> hipFreeAsync(work)
// while the Python shim around it allocated / freed the call's input and output with plain hipMalloc / hipFree (through a
// size-bucketed pool: a freed block is handed to the next caller without hipFree) and copied with hipMemcpy.  This program replays
// that sequence with a transform that CHECKS every element of `work` against the source, for 3 000 calls of random sizes.
//     hipcc --offload-arch=gfx950 -O2 tools/repro_malloc_async.hip -o tools/repro_malloc_async.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

__global__ void convert(const int16_t* src, float* dst, int64_t n) {
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) dst[i] = static_cast<float>(src[i]);
}
// the "transform": out[i] = work[i] * 2 and a count of elements of work that are not what convert() must have left there
__global__ void check(const int16_t* src, const float* work, float* out, int64_t n, unsigned long long* bad, long long* first_bad) {
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const float v = work[i];
        if (v != static_cast<float>(src[i])) { atomicAdd(bad, 1ull); atomicMin(reinterpret_cast<unsigned long long*>(first_bad), static_cast<unsigned long long>(i)); }
        out[i] = v * 2.0f;
    }
}
__global__ void scribble(float* p, int64_t n, float v) {     // what OTHER allocations hold: a value no int16 converts to
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) p[i] = v;
}

int main(int argc, char** argv) {
    const int calls = argc > 1 ? atoi(argv[1]) : 3000;
    const bool other_stream = argc > 2 && atoi(argv[2]) == 1;     // 1: a created stream instead of the default one
    // 0: as the library did (hipMallocAsync / hipFreeAsync around the call); 1: plain hipMalloc / hipFree for the work block instead;
    // 2: as 0 with the upload's source in pinned host memory; 3: as 0 with hipStreamSynchronize between the upload and hipMallocAsync;
    // 4: as 0 with the stream-ordered allocation made BEFORE the upload is queued
    const int variant = argc > 3 ? atoi(argv[3]) : 0;
    hipStream_t s = nullptr;
    if (other_stream) CK(hipStreamCreate(&s));
    std::mt19937_64 rng(12345);
    unsigned long long* bad; long long* first_bad;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first_bad, 8));
    std::vector<std::pair<void*, size_t>> pool;                   // the shim's pool: parked blocks, reused by capacity
    auto pool_get = [&](size_t bytes) -> void* {
        for (size_t i = 0; i < pool.size(); ++i) if (pool[i].second == bytes) { void* p = pool[i].first; pool.erase(pool.begin() + i); return p; }
        void* p; CK(hipMalloc(&p, bytes)); return p;
    };
    auto bucket = [](size_t n) { return (n + (1u << 20) - 1) & ~size_t((1u << 20) - 1); };
    long long failures = 0; unsigned long long total_bad = 0;
    std::vector<int16_t> host;
    int16_t* pinned = nullptr;
    if (variant == 2) CK(hipHostMalloc(reinterpret_cast<void**>(&pinned), 6400000 * 2));
    for (int c = 0; c < calls; ++c) {
        const int64_t n = 300000 + static_cast<int64_t>(rng() % 6000000);          // 0.3 .. 6.3 M samples (1.2 .. 25 MB of floats)
        host.resize(n);
        for (int64_t i = 0; i < n; i += 97) host[i] = static_cast<int16_t>(rng());
        host[0] = static_cast<int16_t>(c);
        const size_t in_b = bucket(n * 2), out_b = bucket(n * 4);
        int16_t* d_in = static_cast<int16_t*>(pool_get(in_b));
        float* d_out = static_cast<float*>(pool_get(out_b));
        float* work = nullptr;
        if (variant == 4) CK(hipMallocAsync(reinterpret_cast<void**>(&work), n * sizeof(float), s));
        if (variant == 2) { memcpy(pinned, host.data(), n * 2); CK(hipMemcpyAsync(d_in, pinned, n * 2, hipMemcpyHostToDevice, s)); }
        else CK(hipMemcpyAsync(d_in, host.data(), n * 2, hipMemcpyHostToDevice, s));
        CK(hipMemsetAsync(bad, 0, 8, s)); CK(hipMemsetAsync(first_bad, 0x7f, 8, s));
        if (variant == 3) CK(hipStreamSynchronize(s));
        if (variant == 1) CK(hipMalloc(reinterpret_cast<void**>(&work), n * sizeof(float)));
        else if (variant != 4) CK(hipMallocAsync(reinterpret_cast<void**>(&work), n * sizeof(float), s));
        hipLaunchKernelGGL(convert, dim3(2048), dim3(256), 0, s, d_in, work, n);
        // other allocations are alive and being written meanwhile, as in the shim (pooled blocks, plain hipMalloc)
        if (c % 3 == 0) {
            float* other; const int64_t m = 200000 + static_cast<int64_t>(rng() % 4000000);
            CK(hipMalloc(&other, m * 4));
            hipLaunchKernelGGL(scribble, dim3(1024), dim3(256), 0, s, other, m, 1.0e9f + c);
            pool.push_back({other, size_t(m * 4)});
            if (pool.size() > 24) { CK(hipFree(pool.front().first)); pool.erase(pool.begin()); }      // hipFree synchronises the device
        }
        hipLaunchKernelGGL(check, dim3(2048), dim3(256), 0, s, d_in, work, d_out, n, bad, first_bad);
        if (variant != 1) CK(hipFreeAsync(work, s));
        unsigned long long h_bad; long long h_first;
        CK(hipMemcpyAsync(&h_bad, bad, 8, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(&h_first, first_bad, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        if (variant == 1) CK(hipFree(work));
        if (h_bad) {
            ++failures; total_bad += h_bad;
            if (failures <= 3) printf("call %d: %llu of %lld elements of the stream-ordered block were not what convert() wrote (first at %lld)\n", c, h_bad, (long long)n, h_first);
        }
        pool.push_back({d_in, in_b}); pool.push_back({d_out, out_b});            // parked, not freed
        while (pool.size() > 24) { CK(hipFree(pool.front().first)); pool.erase(pool.begin()); }
    }
    int rt = 0; CK(hipRuntimeGetVersion(&rt));
    const char* names[] = {"hipMallocAsync work block (what the library did)", "plain hipMalloc work block", "hipMallocAsync + pinned upload source",
                           "hipMallocAsync + stream sync after the upload", "hipMallocAsync queued before the upload"};
    printf("variant %d (%s), %d calls on %s: %lld calls in which the check kernel saw values convert() had not written (%llu elements); HIP runtime %d\n",
           variant, names[variant], calls, other_stream ? "a created stream" : "the default stream", failures, total_bad, rt);
    return failures ? 1 : 0;
}
