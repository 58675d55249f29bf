// HBM practical ceilings on this device, measured the way a streaming kernel can actually drive the memory system:
// U independent accesses in flight per lane (loads issued together, then the stores), grid swept over 1..32
// workgroups per CU, 16 B and 4 B per lane, and the read : write mixes that matter for the STFT path
// (read-only, write-only, 1:1 copy, 1:2 = hop 256 in / 513 bins out).  Buffers are 2 GiB (read) and 4 GiB (written),
// far past the 256 MiB Infinity Cache.  Prints the best grid per row and the whole sweep.
//
//   hipcc --offload-arch=gfx950 -O3 -o hbm_peaks.bin hbm_peaks.hip && ./hbm_peaks.bin
//
// MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; the round-1 version of this file (one access in flight per
// lane, grid fixed at 2048) under-drove the chip and read 4.7-4.9 TB/s for the same copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 0 read, 1 write, 2 copy, 3 read 1 : write 2.  T = v4f (16 B/lane) or float (4 B/lane).  U accesses in flight.
// Block-cyclic: a workgroup takes tiles of 256*U elements, tile index strides by gridDim.x, so at any moment the whole
// chip works on one compact window of the buffers.
template <int MODE, typename T, int U>
__global__ __launch_bounds__(256) void stream_k(const T* __restrict__ in, T* __restrict__ out, size_t n_tiles, float* sink) {
    T acc = T{};
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t base = t * (256 * U) + threadIdx.x;
        T v[U];
        if (MODE != 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = in[base + 256 * u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = acc + T{} + (float)(base + u);
        }
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        } else if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) out[base + 256 * u] = v[u];
        } else {
            const size_t ob = t * (512 * U) + threadIdx.x;
#pragma unroll
            for (int u = 0; u < U; ++u) { out[ob + 512 * u] = v[u]; out[ob + 512 * u + 256] = v[u]; }
        }
    }
    if (MODE == 0) { const float* a = reinterpret_cast<const float*>(&acc); if (a[0] == 123.456f) *sink = a[0]; }
}

// The STFT kernel's own stream shape: a persistent grid of waves, wave w owns a contiguous run of 2052-B rows, per
// row it reads 1024 B (two 8-B loads per lane, prefetched one row ahead) and writes 513 floats with nine 4-B stores
// (four ascending 256-B segments, four descending, one uniform) -- no arithmetic.  SHAPE 1 = the same rows written
// as 16-B stores from a workgroup-wide contiguous span (what an LDS-staged epilogue would emit).
template <int SHAPE>
__global__ __launch_bounds__(256) void rows_k(const float* __restrict__ x, float* __restrict__ out, long n_rows, int n_waves, float* sink) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (SHAPE == 0) {
        if (w >= n_waves) return;
        const long g0 = n_rows * w / n_waves, g1 = n_rows * (w + 1) / n_waves;
        float2 v = *reinterpret_cast<const float2*>(x + g0 * 256 + 2 * lane);
        float2 u = *reinterpret_cast<const float2*>(x + g0 * 256 + 128 + 2 * lane);
        for (long f = g0; f < g1; ++f) {
            const float s = v.x + v.y + u.x + u.y;
            if (f + 1 < g1) {
                v = *reinterpret_cast<const float2*>(x + (f + 1) * 256 + 2 * lane);
                u = *reinterpret_cast<const float2*>(x + (f + 1) * 256 + 128 + 2 * lane);
            }
            float* row = out + f * 513;
#pragma unroll
            for (int m = 0; m < 4; ++m) { row[lane + 64 * m] = s; row[512 - lane - 64 * m] = s; }
            row[256] = s;
        }
    } else if (SHAPE >= 2) {
        // round 3: the same per-row accesses, but the rows are dealt round-robin in runs of R = SHAPE rows (wave w takes runs w, w + n_waves,
        // ...): at any moment the whole grid writes ONE compact window of n_waves * R rows instead of n_waves far-apart streams.  (Memory only: the
        // STFT kernel would also have to reload its 8-block sample window at every run start.)
        if (w >= n_waves) return;
        constexpr int R = SHAPE;
        const long n_runs = (n_rows + R - 1) / R;
        for (long q = w; q < n_runs; q += n_waves) {
            const long g0 = q * R, g1 = g0 + R < n_rows ? g0 + R : n_rows;
            float2 v = *reinterpret_cast<const float2*>(x + g0 * 256 + 2 * lane);
            float2 u = *reinterpret_cast<const float2*>(x + g0 * 256 + 128 + 2 * lane);
            for (long f = g0; f < g1; ++f) {
                const float s = v.x + v.y + u.x + u.y;
                if (f + 1 < g1) {
                    v = *reinterpret_cast<const float2*>(x + (f + 1) * 256 + 2 * lane);
                    u = *reinterpret_cast<const float2*>(x + (f + 1) * 256 + 128 + 2 * lane);
                }
                float* row = out + f * 513;
#pragma unroll
                for (int m = 0; m < 4; ++m) { row[lane + 64 * m] = s; row[512 - lane - 64 * m] = s; }
                row[256] = s;
            }
        }
    } else {
        // workgroup b owns rows [n_rows*b/nb, n_rows*(b+1)/nb): reads them 16 B per lane, writes the 513/256-times larger
        // output span 16 B per lane, 4 accesses in flight
        const int nb = gridDim.x;
        const long r0 = n_rows * blockIdx.x / nb, r1 = n_rows * (blockIdx.x + 1) / nb;
        const v4f* xi = reinterpret_cast<const v4f*>(x + r0 * 256);
        const long n_in = (r1 - r0) * 64;                        // v4f elements
        long ob = (r0 * 513) & ~3L;                              // aligned start inside the output
        const long oe = (r1 * 513) & ~3L;
        v4f* o = reinterpret_cast<v4f*>(out);
        long oi = ob / 4 + threadIdx.x;
        const long oend = oe / 4;
        for (long i = threadIdx.x; i < n_in; i += 256 * 2) {
            const v4f a = xi[i];
            const v4f b = (i + 256 < n_in) ? xi[i + 256] : a;
            // two loads feed four stores (1 : 2 bytes)
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (oi < oend) o[oi] = (k & 1) ? b : a; oi += 256; }
        }
        if (sink && n_in < 0) *sink = 0;
    }
}

// Round 3: the same rows (SHAPE 0), but a wave requests row f + DEPTH while it stores row f, with unconditional (clamped) loads so that
// the compiler's s_waitcnt is exact: vmcnt counts loads and stores in one in-order queue, so with DEPTH = 1 the wait for the next row's
// samples also waits for the stores of the row before (one row of stores in flight per wave); DEPTH = 2 / 3 leave two / three rows of stores
// in flight.  Does the memory system take more from the SAME 3 072 streams if each has more in flight?
template <int DEPTH>
__global__ __launch_bounds__(256) void rows_deep_k(const float* __restrict__ x, float* __restrict__ out, long n_rows, int n_waves) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const long g0 = n_rows * w / n_waves, g1 = n_rows * (w + 1) / n_waves;
    if (g0 >= g1) return;
    float2 v[DEPTH], u[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const long r = g0 + d < g1 ? g0 + d : g1 - 1;
        v[d] = *reinterpret_cast<const float2*>(x + r * 256 + 2 * lane);
        u[d] = *reinterpret_cast<const float2*>(x + r * 256 + 128 + 2 * lane);
    }
    for (long f = g0; f < g1; ++f) {
        const long r = f + DEPTH < g1 ? f + DEPTH : g1 - 1;
        const float2 vn = *reinterpret_cast<const float2*>(x + r * 256 + 2 * lane);
        const float2 un = *reinterpret_cast<const float2*>(x + r * 256 + 128 + 2 * lane);
        const float s = v[0].x + v[0].y + u[0].x + u[0].y;
        float* row = out + f * 513;
#pragma unroll
        for (int m = 0; m < 4; ++m) { row[lane + 64 * m] = s; row[512 - lane - 64 * m] = s; }
        row[256] = s;
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d) { v[d] = v[d + 1]; u[d] = u[d + 1]; }
        v[DEPTH - 1] = vn; u[DEPTH - 1] = un;
    }
}

template <int DEPTH>
void rows_deep(const float* x, float* out, int occ) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long n_rows = 119808L * 8;
    const int n_waves = 256 * 4 * occ, grid = n_waves / 4;
    hipLaunchKernelGGL((rows_deep_k<DEPTH>), dim3(grid), dim3(256), 0, 0, x, out, n_rows, n_waves);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 4;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((rows_deep_k<DEPTH>), dim3(grid), dim3(256), 0, 0, x, out, n_rows, n_waves);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("stft rows, %d row(s) requested ahead   occ %d waves/SIMD: %.2f TB/s  (%.1f us per 119808 rows)\n", DEPTH, occ,
           (double)n_rows * 3076 * reps / (ms * 1e-3) / 1e12, ms * 1e3 / reps / 8);
    fflush(stdout);
}

static float* g_sink;
static double g_best;

template <int MODE, typename T, int U>
void sweep(const char* name, const T* a, T* b, size_t bytes_in, double bytes_moved) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t n_tiles = bytes_in / sizeof(T) / (256 * U);
    double best = 0; int best_grid = 0;
    printf("%-34s", name);
    for (int per_cu = 1; per_cu <= 32; per_cu *= 2) {
        const int grid = 256 * per_cu;
        hipLaunchKernelGGL((stream_k<MODE, T, U>), dim3(grid), dim3(256), 0, 0, a, b, n_tiles, g_sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 4;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_k<MODE, T, U>), dim3(grid), dim3(256), 0, 0, a, b, n_tiles, g_sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double tbs = bytes_moved * reps / (ms * 1e-3) / 1e12;
        printf(" %5.2f", tbs);
        if (tbs > best) { best = tbs; best_grid = per_cu; }
    }
    printf("   best %.2f TB/s @ %d wg/CU\n", best, best_grid);
    g_best = best;
    fflush(stdout);
}


// Round 4: the traffic shape of a workgroup-staged headline kernel, memory + LDS only.  A workgroup of 4 waves owns one
// contiguous run of rows (frames); per step it takes 4 consecutive rows 4j + i (wave i): the 4 * 256 new samples of the step arrive as ONE
// coalesced 16-byte load per lane (256 lanes x 16 B), go into an LDS ring (ds_write_b128), one LDS-only workgroup barrier, then every wave
// reads its frame from the ring (8 ds_read_b64 per lane: z[lane + 64 a], as the register layout wants it) and stores its row with the nine
// dword stores of the product kernel -- so the workgroup writes 4 adjacent rows (8 208 contiguous bytes) and the chip sees a quarter of the
// write fronts of the wave-per-run layout.  The next step's samples are requested before this step's rows are stored.  No arithmetic.
// RING: 4096 samples (16 KiB) per workgroup, so the step j + 1 stores never land on what step j still reads (one barrier per step).
template <int OCC>
__global__ __launch_bounds__(256, OCC) void rows_wg_k(const float* __restrict__ x, float* __restrict__ out, long n_rows, int n_wg) {
    __shared__ __attribute__((aligned(16))) float ring[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x >= n_wg) return;
    const long g0 = (n_rows / 4 * blockIdx.x / n_wg) * 4, g1 = blockIdx.x + 1 == n_wg ? n_rows & ~3L : (n_rows / 4 * (blockIdx.x + 1) / n_wg) * 4;
    if (g0 >= g1) return;
    // prologue: the first 1792 samples of the run (7 x 256) -- two 16-byte loads per lane cover 2048
    const float* src = x + g0 * 256;
    {
        const v4f a = *reinterpret_cast<const v4f*>(src + 4 * threadIdx.x);
        const v4f b = *reinterpret_cast<const v4f*>(src + 1024 + 4 * threadIdx.x);
        *reinterpret_cast<v4f*>(ring + 4 * threadIdx.x) = a;
        *reinterpret_cast<v4f*>(ring + 1024 + 4 * threadIdx.x) = b;
    }
    long pos = 2048;                                         // samples of the run staged so far
    v4f nxt = *reinterpret_cast<const v4f*>(src + pos + 4 * threadIdx.x);
    for (long f = g0; f < g1; f += 4) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();                        // the step's samples are in the ring
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // this wave's frame: samples (f - g0 + wave) * 256 ... + 1024 of the run, ring index modulo 4096
        const int base = static_cast<int>(((f - g0 + wave) * 256) & 4095);
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            typedef float v2f_ __attribute__((ext_vector_type(2)));
            const v2f_ v = *reinterpret_cast<const volatile v2f_*>(ring + ((base + 2 * lane + 128 * a) & 4095));
            s += v.x + v.y;
        }
        // the samples the NEXT step needs beyond what is staged: 1024 more; store the ones already fetched, request the following ones
        *reinterpret_cast<v4f*>(ring + ((pos + 4 * threadIdx.x) & 4095)) = nxt;
        pos += 1024;
        {
            const long want = pos + 4 * threadIdx.x;
            const long lim = (g1 - g0) * 256 + 768 - 4;      // last sample of the run's last frame
            nxt = *reinterpret_cast<const v4f*>(src + (want < lim ? want : lim));
        }
        float* row = out + (f + wave) * 513;
#pragma unroll
        for (int m = 0; m < 4; ++m) { row[lane + 64 * m] = s; row[512 - lane - 64 * m] = s; }
        row[256] = s;
    }
}

template <int OCC>
void rows_wg(const float* x, float* out) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long n_rows = 119808L * 8;
    const int n_wg = 256 * OCC;                              // OCC workgroups of 4 waves per CU = OCC waves per SIMD
    hipLaunchKernelGGL((rows_wg_k<OCC>), dim3(n_wg), dim3(256), 0, 0, x, out, n_rows, n_wg);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 4;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((rows_wg_k<OCC>), dim3(n_wg), dim3(256), 0, 0, x, out, n_rows, n_wg);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("stft rows, workgroup-staged (4 waves, LDS ring, 4 adjacent rows per step)   occ %d waves/SIMD: %.2f TB/s  (%.1f us per 119808 rows)\n", OCC,
           (double)n_rows * 3076 * reps / (ms * 1e-3) / 1e12, ms * 1e3 / reps / 8);
    fflush(stdout);
}

// Round 4: the product's row shape with the row pointer made wave-uniform (v_readfirstlane of both halves), so that the nine stores take the
// `global_store_dword v_offset, v_data, s[base:base+1] offset:imm` form (one address VGPR instead of a 64-bit pair, no 64-bit VALU adds) and the
// loads likewise.  Same traffic as rows_k<0>.
__device__ __forceinline__ long uniform_long(long v) {       // (an element offset, so that the pointer keeps its global address space)
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(static_cast<unsigned long long>(v) >> 32));
    return static_cast<long>((static_cast<unsigned long long>(hi) << 32) | lo);
}
__global__ __launch_bounds__(256) void rows_saddr_k(const float* __restrict__ x, float* __restrict__ out, long n_rows, int n_waves) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= n_waves) return;
    const long g0 = n_rows * w / n_waves, g1 = n_rows * (w + 1) / n_waves;
    long xo = uniform_long(g0 * 256), ro = uniform_long(g0 * 513);
    float2 v = *reinterpret_cast<const float2*>(x + xo + 2 * lane);
    float2 u = *reinterpret_cast<const float2*>(x + xo + 128 + 2 * lane);
    const int up = lane, dn = 512 - lane;                    // the two per-lane offsets (floats)
    for (long f = g0; f < g1; ++f) {
        const float s = v.x + v.y + u.x + u.y;
        xo = uniform_long(xo + 256);
        if (f + 1 < g1) {
            v = *reinterpret_cast<const float2*>(x + xo + 2 * lane);
            u = *reinterpret_cast<const float2*>(x + xo + 128 + 2 * lane);
        }
        float* const row = out + ro;
#pragma unroll
        for (int m = 0; m < 4; ++m) { row[up + 64 * m] = s; row[dn - 64 * m] = s; }
        row[256] = s;
        ro = uniform_long(ro + 513);
    }
}

void rows_saddr(const float* x, float* out, int occ) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long n_rows = 119808L * 8;
    const int n_waves = 256 * 4 * occ, grid = n_waves / 4;
    hipLaunchKernelGGL(rows_saddr_k, dim3(grid), dim3(256), 0, 0, x, out, n_rows, n_waves);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 4;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(rows_saddr_k, dim3(grid), dim3(256), 0, 0, x, out, n_rows, n_waves);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("stft rows, SGPR base + VGPR offset stores    occ %d waves/SIMD: %.2f TB/s  (%.1f us per 119808 rows)\n", occ,
           (double)n_rows * 3076 * reps / (ms * 1e-3) / 1e12, ms * 1e3 / reps / 8);
    fflush(stdout);
}

template <int SHAPE>
void rows(const char* name, const float* x, float* out, int occ) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long n_rows = 119808L * 8;                 // 8 cfg2 batches: 0.98 GB in, 1.97 GB out per launch
    const int n_waves = 256 * 4 * occ, grid = n_waves / 4;
    hipLaunchKernelGGL((rows_k<SHAPE>), dim3(grid), dim3(256), 0, 0, x, out, n_rows, n_waves, g_sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 4;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((rows_k<SHAPE>), dim3(grid), dim3(256), 0, 0, x, out, n_rows, n_waves, g_sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n_rows * 3076;
    printf("%-34s occ %d waves/SIMD: %.2f TB/s  (%.1f us per 119808 rows)\n", name, occ, bytes * reps / (ms * 1e-3) / 1e12, ms * 1e3 / reps / 8);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)2 << 30;      // 2 GiB read buffer, 4 GiB written buffer
    void *a, *b;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, 2 * bytes)); CHECK(hipMalloc(&g_sink, 4));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, 2 * bytes));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; columns = 1 2 4 8 16 32 workgroups (256 threads) per CU; TB/s of bytes moved\n", prop.gcnArchName, prop.multiProcessorCount);
    const v4f* a4 = (const v4f*)a; v4f* b4 = (v4f*)b; const float* a1 = (const float*)a; float* b1 = (float*)b;
    const double B = (double)bytes;
    if (argc > 2 && !strcmp(argv[1], "sustain")) {     // ./hbm_peaks.bin sustain <rows|read|write|mix> [secs]: a steady stream for tools/telemetry.py
        const double secs = argc > 3 ? atof(argv[3]) : 4.0;
        auto t0 = std::chrono::steady_clock::now();
        long n = 0; double el = 0;
        while (el < secs) {
            for (int i = 0; i < 4; ++i) {
                if (!strcmp(argv[2], "rows")) hipLaunchKernelGGL((rows_k<0>), dim3(2048), dim3(256), 0, 0, a1, b1, 119808L * 8, 8192, g_sink);
                else if (!strcmp(argv[2], "read")) hipLaunchKernelGGL((stream_k<0, v4f, 4>), dim3(512), dim3(256), 0, 0, a4, b4, bytes / 16 / 1024, g_sink);
                else if (!strcmp(argv[2], "write")) hipLaunchKernelGGL((stream_k<1, v4f, 4>), dim3(256), dim3(256), 0, 0, a4, b4, bytes / 16 / 1024, g_sink);
                else hipLaunchKernelGGL((stream_k<3, float, 16>), dim3(256), dim3(256), 0, 0, a1, b1, bytes / 4 / 4096, g_sink);
            }
            CHECK(hipDeviceSynchronize());
            n += 4;
            el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        const double per = !strcmp(argv[2], "rows") ? 119808.0 * 8 * 3076 : !strcmp(argv[2], "mix") ? 3 * B : B;
        printf("sustain %s: %.2f TB/s over %.1f s\n", argv[2], per * n / el / 1e12, el);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "deep")) {          // ./hbm_peaks.bin deep: rows requested 1 / 2 / 3 ahead of the stores
        float *a1, *b1;
        CHECK(hipMalloc(&a1, (size_t)2 << 30)); CHECK(hipMalloc(&b1, (size_t)4 << 30));
        CHECK(hipMemset(a1, 0, (size_t)2 << 30));
        for (int rep = 0; rep < 2; ++rep)
            for (int occ : {2, 3, 4}) { rows_deep<1>(a1, b1, occ); rows_deep<2>(a1, b1, occ); rows_deep<3>(a1, b1, occ); }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "wgstage")) {       // ./hbm_peaks.bin wgstage: the workgroup-staged row model against the wave-per-run one
        for (int rep = 0; rep < 2; ++rep) {
            rows<0>("stft rows (wave per run, product)", a1, b1, 2);
            rows<0>("stft rows (wave per run, product)", a1, b1, 3);
            rows<0>("stft rows (wave per run, product)", a1, b1, 4);
            rows_wg<2>(a1, b1); rows_wg<3>(a1, b1); rows_wg<4>(a1, b1);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "saddr")) {         // ./hbm_peaks.bin saddr: the product's rows with 64-bit VGPR addresses against SGPR base + VGPR offset
        for (int rep = 0; rep < 2; ++rep)
            for (int occ : {2, 3, 4}) { rows<0>("stft rows (64-bit VGPR addresses)", a1, b1, occ); rows_saddr(a1, b1, occ); }
        return 0;
    }
    const bool rows_only = argc > 1 && !strcmp(argv[1], "rows");     // ./hbm_peaks.bin rows: only the STFT row-pattern models
    if (!rows_only) {
    sweep<0, v4f, 1>("read   16B x1", a4, b4, bytes, B);
    sweep<0, v4f, 4>("read   16B x4", a4, b4, bytes, B);
    sweep<0, v4f, 8>("read   16B x8", a4, b4, bytes, B);
    sweep<1, v4f, 1>("write  16B x1", a4, b4, bytes, B);
    sweep<1, v4f, 4>("write  16B x4", a4, b4, bytes, B);
    sweep<1, v4f, 8>("write  16B x8", a4, b4, bytes, B);
    sweep<2, v4f, 1>("copy   16B x1 (r+w)", a4, b4, bytes, 2 * B);
    sweep<2, v4f, 4>("copy   16B x4 (r+w)", a4, b4, bytes, 2 * B);
    sweep<2, v4f, 8>("copy   16B x8 (r+w)", a4, b4, bytes, 2 * B);
    sweep<3, v4f, 1>("r1:w2  16B x1", a4, b4, bytes, 3 * B);
    sweep<3, v4f, 4>("r1:w2  16B x4", a4, b4, bytes, 3 * B);
    sweep<3, v4f, 8>("r1:w2  16B x8", a4, b4, bytes, 3 * B);
    sweep<0, float, 4>("read    4B x4", a1, b1, bytes, B);
    sweep<0, float, 16>("read    4B x16", a1, b1, bytes, B);
    sweep<1, float, 4>("write   4B x4", a1, b1, bytes, B);
    sweep<1, float, 16>("write   4B x16", a1, b1, bytes, B);
    sweep<2, float, 16>("copy    4B x16 (r+w)", a1, b1, bytes, 2 * B);
    sweep<3, float, 4>("r1:w2   4B x4", a1, b1, bytes, 3 * B);
    sweep<3, float, 16>("r1:w2   4B x16", a1, b1, bytes, 3 * B);
    }
    for (int occ : {1, 2, 4, 8}) rows<0>("stft rows: 2x8B in, 9x4B out/row", a1, b1, occ);
    for (int occ : {1, 2, 4, 8}) rows<1>("stft rows, 16B contiguous spans", a1, b1, occ);
    for (int occ : {2, 3, 4}) rows<0>("stft rows (again, for the A/B)", a1, b1, occ);
    for (int occ : {2, 3, 4}) rows<4>("stft rows, round-robin runs of 4", a1, b1, occ);
    for (int occ : {2, 3, 4}) rows<8>("stft rows, round-robin runs of 8", a1, b1, occ);
    for (int occ : {2, 3, 4}) rows<16>("stft rows, round-robin runs of 16", a1, b1, occ);
    for (int occ : {2, 3, 4}) rows<32>("stft rows, round-robin runs of 32", a1, b1, occ);
    return 0;
}
