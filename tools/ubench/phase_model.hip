// Micro-benchmark (design aid, not part of the product): do a wave's VALU phases and its LDS exchange phases overlap across the
// waves of a SIMD / CU the way the register-FFT kernels assume?  One "phase" = NV independent v_fma_f32 (8 chains), then NW
// ds_write_b64 + NR ds_read_b64 through the wave's own conflict-free slab, then s_waitcnt lgkmcnt(0) and a use of what was read.
// Variants: VALU only, LDS only, both; 1..4 waves per SIMD.  If both ~ max(VALU, LDS) the units overlap; if both ~ sum they do not.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/phase_model.hip -o tools/ubench/phase_model.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) volatile v2f lds_v2f;

// WMODE: 0 ds_write_b64, 1 ds_write_b32 x2, 2 ds_write_b128 (two values per instruction)
template <int NV, int NW, int NR, int WMODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    extern __shared__ v2f lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2f* slab = lds + wave * (64 * 9);                        // 4.5 KiB per wave
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    const long long c0 = __builtin_amdgcn_s_memtime();
    const long long rt0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < NV / 8; ++u)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        if (NW > 0) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if (WMODE == 0) *(lds_v2f*)(slab + lane + 64 * (w & 7)) = v2f{a0, a1};
                else if (WMODE == 1) {
                    *(__attribute__((address_space(3))) volatile float*)((float*)slab + lane + 64 * (w & 7)) = a0;
                    *(__attribute__((address_space(3))) volatile float*)((float*)slab + lane + 64 * (8 + (w & 7))) = a1;
                } else if ((w & 1) == 0) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    *(__attribute__((address_space(3))) volatile v4f*)(slab + 2 * lane + 128 * ((w >> 1) & 3)) = v4f{a0, a1, a2, a3};
                }
            }
        }
        if (NR > 0) {
            v2f acc = {0.f, 0.f};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < NR; ++r) { const v2f v = *(lds_v2f*)(slab + (lane ^ 1) + 64 * (r & 7)); acc += v; }
            a0 += acc.x * 1e-30f; a1 += acc.y * 1e-30f;       // the next phase depends on what was read
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if (cyc && blockIdx.x == 0 && threadIdx.x == 0) *cyc = clock64() - c0;
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// The same work per phase, software-pipelined inside the wave: the stores + loads of exchange group g+1 are ISSUED, then the VALU block
// that consumes group g runs (its operands arrived one block ago), so a wave's own LDS traffic is in flight behind its own VALU issue.
// G groups per phase: each group = NW/G stores + NR/G loads + NV/G VALU.
template <int NV, int NW, int NR, int G>
__global__ __launch_bounds__(256) void kpipe(float* out, int iters, long long* cyc) {
    extern __shared__ v2f lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2f* slab = lds + wave * (64 * 9);
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    v2f cur = {0.f, 0.f};
    const long long c0 = clock64();
    for (int i = 0; i < iters * G; ++i) {
        v2f nxt = {0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW / G; ++w) *(lds_v2f*)(slab + lane + 64 * (w & 7)) = v2f{a0, a1};
#pragma unroll
        for (int r = 0; r < NR / G; ++r) { const v2f v = *(lds_v2f*)(slab + (lane ^ 1) + 64 * (r & 7)); nxt += v; }
        a2 += cur.x * 1e-30f; a3 += cur.y * 1e-30f;           // this block consumes the PREVIOUS group's loads
#pragma unroll
        for (int u = 0; u < NV / G / 8; ++u)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        cur = nxt;
    }
    if (cyc && blockIdx.x == 0 && threadIdx.x == 0) *cyc = clock64() - c0;
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + cur.x;
}

template <int NV, int NW, int NR, int G>
double run_pipe(int occ, float* out, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nwg = 256 * occ;
    const size_t lds = 4 * 64 * 9 * sizeof(v2f);
    hipLaunchKernelGGL((kpipe<NV, NW, NR, G>), dim3(nwg), dim3(256), lds, 0, out, 50, (long long*)nullptr);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((kpipe<NV, NW, NR, G>), dim3(nwg), dim3(256), lds, 0, out, iters, (long long*)nullptr);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("  NV=%4d NW=%2d NR=%2d PIPELINED in %d groups occ=%d: %.3f ms\n", NV, NW, NR, G, occ, ms);
    return ms;
}

// Role split: odd blocks run LDS-only phases, even blocks VALU-only phases (same counts as the mixed kernel): do the SIMD's VALU pipe and
// the CU's LDS pipe run concurrently when DIFFERENT waves use them?  stamps[2*b], [2*b+1]: s_memtime / s_memrealtime deltas of block b's wave 0.
template <int NV, int NW, int NR>
__global__ __launch_bounds__(256) void ksplit(float* out, int iters, unsigned long long* stamps) {
    extern __shared__ v2f lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2f* slab = lds + wave * (64 * 9);
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if ((blockIdx.x >> 8) & 1) {                            // blocks b and b + 256 share a CU when the dispatcher deals round-robin (checked on the host from HW_ID)
        for (int i = 0; i < iters; ++i) {
            v2f acc = {0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; ++w) *(lds_v2f*)(slab + lane + 64 * (w & 7)) = v2f{a0, a1};
#pragma unroll
            for (int r = 0; r < NR; ++r) { const v2f v = *(lds_v2f*)(slab + (lane ^ 1) + 64 * (r & 7)); acc += v; }
            a0 += acc.x * 1e-30f; a1 += acc.y * 1e-30f;
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < NV / 8; ++u)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
    }
    if (threadIdx.x == 0) {
        stamps[3 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        stamps[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
        // HW_REG_HW_ID (4): cu_id [11:8], sh_id [12], se_id [15:13]; HW_REG_XCC_ID (20): xcc_id [3:0]
        stamps[3 * blockIdx.x + 2] = ((__builtin_amdgcn_s_getreg((4) | (8 << 6) | ((8 - 1) << 11)) & 0xff) << 4) | (__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 0xf);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int NV, int NW, int NR>
void run_split(int occ, float* out, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nwg = 256 * occ;
    const size_t lds = 4 * 64 * 9 * sizeof(v2f);
    unsigned long long* st; CHECK(hipHostMalloc(&st, nwg * 24));
    hipLaunchKernelGGL((ksplit<NV, NW, NR>), dim3(nwg), dim3(256), lds, 0, out, 50, st);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((ksplit<NV, NW, NR>), dim3(nwg), dim3(256), lds, 0, out, iters, st);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double tv = 0, tl = 0, cv = 0, cl = 0; int nv = 0, nl = 0;
    static int mixv[4096], mixl[4096];
    for (int i = 0; i < 4096; ++i) mixv[i] = mixl[i] = 0;
    for (int b2 = 0; b2 < nwg; ++b2) {
        const double us = st[3 * b2 + 1] / 100.0, ghz = st[3 * b2] / (st[3 * b2 + 1] * 10.0);
        if ((b2 >> 8) & 1) { tl += us; cl += ghz; ++nl; ++mixl[st[3 * b2 + 2] & 4095]; } else { tv += us; cv += ghz; ++nv; ++mixv[st[3 * b2 + 2] & 4095]; }
    }
    int cus = 0, balanced = 0;
    for (int i = 0; i < 4096; ++i) if (mixv[i] + mixl[i]) { ++cus; balanced += mixv[i] == mixl[i]; }
    printf("  (placement: %d CUs seen, %d of them hold as many VALU blocks as LDS blocks)\n", cus, balanced);
    printf("  ROLE SPLIT NV=%d | NW=%d NR=%d, %d waves/SIMD in all: %.3f ms; VALU blocks lived %.0f us at %.2f GHz, LDS blocks %.0f us at %.2f GHz (means)\n",
           NV, NW, NR, occ, ms, tv / nv, cv / nv, tl / nl, cl / nl);
    CHECK(hipHostFree(st));
}

// Scheduling variants of the mixed phase: PM 1 = s_setprio 3 around the LDS block (0 in the VALU block), 2 = the reverse,
// 3 = waves of odd blocks start half a phase later (their first VALU block is skipped), 4 = 3 + 1, 5 = by SIMD pair.
template <int NV, int NW, int NR, int PM>
__global__ __launch_bounds__(256) void kprio(float* out, int iters, unsigned long long* st) {
    extern __shared__ v2f lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2f* slab = lds + wave * (64 * 9);
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const bool late = (PM == 3 || PM == 4) && ((blockIdx.x >> 8) & 1);
    for (int i = 0; i < iters; ++i) {
        if (PM == 1 || PM == 4) __builtin_amdgcn_s_setprio(0);
        if (PM == 2) __builtin_amdgcn_s_setprio(3);
        if (!(late && i == 0)) {
#pragma unroll
            for (int u = 0; u < NV / 8; ++u)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
        if (PM == 1 || PM == 4) __builtin_amdgcn_s_setprio(3);
        if (PM == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int w = 0; w < NW; ++w) *(lds_v2f*)(slab + lane + 64 * (w & 7)) = v2f{a0, a1};
        v2f acc = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < NR; ++r) { const v2f v = *(lds_v2f*)(slab + (lane ^ 1) + 64 * (r & 7)); acc += v; }
        a0 += acc.x * 1e-30f; a1 += acc.y * 1e-30f;
    }
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { st[0] = __builtin_amdgcn_s_memtime() - t0; st[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int NV, int NW, int NR, int PM>
void run_prio(int occ, float* out, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nwg = 256 * occ;
    const size_t lds = 4 * 64 * 9 * sizeof(v2f);
    unsigned long long* st; CHECK(hipHostMalloc(&st, 16)); st[0] = st[1] = 1;
    hipLaunchKernelGGL((kprio<NV, NW, NR, PM>), dim3(nwg), dim3(256), lds, 0, out, 50, st);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((kprio<NV, NW, NR, PM>), dim3(nwg), dim3(256), lds, 0, out, iters, st);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ghz = st[0] / (st[1] * 10.0);
    printf("  NV=%4d NW=%2d NR=%2d sched variant %d occ=%d: %.3f ms = %.0f cycles per wave-phase per SIMD at %.2f GHz\n", NV, NW, NR, PM, occ, ms,
           ms * 1e6 * ghz / iters / occ, ghz);
    CHECK(hipHostFree(st));
}

// The exchange form proposed for rbig in round 4: stores by ds_write_addtid_b32 (no address VGPR: M0 + offset + 4 * lane; two planes, re / im),
// loads by ds_read_b128 of four neighbouring lanes' dwords.  NW complex values stored = 2 * NW addtid stores; NR complex loaded = NR / 2 b128 loads.
// RPAT: 0 = every lane reads its own 16 bytes (trivially conflict-free), 1 = the 8-slot pattern of the design note (slot = lane & 7, block = lane >> 3,
// slot bases staggered by 16 * {0, 1, 8, 9} bytes)
template <int NV, int NW, int NR, int RPAT>
__global__ __launch_bounds__(256) void kaddtid(float* out, int iters, unsigned long long* st) {
    extern __shared__ v2f lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int kSlot = 256 + 16 * 10;                       // bytes between slots of a plane (room for the stagger)
    constexpr int kPlane = 8 * kSlot + 256;                    // bytes between the re and the im plane
    const unsigned base = wave * 2 * kPlane;                   // this wave's slab, byte address in LDS
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    const int sl = lane & 7, blk = lane >> 3;
    const int stag[8] = {0, 1, 8, 9, 0, 1, 8, 9};
    const unsigned raddr = RPAT == 0 ? base + 16 * lane : base + sl * kSlot + 16 * stag[sl] + 32 * blk;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < NV / 8; ++u)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#pragma unroll
        for (int g = 0; g < NW / 8; ++g) {                     // one group = 8 complex per lane = 16 addtid stores, then 4 b128 loads
            asm volatile("s_mov_b32 m0, %8\n"
                         "ds_write_addtid_b32 %0 offset:0\n ds_write_addtid_b32 %1 offset:%9\n"
                         "ds_write_addtid_b32 %2 offset:%10\n ds_write_addtid_b32 %3 offset:%9+%10\n"
                         "ds_write_addtid_b32 %4 offset:2*%10\n ds_write_addtid_b32 %5 offset:%9+2*%10\n"
                         "ds_write_addtid_b32 %6 offset:3*%10\n ds_write_addtid_b32 %7 offset:%9+3*%10\n"
                         "ds_write_addtid_b32 %0 offset:4*%10\n ds_write_addtid_b32 %1 offset:%9+4*%10\n"
                         "ds_write_addtid_b32 %2 offset:5*%10\n ds_write_addtid_b32 %3 offset:%9+5*%10\n"
                         "ds_write_addtid_b32 %4 offset:6*%10\n ds_write_addtid_b32 %5 offset:%9+6*%10\n"
                         "ds_write_addtid_b32 %6 offset:7*%10\n ds_write_addtid_b32 %7 offset:%9+7*%10\n"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(base), "n"(kPlane), "n"(kSlot) : "memory");
            if (NR > 0) {
                v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const v4f v = *(__attribute__((address_space(3))) volatile v4f*)(uintptr_t)(raddr + (r & 1) * 16 + (r >> 1) * kPlane);
                    acc += v;
                }
                a0 += (acc.x + acc.y) * 1e-30f; a1 += (acc.z + acc.w) * 1e-30f;
            }
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { st[0] = __builtin_amdgcn_s_memtime() - t0; st[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int NV, int NW, int NR, int RPAT>
void run_addtid(int occ, float* out, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nwg = 256 * occ;
    const size_t lds = 4 * 2 * (8 * (256 + 160) + 256) + 1024;
    unsigned long long* st; CHECK(hipHostMalloc(&st, 16)); st[0] = st[1] = 1;
    hipLaunchKernelGGL((kaddtid<NV, NW, NR, RPAT>), dim3(nwg), dim3(256), lds, 0, out, 50, st);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((kaddtid<NV, NW, NR, RPAT>), dim3(nwg), dim3(256), lds, 0, out, iters, st);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ghz = st[0] / (st[1] * 10.0);
    printf("  ADDTID NV=%4d, %2d complex stored (addtid x2), %2d loaded (b128 / 2, pattern %d) occ=%d: %.3f ms = %.0f cycles per wave-phase per SIMD at %.2f GHz\n",
           NV, NW, NR, RPAT, occ, ms, ms * 1e6 * ghz / iters / occ, ghz);
    CHECK(hipHostFree(st));
}

template <int NV, int NW, int NR, int WMODE>
double run(int occ, float* out, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nwg = 256 * occ;
    const size_t lds = 4 * 64 * 9 * sizeof(v2f);
    long long* cyc; CHECK(hipHostMalloc(&cyc, 32)); *cyc = 0;
    hipLaunchKernelGGL((k<NV, NW, NR, WMODE>), dim3(nwg), dim3(256), lds, 0, out, 50, (long long*)nullptr);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NV, NW, NR, WMODE>), dim3(nwg), dim3(256), lds, 0, out, iters, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ghz = cyc[1] / (cyc[2] * 10.0);                    // the LAST block's wave 0: s_memtime ticks per s_memrealtime tick (100 MHz)
    const double cyc_per_phase_simd = ms * 1e6 * ghz / iters / occ; // SIMD cycles per wave-phase at that clock
    printf("  NV=%4d NW=%2d NR=%2d wmode=%d occ=%d: %.3f ms  first wave: %7.1f cycles/phase   per SIMD: %7.1f cycles per wave-phase   clock %.2f GHz\n",
           NV, NW, NR, WMODE, occ, ms, (double)*cyc / iters, cyc_per_phase_simd, ghz);
    CHECK(hipHostFree(cyc));
    return cyc_per_phase_simd;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    const int iters = 4000;
    for (int occ = 1; occ <= 4; ++occ) {
        printf("== %d wave(s) per SIMD (%d waves per CU)\n", occ, 4 * occ);
        printf(" rbig T=4-like phase (400 VALU, 32 writes, 32 reads)\n");
        const double v = run<400, 0, 0, 0>(occ, out, iters);
        const double l = run<0, 32, 32, 0>(occ, out, iters);
        const double b = run<400, 32, 32, 0>(occ, out, iters);
        printf("   -> both / (valu + lds) = %.2f, both / max = %.2f\n", b / (v + l), b / (v > l ? v : l));
        if (occ % 2 == 0) run_split<400, 32, 32>(occ, out, iters);
        run_addtid<0, 32, 32, 0>(occ, out, iters);
        run_addtid<0, 32, 32, 1>(occ, out, iters);
        run_addtid<0, 32, 0, 0>(occ, out, iters);
        run_addtid<400, 32, 32, 1>(occ, out, iters);
        run_addtid<800, 32, 32, 1>(occ, out, iters);
        run_prio<400, 32, 32, 0>(occ, out, iters);
        run_prio<400, 32, 32, 1>(occ, out, iters);
        run_prio<400, 32, 32, 2>(occ, out, iters);
        run_prio<400, 32, 32, 3>(occ, out, iters);
        run_prio<400, 32, 32, 4>(occ, out, iters);
        run_prio<400, 0, 0, 0>(occ, out, iters);
        run_prio<0, 32, 32, 0>(occ, out, iters);
        run_prio<800, 32, 32, 0>(occ, out, iters);
        run_prio<800, 32, 32, 1>(occ, out, iters);
        run_prio<800, 0, 0, 0>(occ, out, iters);
        run_prio<400, 16, 16, 0>(occ, out, iters);
        run_prio<0, 16, 16, 0>(occ, out, iters);
        run_pipe<400, 32, 32, 4>(occ, out, iters);
        run_pipe<400, 32, 32, 2>(occ, out, iters);
        run_pipe<400, 32, 32, 1>(occ, out, iters);
        run<0, 32, 0, 0>(occ, out, iters);
        run<0, 0, 32, 0>(occ, out, iters);
        run<0, 32, 32, 1>(occ, out, iters);
        run<0, 32, 32, 2>(occ, out, iters);
        printf(" rbig T=2-like phase (200 VALU, 16 writes, 16 reads)\n");
        const double v2 = run<200, 0, 0, 0>(occ, out, iters);
        const double l2 = run<0, 16, 16, 0>(occ, out, iters);
        const double b2 = run<200, 16, 16, 0>(occ, out, iters);
        printf("   -> both / (valu + lds) = %.2f, both / max = %.2f\n", b2 / (v2 + l2), b2 / (v2 > l2 ? v2 : l2));
    }
    return 0;
}
