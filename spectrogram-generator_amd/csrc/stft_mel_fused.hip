// stft_mel_fused.hip -- BASELINE cfg3: STFT (nperseg = nfft = 1024, f32, PSD) with the mel filterbank fused as an
// MFMA epilogue; the linear spectrum never reaches HBM.  Algorithmic bytes per frame: hop*4 in + n_mels*4 out
// (1344 B at hop 256 / 80 bands, against 3076 + 2052 + 320 B for sg_stft followed by sg_mel).
//
// Two kernels.  stft1024_mel_kernel (the default, described next) lets all four waves of a workgroup do both phases in
// turn; three workgroups per CU overlap each other's phases.  stft1024_mel_ws_kernel (opt-in: SPECTRO_FUSED_WS=1)
// specialises the waves of one 1024-thread workgroup per CU: eight PRODUCER waves run the r8x3 FFT pipeline (two frames
// each per 16-frame tile) into one of two LDS tiles while CONSUMER waves contract the PREVIOUS tile with the mel weights
// on the matrix cores, so the VALU and the MFMA phase of consecutive tiles overlap by construction.  Measured (cfg2 batch,
// 80 mels, sustained, 2.39 GHz, not power-limited): default 130.9 us, wave-specialised 151.6 us; its producers alone take
// 118 us -- two 64 KiB-class LDS tiles plus eight exchange slabs leave room for only two FFT waves per SIMD, and a wave's
// frame is a long dependent chain (2.0 us per frame per wave) that needs three or four waves per SIMD to fill the VALU --
// and its consumers alone 132 us (the 39 KiB of weights fit neither the LDS beside the tiles nor the L1, so every k block
// waits for L2).  The overlap is real but buys less than the occupancy it costs; kept for the record and for A/B.
//
// A 256-thread workgroup owns a tile of 16 consecutive frames of one clip:
//   phase 1  each of the 4 wavefronts runs the r8x3 pipeline (stft_r8x3.hip: register radix-8 x3, padded LDS
//            transposes, split pass) on 4 consecutive frames and writes |X|^2*scale rows into a shared LDS tile
//            [16][520] instead of global memory;
//   phase 2  the tile is the A operand of v_mfma_f32_16x16x4_f32 (exact f32): wave w takes the 16-bin k blocks
//            w, w+4, ... and, for every 16-band mel tile whose triangles cover the block, one float4 of the packed
//            transposed weights (sg_mel_pack_weights) as B; partial accumulators are summed across the 4 waves
//            through LDS and stored as [frame][mel] rows (optionally 10*log10).
// The mel definition is this library's own (the reference has no mel stage; parity unpinned).
#include "spectro_internal.h"
#include "fft_wave.h"

#include <cmath>
#include <cstdlib>

#ifndef SG_FUSED_FULL_BARRIER
#define SG_FUSED_FULL_BARRIER 0
#endif
#ifndef SG_FUSED_PRIO
#define SG_FUSED_PRIO 1         // wave priority rises along a frame and stays high through the mel phase; 0 = off
#endif

namespace sg {
namespace {

using namespace wavefft;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: every wave would wait at each tile
// boundary for its prefetched samples and -- worse -- for the mel rows it has just stored (1-2 us of write latency per
// barrier, 30 tiles per workgroup).  What the tiles and partial sums need is that the LDS writes have landed.
__device__ __forceinline__ void lds_barrier() {
#if SG_FUSED_FULL_BARRIER
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
constexpr int kM = 512, kN = 1024;
constexpr int kS1 = 72, kS2 = 66, kSlab = 8 * kS1;
constexpr int kRow = 520;                 // floats per tile row: 513 bins + zero pad, 16-byte aligned rows
constexpr int kTileFloats = 16 * kRow + 16;
constexpr int kMaxTiles = 8;

struct FusedParams {
    const float* x;
    int64_t clip_stride;
    int n_frames, hop;
    int tiles_per_clip;
    int64_t total_tiles;
    int n_wgs;
    float* out;                // [clip][frame][n_mels]
    int64_t out_clip_stride;
    const float2* win2;
    const float2* tw;          // r8x3 per-lane twiddles [18][64]
    float scale;
    const float* wt;           // packed mel weights [16*NT][k_pad]
    int k_pad, n_mels, log_scale;
    int form;                  // the caller's flags word (SG_MEL_FORM_*)
    int k_lo[kMaxTiles], k_hi[kMaxTiles];
    int debug;                 // ablation aid (SPECTRO_FUSED_DEBUG): 1 = consumers skip the MFMA loop, 2 = producers skip the FFT, 4 = no tile-row writes
};

// Round-2 changes to this kernel (found while building the wave-specialised form below): workgroup barriers that order LDS
// traffic only, and prefetch loads issued unconditionally; 132.7 -> 130.9 us per cfg2 batch.  (Fetching the weights one k
// block ahead with NT unconditional loads per block cost registers and L2 traffic: 216 us.)
// H = hop/128 for hop 256 (register sliding window like stft_r8x3), 0 otherwise.
template <bool DETREND, int NT, int H>
__global__ __launch_bounds__(256) void stft1024_mel_kernel(const FusedParams p) {
    __shared__ __attribute__((aligned(16))) float2 slabs[4 * kSlab];
    __shared__ __attribute__((aligned(16))) float tile[kTileFloats];       // phase 2 partial sums alias the front of it
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2* const buf = slabs + wave * kSlab;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    if (wg >= p.n_wgs) return;
    int64_t tid_tile = p.total_tiles * wg / p.n_wgs;
    const int64_t tile_end = p.total_tiles * (wg + 1) / p.n_wgs;

    float2 w[8], t1[7], t2[7], t3[4];
#pragma unroll
    for (int a = 0; a < 8; ++a) w[a] = p.win2[lane + 64 * a];
#pragma unroll
    for (int r = 0; r < 7; ++r) { t1[r] = p.tw[r * 64 + lane]; t2[r] = p.tw[(7 + r) * 64 + lane]; }
#pragma unroll
    for (int m = 0; m < 4; ++m) t3[m] = p.tw[(14 + m) * 64 + lane];

    const int j0 = lane & 7, hi = lane >> 3;
    float2* const x1w = buf + hi * kS1 + j0;
    float2* const x1r = buf + lane;
    float2* const x2w = buf + j0 * kS2 + hi;
    float2* const x2r = buf + lane;
    float2* const x3w = buf + lane;
    const float2* const x3b = buf + (kM - lane);
    // sqrt of the PSD scale rides on the window registers (stft_r8x3.hip)
    {
        const float sq = sqrtf(p.scale * 0.5f);
#pragma unroll
        for (int a = 0; a < 8; ++a) { w[a].x *= sq; w[a].y *= sq; }
    }
    const float r0 = lane == 0 ? 0.5f : 1.0f;
    const int mi = lane & 15, kq = lane >> 4;

    for (; tid_tile < tile_end; ++tid_tile) {
        const int clip = static_cast<int>(tid_tile / p.tiles_per_clip);
        const int ft0 = static_cast<int>(tid_tile - static_cast<int64_t>(clip) * p.tiles_per_clip) * 16;
        const int my_f0 = ft0 + 4 * wave;
        const int n_my = max(0, min(4, p.n_frames - my_f0));
        const float* const xclip = p.x + static_cast<int64_t>(clip) * p.clip_stride + 2 * lane;

        // zero the pad columns of this wave's rows (bins 513..519) and, once, the 16 floats behind the last row:
        // phase 2 reads k up to k_pad-1 = 527 and the padded weights there are zero, but 0 * garbage must stay finite
        if (lane < 28) tile[(4 * wave + lane / 7) * kRow + 513 + lane % 7] = 0.f;
        if (wave == 3 && lane < 16) tile[16 * kRow + lane] = 0.f;

        // ---------------- phase 1: 4 frames per wave through the r8x3 pipeline ----------------
        // (loads are unconditional, from an address that is always valid: under a wave-uniform `if` the compiler merges the
        //  loaded and not-loaded register sets with copies right behind the load, i.e. waits for it on the spot)
        float2 raw[8];
        {
            const float* const src0 = n_my > 0 ? xclip + static_cast<int64_t>(my_f0) * p.hop : p.x + 2 * lane;
#pragma unroll
            for (int k = 0; k < 8; ++k) raw[k] = *reinterpret_cast<const float2*>(src0 + 128 * k);
        }
        for (int i = 0; i < 4; ++i) {
            float* const trow = tile + (4 * wave + i) * kRow;
            if (i >= n_my) {                         // frames past the end of the clip: a finite dummy row
#pragma unroll
                for (int m = 0; m < 4; ++m) { trow[lane + 64 * m] = 0.f; trow[kM - lane - 64 * m] = 0.f; }
                trow[256] = 0.f;
                continue;
            }
            float2 a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = raw[k];
            {                                        // prefetch the next frame (or this one again) before this frame's FFT
                const float* const nxt = xclip + static_cast<int64_t>(my_f0 + (i + 1 < n_my ? i + 1 : i)) * p.hop;
                if (H > 0) {
#pragma unroll
                    for (int k = 0; k + H < 8; ++k) raw[k] = raw[k + H];
#pragma unroll
                    for (int k = (H > 0 ? 8 - H : 0); k < 8; ++k) raw[k] = *reinterpret_cast<const float2*>(nxt + 128 * k);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) raw[k] = *reinterpret_cast<const float2*>(nxt + 128 * k);
                }
            }
            if (DETREND) {
                float s = a[0].x + a[0].y;
#pragma unroll
                for (int k = 1; k < 8; ++k) s += a[k].x + a[k].y;
                const float mean = wave_sum(s) * (1.0f / kN);
#pragma unroll
                for (int k = 0; k < 8; ++k) { a[k].x -= mean; a[k].y -= mean; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { a[k].x *= w[k].x; a[k].y *= w[k].y; }
            if (SG_FUSED_PRIO) __builtin_amdgcn_s_setprio(0);
            radix8(a);
#pragma unroll
            for (int r = 1; r < 8; ++r) a[r] = cmul(a[r], t1[r - 1]);
#pragma unroll
            for (int r = 0; r < 8; ++r) lds_put(x1w + 8 * r, a[r]);
            wave_lds_fence();
#pragma unroll
            for (int b = 0; b < 8; ++b) a[b] = lds_get(x1r + b * kS1);
            wave_lds_fence();
            if (SG_FUSED_PRIO) __builtin_amdgcn_s_setprio(1);
            radix8(a);
#pragma unroll
            for (int s = 1; s < 8; ++s) a[s] = cmul(a[s], t2[s - 1]);
#pragma unroll
            for (int s = 0; s < 8; ++s) lds_put(x2w + 8 * s, a[s]);
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = lds_get(x2r + j * kS2);
            wave_lds_fence();
            if (SG_FUSED_PRIO) __builtin_amdgcn_s_setprio(2);
            radix8(a);
#pragma unroll
            for (int t = 4; t < 8; ++t) lds_put(x3w + 64 * t, a[t]);
            if (lane == 0) lds_put(buf + kM, a[0]);
            wave_lds_fence();
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float2 A = a[m];
                const float2 B = lds_get(x3b - 64 * m);
                const float2 cs = t3[m];
                const float2 S = make_float2(A.x + B.x, A.y - B.y);
                const float2 D = make_float2(A.x - B.x, A.y + B.y);
                const float2 T = make_float2(fmaf(cs.y, D.x, -cs.x * D.y), fmaf(cs.x, D.x, cs.y * D.y));
                const float2 Xk = csub(S, T), Xm = cadd(S, T);
                float pk = fmaf(Xk.x, Xk.x, Xk.y * Xk.y), pm = fmaf(Xm.x, Xm.x, Xm.y * Xm.y);
                if (m == 0) { pk *= r0; pm *= r0; }
                trow[lane + 64 * m] = pk;
                trow[kM - lane - 64 * m] = pm;
            }
            if (lane == 0) trow[256] = fmaf(a[4].x, a[4].x, a[4].y * a[4].y) * 4.0f;
            wave_lds_fence();
        }
        if (SG_FUSED_PRIO) __builtin_amdgcn_s_setprio(3);
        lds_barrier();                                              // tile complete

        // ---------------- phase 2: mel contraction on the matrix cores ----------------
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* const wrow = p.wt + static_cast<int64_t>(mi) * p.k_pad + 4 * kq;
        for (int k0 = 16 * wave; k0 < p.k_pad; k0 += 64) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(tile + mi * kRow + k0 + 4 * kq);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (k0 >= p.k_lo[t] && k0 < p.k_hi[t]) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(wrow + static_cast<int64_t>(16 * t) * p.k_pad + k0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc[t], 0, 0, 0);
                }
            }
        }
        lds_barrier();                                              // every wave is done reading the tile
        f32x4* const part = reinterpret_cast<f32x4*>(tile);           // [wave][t][lane] f32x4
#pragma unroll
        for (int t = 0; t < NT; ++t) part[(wave * NT + t) * 64 + lane] = acc[t];
        lds_barrier();
        // reduce the 4 partials; accumulator map: col = l&15 (mel), row = 4*(l>>4) + reg (frame)
        for (int o = threadIdx.x; o < 16 * 16 * NT; o += 256) {
            const int row = o / (16 * NT), col = o - row * (16 * NT);
            const int t = col >> 4, l = (col & 15) + 16 * (row >> 2), r = row & 3;
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) v += tile[((ww * NT + t) * 64 + l) * 4 + r];
            const int f = ft0 + row;
            if (col < p.n_mels && f < p.n_frames) {
                if (p.log_scale) v = 10.0f * log10f(fmaxf(v, 1e-10f));
                p.out[static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(f) * p.n_mels + col] = v;
            }
        }
        lds_barrier();                                              // the next tile overwrites the aliased region
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised form.  LDS (dynamic): slabs[8] (36 KiB) | tile[2] (65 KiB) | part[4][NT][64] float4 (NT KiB x 4).
// Round r of a workgroup (r = 0 .. n_tiles):   producers: tile r -> tile[r & 1]          (skipped in the last round)
//                                               consumers: tile r-1 from tile[(r-1) & 1]  (skipped in round 0)
//   B1 (about a third into the producers' round): every consumer has finished reading the partial sums of tile r-2
//   B2 (end of round): tile r complete, partial sums of tile r-1 complete.
// A consumer's round: reduce + store tile r-2 | B1 | MFMA over its k blocks of tile r-1, partial sums to LDS | B2.
constexpr int kProd = 8;

// CONS consumer waves: 8 (two per SIMD, 16 waves in all) while their partial sums fit the LDS (NT <= 5), else 4.
template <bool DETREND, int NT, int H, int CONS>
__global__ __launch_bounds__(64 * (kProd + CONS)) void stft1024_mel_ws_kernel(const FusedParams p) {
    constexpr int kCons = CONS;
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];
    float2* const slabs = reinterpret_cast<float2*>(ws_lds);
    float* const tiles = reinterpret_cast<float*>(ws_lds + kProd * kSlab * sizeof(float2));
    f32x4* const part = reinterpret_cast<f32x4*>(ws_lds + kProd * kSlab * sizeof(float2) + 2 * kTileFloats * sizeof(float));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    if (wg >= p.n_wgs) return;
    const int64_t tile_begin = p.total_tiles * wg / p.n_wgs;
    const int n_tiles = static_cast<int>(p.total_tiles * (wg + 1) / p.n_wgs - tile_begin);

    if (wave < kProd) {
        // ================================ producer: two frames of every tile ================================
        float2* const buf = slabs + wave * kSlab;
        float2 w[8], t1[7], t2[7], t3[4];
#pragma unroll
        for (int a = 0; a < 8; ++a) w[a] = p.win2[lane + 64 * a];
#pragma unroll
        for (int r = 0; r < 7; ++r) { t1[r] = p.tw[r * 64 + lane]; t2[r] = p.tw[(7 + r) * 64 + lane]; }
#pragma unroll
        for (int m = 0; m < 4; ++m) t3[m] = p.tw[(14 + m) * 64 + lane];
        const int j0 = lane & 7, hi = lane >> 3;
        float2* const x1w = buf + hi * kS1 + j0;
        float2* const x1r = buf + lane;
        float2* const x2w = buf + j0 * kS2 + hi;
        float2* const x2r = buf + lane;
        float2* const x3w = buf + lane;
        const float2* const x3b = buf + (kM - lane);
        {
            const float sq = sqrtf(p.scale * 0.5f);      // sqrt of the PSD scale rides on the window registers (stft_r8x3.hip)
#pragma unroll
            for (int a = 0; a < 8; ++a) { w[a].x *= sq; w[a].y *= sq; }
        }
        const float r0 = lane == 0 ? 0.5f : 1.0f;

        // where frame `i` (0 / 1) of this wave in tile `t` lives; n = frames of it that exist (0..2)
        auto locate = [&](int t, const float*& src, int& n) {
            const int64_t gt = tile_begin + t;
            const int clip = static_cast<int>(gt / p.tiles_per_clip);
            const int f0 = static_cast<int>(gt - static_cast<int64_t>(clip) * p.tiles_per_clip) * 16 + 2 * wave;
            n = max(0, min(2, p.n_frames - f0));
            src = p.x + static_cast<int64_t>(clip) * p.clip_stride + 2 * lane + static_cast<int64_t>(f0) * p.hop;
        };

        // FFT of the windowed frame in a[] -> PSD row `trow`; `mid_barrier`: join B1 between pass 2 and pass 3
        auto process = [&](float2 (&a)[8], float* trow, bool mid_barrier) {
            if (SG_FUSED_PRIO) __builtin_amdgcn_s_setprio(0);
            radix8(a);
#pragma unroll
            for (int r = 1; r < 8; ++r) a[r] = cmul(a[r], t1[r - 1]);
#pragma unroll
            for (int r = 0; r < 8; ++r) lds_put(x1w + 8 * r, a[r]);
            wave_lds_fence();
#pragma unroll
            for (int b = 0; b < 8; ++b) a[b] = lds_get(x1r + b * kS1);
            wave_lds_fence();
            if (SG_FUSED_PRIO) __builtin_amdgcn_s_setprio(1);
            radix8(a);
#pragma unroll
            for (int s = 1; s < 8; ++s) a[s] = cmul(a[s], t2[s - 1]);
#pragma unroll
            for (int s = 0; s < 8; ++s) lds_put(x2w + 8 * s, a[s]);
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = lds_get(x2r + j * kS2);
            wave_lds_fence();
            if (mid_barrier) lds_barrier();                              // B1
            if (SG_FUSED_PRIO) __builtin_amdgcn_s_setprio(2);
            radix8(a);
#pragma unroll
            for (int t = 4; t < 8; ++t) lds_put(x3w + 64 * t, a[t]);
            lds_put(buf + kM + lane, a[0]);
            wave_lds_fence();
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float2 A = a[m];
                const float2 B = lds_get(x3b - 64 * m);
                const float2 cs = t3[m];
                const float2 S = make_float2(A.x + B.x, A.y - B.y);
                const float2 D = make_float2(A.x - B.x, A.y + B.y);
                const float2 T = make_float2(fmaf(cs.y, D.x, -cs.x * D.y), fmaf(cs.x, D.x, cs.y * D.y));
                const float2 Xk = csub(S, T), Xm = cadd(S, T);
                float pk = fmaf(Xk.x, Xk.x, Xk.y * Xk.y), pm = fmaf(Xm.x, Xm.x, Xm.y * Xm.y);
                if (m == 0) { pk *= r0; pm *= r0; }
                trow[lane + 64 * m] = pk;
                trow[kM - lane - 64 * m] = pm;
            }
            {
                const float zx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[4].x), 0));
                const float zy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[4].y), 0));
                trow[256] = fmaf(zx, zx, zy * zy) * 4.0f;                   // wave-uniform value, every lane stores it
            }
            wave_lds_fence();
        };
        auto prep = [&](float2 (&a)[8]) {
            if (DETREND) {
                float s = a[0].x + a[0].y;
#pragma unroll
                for (int k = 1; k < 8; ++k) s += a[k].x + a[k].y;
                const float mean = wave_sum(s) * (1.0f / kN);
#pragma unroll
                for (int k = 0; k < 8; ++k) { a[k].x -= mean; a[k].y -= mean; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { a[k].x *= w[k].x; a[k].y *= w[k].y; }
        };
        auto zero_row = [&](float* trow) {
#pragma unroll
            for (int m = 0; m < 4; ++m) { trow[lane + 64 * m] = 0.f; trow[kM - lane - 64 * m] = 0.f; }
            trow[256] = 0.f;
        };

        // Prefetches are UNCONDITIONAL loads from an address that is always valid (the clip start when there is nothing to
        // fetch): a load under a wave-uniform `if` makes the compiler merge "loaded" and "not loaded" register sets with
        // copies right behind the load -- i.e. an s_waitcnt vmcnt(0) that exposes the whole HBM latency once per tile.
        const float* const safe = p.x + 2 * lane;
        float2 raw[8];
        const float* src = safe;
        int n_my = 0;
        if (n_tiles > 0) locate(0, src, n_my);
        {
            const float* const from = n_my > 0 ? src : safe;
#pragma unroll
            for (int k = 0; k < 8; ++k) raw[k] = *reinterpret_cast<const float2*>(from + 128 * k);
        }
        for (int r = 0; r <= n_tiles; ++r) {
            if (r == n_tiles) {                       // last round: only the consumers have work
                lds_barrier();                        // B1
                lds_barrier();                        // B2
                break;
            }
            float* const tile = tiles + (r & 1) * kTileFloats;
            float* const row0 = tile + (2 * wave) * kRow;
            // the pad columns 513..519 of this wave's two rows and, once, the 16 floats behind the last row: phase 2 reads k
            // up to k_pad-1 = 527 against zero weights, and 0 * garbage must stay finite
            if (lane < 14) row0[(lane / 7) * kRow + 513 + lane % 7] = 0.f;
            if (wave == kProd - 1 && lane < 16) tile[16 * kRow + lane] = 0.f;

            // ---- frame 0 of this wave (B1 sits inside it) ----
            float2 a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = raw[k];
            {   // prefetch frame 1 (or, harmlessly, frame 0 again)
                const float* const nxt = n_my > 1 ? src + p.hop : (n_my > 0 ? src : safe);
                if (H > 0) {
#pragma unroll
                    for (int k = 0; k + H < 8; ++k) raw[k] = raw[k + H];
#pragma unroll
                    for (int k = (H > 0 ? 8 - H : 0); k < 8; ++k) raw[k] = *reinterpret_cast<const float2*>(nxt + 128 * k);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) raw[k] = *reinterpret_cast<const float2*>(nxt + 128 * k);
                }
            }
            if (n_my > 0 && !(p.debug & 2)) {
                prep(a);
                process(a, row0, true);
            } else {
                zero_row(row0);
                lds_barrier();                        // B1
            }
            // ---- frame 1 ----
            const bool have1 = n_my > 1;
            float2 a1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a1[k] = raw[k];
            // prefetch this wave's first frame of the next tile (its samples are 16 frames on, or in the next clip)
            int n_next = 0;
            const float* src_next = safe;
            if (r + 1 < n_tiles) locate(r + 1, src_next, n_next);
            {
                const float* const from = n_next > 0 ? src_next : safe;
#pragma unroll
                for (int k = 0; k < 8; ++k) raw[k] = *reinterpret_cast<const float2*>(from + 128 * k);
            }
            if (have1 && !(p.debug & 2)) {
                prep(a1);
                process(a1, row0 + kRow, false);
            } else {
                zero_row(row0 + kRow);
            }
            src = src_next;
            n_my = n_next;
            if (SG_FUSED_PRIO) __builtin_amdgcn_s_setprio(0);
            lds_barrier();                            // B2
        }
    } else {
        // ================================ consumer: mel contraction of the previous tile ================================
        const int cw = wave - kProd;                  // 0..3: k blocks cw, cw + 4, ...
        const int ctid = threadIdx.x - 64 * kProd;    // 0..255 among the consumers
        const int mi = lane & 15, kq = lane >> 4;
        if (SG_FUSED_PRIO) __builtin_amdgcn_s_setprio(3);
        const float* const wrow = p.wt + static_cast<int64_t>(mi) * p.k_pad + 4 * kq;

        // sum the four partial accumulators of tile t and store its [16][n_mels] rows
        auto reduce_store = [&](int t_idx) {
            const int64_t gt = tile_begin + t_idx;
            const int clip = static_cast<int>(gt / p.tiles_per_clip);
            const int ft0 = static_cast<int>(gt - static_cast<int64_t>(clip) * p.tiles_per_clip) * 16;
            const float* const pf = reinterpret_cast<const float*>(part);
            // accumulator map: col = l & 15 (mel), row = 4 * (l >> 4) + reg (frame)
            for (int o = ctid; o < 16 * 16 * NT; o += 64 * kCons) {
                const int row = o / (16 * NT), col = o - row * (16 * NT);
                const int t = col >> 4, l = (col & 15) + 16 * (row >> 2), rr = row & 3;
                float v = 0.f;
#pragma unroll
                for (int ww = 0; ww < kCons; ++ww) v += pf[((ww * NT + t) * 64 + l) * 4 + rr];
                const int f = ft0 + row;
                if (col < p.n_mels && f < p.n_frames) {
                    if (p.log_scale) v = 10.0f * log10f(fmaxf(v, 1e-10f));
                    p.out[static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(f) * p.n_mels + col] = v;
                }
            }
        };

        for (int r = 0; r <= n_tiles; ++r) {
            if (r >= 2) reduce_store(r - 2);
            lds_barrier();                          // B1: the partial sums may be overwritten
            if (r >= 1 && !(p.debug & 1)) {
                const float* const tile = tiles + ((r - 1) & 1) * kTileFloats;
                f32x4 acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                // the weights come from L2 (39 KiB per bank: neither LDS nor L1 has room beside the tiles): the loads of k block
                // j + 1 are issued before the MFMAs of block j
                f32x4 bvn[NT];
                auto fetch = [&](int k0) {          // unconditional loads (see the producers' prefetch): an inactive (k block,
#pragma unroll                                    // tile) pair reads one fixed, cache-resident float4 instead
                    for (int t = 0; t < NT; ++t) {
                        const bool active = k0 < p.k_pad && k0 >= p.k_lo[t] && k0 < p.k_hi[t];
                        const float* const src = active ? wrow + static_cast<int64_t>(16 * t) * p.k_pad + k0 : p.wt + 4 * kq;
                        bvn[t] = *reinterpret_cast<const f32x4*>(src);
                    }
                };
                fetch(16 * cw);
                for (int k0 = 16 * cw; k0 < p.k_pad; k0 += 16 * kCons) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(tile + mi * kRow + k0 + 4 * kq);
                    f32x4 bv[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) bv[t] = bvn[t];
                    fetch(k0 + 16 * kCons);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (k0 >= p.k_lo[t] && k0 < p.k_hi[t]) {
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[t][0], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[t][1], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[t][2], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[t][3], acc[t], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) part[(cw * NT + t) * 64 + lane] = acc[t];
            }
            lds_barrier();                          // B2
        }
        // the last tile's partial sums are complete after the final B2
        if (n_tiles >= 1) reduce_store(n_tiles - 1);
    }
}

template <bool DETREND, int NT, int H, int CONS>
int launch_ws_one(const FusedParams& prm, int n_wg, hipStream_t s) {
    const size_t lds = kProd * kSlab * sizeof(float2) + 2 * kTileFloats * sizeof(float) + static_cast<size_t>(CONS) * NT * 64 * sizeof(f32x4);
    auto kern = stft1024_mel_ws_kernel<DETREND, NT, H, CONS>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(stft1024_mel_ws)");
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * (kProd + CONS)), lds, s, prm);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft1024_mel_ws launch");
}

template <bool DETREND, int H>
int launch_ws_nt(const FusedParams& prm, int nt, int n_wg, hipStream_t s) {
    const int cons = (prm.form & SG_MEL_FORM_CONS4) ? 4 : 8;
    switch (nt) {
        case 1: return cons == 8 ? launch_ws_one<DETREND, 1, H, 8>(prm, n_wg, s) : launch_ws_one<DETREND, 1, H, 4>(prm, n_wg, s);
        case 2: return cons == 8 ? launch_ws_one<DETREND, 2, H, 8>(prm, n_wg, s) : launch_ws_one<DETREND, 2, H, 4>(prm, n_wg, s);
        case 3: return cons == 8 ? launch_ws_one<DETREND, 3, H, 8>(prm, n_wg, s) : launch_ws_one<DETREND, 3, H, 4>(prm, n_wg, s);
        case 4: return cons == 8 ? launch_ws_one<DETREND, 4, H, 8>(prm, n_wg, s) : launch_ws_one<DETREND, 4, H, 4>(prm, n_wg, s);
        case 5: return cons == 8 ? launch_ws_one<DETREND, 5, H, 8>(prm, n_wg, s) : launch_ws_one<DETREND, 5, H, 4>(prm, n_wg, s);
        case 6: return launch_ws_one<DETREND, 6, H, 4>(prm, n_wg, s);       // 8 consumers' partial sums no longer fit the LDS
        case 7: return launch_ws_one<DETREND, 7, H, 4>(prm, n_wg, s);
        default: return launch_ws_one<DETREND, 8, H, 4>(prm, n_wg, s);
    }
}

template <bool DETREND, int H>
int launch_nt(const FusedParams& prm, int nt, int n_wg, hipStream_t s) {
#define SG_FUSED_CASE(N) case N: hipLaunchKernelGGL((stft1024_mel_kernel<DETREND, N, H>), dim3(n_wg), dim3(256), 0, s, prm); break;
    switch (nt) {
        SG_FUSED_CASE(1) SG_FUSED_CASE(2) SG_FUSED_CASE(3) SG_FUSED_CASE(4)
        SG_FUSED_CASE(5) SG_FUSED_CASE(6) SG_FUSED_CASE(7) default: SG_FUSED_CASE(8)
    }
#undef SG_FUSED_CASE
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft1024_mel launch");
}

}  // namespace
}  // namespace sg

using namespace sg;

extern "C" int sg_stft_mel(const sg_plan* plan, const float* x_dev, int64_t n_samples, int64_t clip_stride, int n_clips,
                           const float* packed_weights_dev, int weights_n_bins, int n_mels, const int* tile_k_lo,
                           const int* tile_k_hi, int log_scale, float* mel_dev, int64_t out_clip_stride, void* stream) {
    if (!plan || !x_dev || !packed_weights_dev || !mel_dev) { set_error("null pointer"); return SG_ERR_ARG; }
    if (plan->kernel != Kernel::R8X3 || plan->mode != SG_MODE_PSD) {
        set_error("fused STFT+mel needs an f32 nperseg = nfft = 1024 PSD plan (kernel r8x3)");
        return SG_ERR_UNSUPPORTED;
    }
    if (n_mels < 1 || n_mels > 16 * kMaxTiles || n_clips < 0 || n_samples < 0) { set_error("bad sizes"); return SG_ERR_ARG; }
    if (weights_n_bins != plan->nfft / 2 + 1) {
        // the kernel indexes the packed bank with the plan's row pitch: a bank built for another nfft would be read out of bounds
        set_error("mel bank was packed for %d bins, the plan has %d", weights_n_bins, plan->nfft / 2 + 1);
        return SG_ERR_ARG;
    }
    if ((plan->hop & 1) || (clip_stride & 1) || (reinterpret_cast<uintptr_t>(x_dev) & 7)) {
        set_error("fused STFT+mel needs an even hop / clip stride and 8-byte aligned input");
        return SG_ERR_UNSUPPORTED;
    }
    const int64_t n_frames = n_samples < plan->nperseg ? 0 : (n_samples - plan->nperseg) / plan->hop + 1;
    if (n_frames == 0 || n_clips == 0) return SG_OK;
    if (n_frames > INT32_MAX - 16) { set_error("fused STFT+mel: more than 2^31 frames per clip"); return SG_ERR_ARG; }
    if (n_clips > 1 && (clip_stride < n_samples || out_clip_stride < n_frames * n_mels)) { set_error("bad strides"); return SG_ERR_ARG; }
    FusedParams prm{};
    prm.x = x_dev; prm.clip_stride = clip_stride; prm.n_frames = static_cast<int>(n_frames); prm.hop = plan->hop;
    prm.tiles_per_clip = static_cast<int>((n_frames + 15) / 16);
    prm.total_tiles = static_cast<int64_t>(prm.tiles_per_clip) * n_clips;
    const bool use_v1 = !(log_scale & SG_MEL_FORM_WS);            // default kernel; SG_MEL_FORM_WS picks the wave-specialised form
    int64_t n_wgs = static_cast<int64_t>(plan->n_cu) * (use_v1 ? 3 : 1);      // ws: 133-137 KiB of LDS = one workgroup per CU
    if (n_wgs > prm.total_tiles) n_wgs = prm.total_tiles;
    prm.n_wgs = static_cast<int>(n_wgs);
    prm.out = mel_dev; prm.out_clip_stride = out_clip_stride;
    prm.win2 = static_cast<const float2*>(plan->win_dev);
    prm.tw = static_cast<const float2*>(plan->r8_tw_dev);
    prm.scale = static_cast<float>(plan->scale);
    prm.wt = packed_weights_dev;
    prm.k_pad = (plan->nfft / 2 + 1 + 15) & ~15;
    prm.n_mels = n_mels; prm.log_scale = (log_scale & SG_MEL_LOG) != 0; prm.form = log_scale;
    if (const char* e = SG_TUNE_ENV("SPECTRO_FUSED_DEBUG")) prm.debug = atoi(e);
    const int nt = (n_mels + 15) / 16;
    for (int t = 0; t < nt; ++t) {
        prm.k_lo[t] = tile_k_lo ? tile_k_lo[t] & ~15 : 0;
        prm.k_hi[t] = tile_k_hi ? (tile_k_hi[t] + 15) & ~15 : prm.k_pad;
        if (prm.k_lo[t] < 0) prm.k_lo[t] = 0;
        if (prm.k_hi[t] > prm.k_pad) prm.k_hi[t] = prm.k_pad;
    }
    auto s = static_cast<hipStream_t>(stream);
    const bool det = plan->detrend == SG_DETREND_CONSTANT;
    if (use_v1) {
        if (plan->hop == 256) return det ? launch_nt<true, 2>(prm, nt, prm.n_wgs, s) : launch_nt<false, 2>(prm, nt, prm.n_wgs, s);
        return det ? launch_nt<true, 0>(prm, nt, prm.n_wgs, s) : launch_nt<false, 0>(prm, nt, prm.n_wgs, s);
    }
    if (plan->hop == 256) return det ? launch_ws_nt<true, 2>(prm, nt, prm.n_wgs, s) : launch_ws_nt<false, 2>(prm, nt, prm.n_wgs, s);
    return det ? launch_ws_nt<true, 0>(prm, nt, prm.n_wgs, s) : launch_ws_nt<false, 0>(prm, nt, prm.n_wgs, s);
}
