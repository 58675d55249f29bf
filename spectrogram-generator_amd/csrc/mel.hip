// mel.hip -- mel filterbank epilogue (BASELINE cfg3).  NOT part of the reference ("no mel
// anywhere"): the filterbank definition below is this library's own and is verified only against its
// numpy restatement (oracle/mel_oracle.py) -- parity with the reference is "unpinned" for this stage.
//
// mel[f][m] = sum_k S[f][k] * W[k][m] is a dense contraction, so it runs on the matrix cores with the exact
// f32 MFMA (v_mfma_f32_16x16x4_f32: bit-for-bit an f32 fmaf chain, no TF32-style truncation on gfx950).
// Triangular filters make W block-sparse: for each tile of 16 mel bands only the k-range its triangles cover
// is multiplied (ranges from sg_mel_tile_ranges), which cuts the MFMA count ~4x versus the dense 513x80 product.
// HBM-bound by the spectrum read: 2052 B in + n_mels*4 B out per frame at nfft = 1024.
#include "spectro_internal.h"

#include <cmath>

namespace sg {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxTiles = 8;     // n_mels <= 128

struct MelParams {
    const float* spec;       // [n_frames][n_bins]
    int64_t n_frames;
    int n_bins, n_mels, n_tiles, log_scale;
    const float* wt;         // transposed, padded: [16*n_tiles][k_pad]  (k contiguous, zero padded)
    int k_pad;               // n_bins rounded up to a multiple of 16
    float* out;              // [n_frames][n_mels]
    int k_lo[kMaxTiles], k_hi[kMaxTiles];   // per mel tile: k range (multiples of 16, hi exclusive) with non-zero weights
    // LDSW kernels: the non-zero (mel tile, k block) pairs, 1 KiB of weights each, live in LDS for the whole kernel
    int n_pairs;
    int pair_base[kMaxTiles];               // tile t's pairs are pair_base[t] + (k0 - k_lo[t]) / 16
    int64_t n_tiles16;                      // ceil(n_frames / 16)
    int64_t n_waves;                        // persistent grid: waves striding over the 16-frame tiles
};
constexpr int kMaxPairs = 63;               // 63 KiB of LDS

// One wavefront per 16 consecutive frames.  The contraction index is consumed 16 at a time: lane (i = l&15,
// kq = l>>4) loads the spectrum row f0+i at k = 16j + 4kq .. +3 ONCE and, for every mel tile whose triangles
// cover this k block, ONE float4 of the transposed weights of mel column 16t+i at the same k; MFMA step e
// multiplies element e of both.  Any bijection between (step, kq) and k is a valid order for the sum as long as
// A and B agree, so no cross-lane shuffle is needed.  All NT tile accumulators stay live, so the spectrum is read
// from HBM exactly once.  Accumulator map: col = l&15 (mel), row = 4*(l>>4) + reg (frame).
// LDSW: persistent workgroups keep the block-sparse weights in LDS (loaded once per workgroup) and stride over the
// 16-frame tiles; otherwise (a bank too dense for LDS) one wave per tile with the weights read through L1/L2 -- those
// reads were as many bytes as the spectrum itself and cost 30 of 113 us on the cfg3 batch.
template <int NT, bool LDSW>
__global__ __launch_bounds__(256) void mel_kernel(const MelParams p) {
    extern __shared__ __attribute__((aligned(16))) float wl[];           // [pair][lane][4]
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    if (LDSW) {
        for (int idx = threadIdx.x; idx < p.n_pairs * 64; idx += 256) {
            const int pair = idx >> 6, l = idx & 63;
            int t = 0;
#pragma unroll
            for (int tt = 1; tt < NT; ++tt) if (pair >= p.pair_base[tt]) t = tt;
            const int k0 = p.k_lo[t] + 16 * (pair - p.pair_base[t]);
            *reinterpret_cast<f32x4*>(wl + idx * 4) =
                *reinterpret_cast<const f32x4*>(p.wt + static_cast<int64_t>(16 * t + (l & 15)) * p.k_pad + k0 + 4 * (l >> 4));
        }
        __syncthreads();
    }
    const float* const wrow = p.wt + static_cast<int64_t>(i) * p.k_pad + 4 * kq;
    const int k_last = p.n_bins - 1 - 4 * kq;            // last valid index relative to arow
    const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    for (int64_t tile = wave0; tile < p.n_tiles16; tile += p.n_waves) {
    const int64_t f0 = tile * 16;
    int64_t fa = f0 + i;
    if (fa >= p.n_frames) fa = p.n_frames - 1;           // clamp: rows past the end are computed but not stored
    const float* const arow = p.spec + fa * p.n_bins + 4 * kq;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int k_full = (p.n_bins - 16) & ~15;            // blocks below this never touch the row end
    auto load_a = [&](int k0) -> f32x4 {
        if (k0 < k_full) return f32x4{arow[k0], arow[k0 + 1], arow[k0 + 2], arow[k0 + 3]};   // rows are only 4-byte aligned (n_bins odd)
        // tail of the row (and the look-ahead past it): clamp -- the padded weights are zero there
        return f32x4{arow[min(k0, k_last)], arow[min(k0 + 1, k_last)], arow[min(k0 + 2, k_last)], arow[min(k0 + 3, k_last)]};
    };
    // The spectrum is streamed kAhead k-blocks ahead of the MFMAs (two register sets).  Measured on the cfg3 batch with
    // the weights in LDS: look-ahead 1 / 2 / 4 / 8 blocks = 84 / 78 / 104 / 160 us (deeper queues of 4-byte gathers
    // thrash the L1) -- 2 it is; the one-wave-per-tile form already has 8 waves/SIMD of loads in flight and uses none.
    constexpr int kAhead = LDSW ? 2 : 1;
    f32x4 nxt[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) nxt[u] = LDSW ? load_a(16 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kg = 0; kg < p.k_pad; kg += 16 * kAhead) {
        f32x4 cur[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) cur[u] = LDSW ? nxt[u] : load_a(kg + 16 * u);
        if (LDSW) {
#pragma unroll
            for (int u = 0; u < kAhead; ++u) nxt[u] = load_a(kg + 16 * (kAhead + u));
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int k0 = kg + 16 * u;
            if (k0 >= p.k_pad) break;
            const f32x4 a = cur[u];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (k0 >= p.k_lo[t] && k0 < p.k_hi[t]) {      // wave-uniform: block sparsity of the triangular bank
                    const f32x4 b = LDSW ? *reinterpret_cast<const f32x4*>(wl + ((p.pair_base[t] + ((k0 - p.k_lo[t]) >> 4)) * 64 + lane) * 4)
                                         : *reinterpret_cast<const f32x4*>(wrow + static_cast<int64_t>(16 * t) * p.k_pad + k0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = 16 * t + i;
        if (col < p.n_mels) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t f = f0 + 4 * kq + r;
                if (f < p.n_frames) {
                    float v = acc[t][r];
                    if (p.log_scale) v = 10.0f * log10f(fmaxf(v, 1e-10f));
                    p.out[f * p.n_mels + col] = v;
                }
            }
        }
    }
    }
}

template <int NT>
void launch_mel(const MelParams& p, unsigned grid, bool ldsw, hipStream_t s) {
    if (ldsw) hipLaunchKernelGGL((mel_kernel<NT, true>), dim3(grid), dim3(256), static_cast<size_t>(p.n_pairs) * 1024, s, p);
    else hipLaunchKernelGGL((mel_kernel<NT, false>), dim3(grid), dim3(256), 0, s, p);
}

// ---------------------------------------------------------------------------------------------------------------------
// Band-sparse form (sg_mel_sparse_pack, host_shim.cpp): a triangular bank has two non-zero weights per bin, so a row's mel
// spectrum is ~2 n_bins multiply-adds.  One wavefront per row, rows strided over a persistent grid: the row is fetched with
// coalesced 4-byte loads one row ahead, parked in the wave's LDS row, every lane gathers the <= 8 bins of its work items against
// weights held in registers, the lane that owns a band adds the band's (adjacent) partial sums.  HBM-read bound.
struct MelSparseParams {
    const float* spec;
    int64_t n_frames;
    int n_bins, n_mels, log_scale;
    const int* start; const float* w; const int* first; const int* count;
    float* out;
};
constexpr int kSparseWaves = 4;

template <int IPL>
__global__ __launch_bounds__(64 * kSparseWaves) void mel_sparse_kernel(const MelSparseParams p) {
    extern __shared__ __attribute__((aligned(16))) float sl[];          // per wave: row[row_len] | part[64 * IPL]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row_len = (p.n_bins + 8 + 15) & ~15;                       // a last item may read 7 slots past the last bin
    float* const row = sl + wave * (row_len + 64 * IPL);
    float* const part = row + row_len;
    const int per_lane = (p.n_bins + 63) / 64;                            // bins lane, lane + 64, ... (<= 33 for nfft 4096)

    float mw[IPL][8];
    const float* gather[IPL];
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        gather[i] = row + p.start[lane + 64 * i];
#pragma unroll
        for (int c = 0; c < 8; ++c) mw[i][c] = p.w[(c * IPL + i) * 64 + lane];
    }
    int bfirst[2] = {0, 0}, bcount[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        if (j < p.n_mels) { bfirst[q] = p.first[j]; bcount[q] = p.count[j]; }
    }
    for (int k = p.n_bins + lane; k < row_len; k += 64) row[k] = 0.f;     // finite slots behind the last bin (zero weights)

    const int64_t n_waves = static_cast<int64_t>(gridDim.x) * kSparseWaves;
    int64_t f = static_cast<int64_t>(blockIdx.x) * kSparseWaves + wave;
    constexpr int kMaxPer = 9;                                            // register-resident prefetch for nfft <= 1024 (9 x 64 >= 513)
    float nxt[kMaxPer];
    const bool reg_path = per_lane <= kMaxPer;
    auto fetch = [&](int64_t r) {
        const float* const src = p.spec + (r < p.n_frames ? r : p.n_frames - 1) * p.n_bins;      // unconditional, always valid
#pragma unroll
        for (int m = 0; m < kMaxPer; ++m) { const int k = lane + 64 * m; nxt[m] = src[k < p.n_bins ? k : p.n_bins - 1]; }
    };
    if (reg_path && f < p.n_frames) fetch(f);
    for (; f < p.n_frames; f += n_waves) {
        if (reg_path) {
#pragma unroll
            for (int m = 0; m < kMaxPer; ++m) { const int k = lane + 64 * m; if (k < p.n_bins) row[k] = nxt[m]; }
            fetch(f + n_waves);
        } else {
            const float* const src = p.spec + f * p.n_bins;
            for (int k = lane; k < p.n_bins; k += 64) row[k] = src[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < IPL; ++i) {
            float acc = mw[i][0] * gather[i][0];
#pragma unroll
            for (int c = 1; c < 8; ++c) acc = fmaf(mw[i][c], gather[i][c], acc);
            part[lane + 64 * i] = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* const orow = p.out + f * p.n_mels;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q == 1 && p.n_mels <= 64) break;
            float v = 0.f;
            for (int t = 0; t < bcount[q]; ++t) v += part[bfirst[q] + t];
            if (p.log_scale) v = 3.01029995663981195f * __log2f(fmaxf(v, 1e-10f));
            if (lane + 64 * q < p.n_mels) orow[lane + 64 * q] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int IPL>
int launch_sparse(const MelSparseParams& prm, int n_cu, hipStream_t s) {
    const int row_len = (prm.n_bins + 8 + 15) & ~15;
    const size_t lds = static_cast<size_t>(kSparseWaves) * (row_len + 64 * IPL) * sizeof(float);
    int64_t n_wg = (prm.n_frames + kSparseWaves - 1) / kSparseWaves;
    const int64_t cap = static_cast<int64_t>(n_cu) * 8;                   // 32 waves per CU
    if (n_wg > cap) n_wg = cap;
    auto kern = mel_sparse_kernel<IPL>;
    if (lds > 64 * 1024) return SG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(n_wg)), dim3(64 * kSparseWaves), lds, s, prm);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "mel_sparse launch");
}

}  // namespace
}  // namespace sg

using namespace sg;

extern "C" {

int sg_mel_sparse(const float* spec_dev, int64_t n_frames, int n_bins, const int32_t* item_start_dev, const float* item_w_dev,
                  const int32_t* band_first_dev, const int32_t* band_count_dev, int items_per_lane, int n_mels, int log_scale,
                  float* mel_dev, void* stream) {
    if (n_frames < 0 || n_bins < 1 || n_bins > 8193 || n_mels < 1 || n_mels > 128 || items_per_lane < 1 || items_per_lane > 4) {
        set_error("sg_mel_sparse: bad sizes");
        return SG_ERR_ARG;
    }
    if (n_frames == 0) return SG_OK;
    if (!spec_dev || !mel_dev || !item_start_dev || !item_w_dev || !band_first_dev || !band_count_dev) { set_error("null device pointer"); return SG_ERR_ARG; }
    MelSparseParams prm{spec_dev, n_frames, n_bins, n_mels, log_scale, item_start_dev, item_w_dev, band_first_dev, band_count_dev, mel_dev};
    const int n_cu = device_cu_count();
    auto s = static_cast<hipStream_t>(stream);
    int rc;
    switch (items_per_lane) {
        case 1: rc = launch_sparse<1>(prm, n_cu, s); break;
        case 2: rc = launch_sparse<2>(prm, n_cu, s); break;
        case 3: rc = launch_sparse<3>(prm, n_cu, s); break;
        default: rc = launch_sparse<4>(prm, n_cu, s); break;
    }
    if (rc == SG_ERR_UNSUPPORTED) set_error("sg_mel_sparse: row too long for the LDS");
    return rc;
}

int sg_mel(const float* spec_dev, int64_t n_frames, int n_bins, const float* weights_dev, int n_mels,
           const int* tile_k_lo, const int* tile_k_hi, int log_scale, float* mel_dev, void* stream) {
    if (!spec_dev || !weights_dev || !mel_dev) { set_error("null pointer"); return SG_ERR_ARG; }
    if (n_frames < 0 || n_bins < 1 || n_mels < 1 || n_mels > 16 * kMaxTiles) { set_error("bad mel sizes (n_mels <= 128)"); return SG_ERR_ARG; }
    if (n_frames == 0) return SG_OK;
    auto s = static_cast<hipStream_t>(stream);
    MelParams p{};
    p.spec = spec_dev; p.n_frames = n_frames; p.n_bins = n_bins; p.n_mels = n_mels;
    p.n_tiles = (n_mels + 15) / 16; p.log_scale = log_scale; p.wt = weights_dev; p.out = mel_dev;
    p.k_pad = (n_bins + 15) & ~15;
    for (int t = 0; t < p.n_tiles; ++t) {
        p.k_lo[t] = tile_k_lo ? tile_k_lo[t] & ~15 : 0;
        p.k_hi[t] = tile_k_hi ? (tile_k_hi[t] + 15) & ~15 : p.k_pad;
        if (p.k_lo[t] < 0) p.k_lo[t] = 0;
        if (p.k_hi[t] > p.k_pad) p.k_hi[t] = p.k_pad;
    }
    const int64_t tiles = (n_frames + 15) / 16;
    p.n_tiles16 = tiles;
    int n_pairs = 0;
    for (int t = 0; t < p.n_tiles; ++t) { p.pair_base[t] = n_pairs; n_pairs += (p.k_hi[t] - p.k_lo[t]) / 16; }
    p.n_pairs = n_pairs;
    const bool ldsw = n_pairs >= 1 && n_pairs <= kMaxPairs;
    unsigned grid = static_cast<unsigned>((tiles + 3) / 4);
    if (ldsw) {                                          // persistent: as many workgroups as fit the CUs' LDS (160 KiB each)
        int dev = 0, n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        int per_cu = (160 * 1024) / (n_pairs * 1024);
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        const unsigned cap = static_cast<unsigned>(n_cu) * per_cu;
        if (grid > cap) grid = cap;
    }
    p.n_waves = static_cast<int64_t>(grid) * 4;
    switch (p.n_tiles) {
        case 1: launch_mel<1>(p, grid, ldsw, s); break;
        case 2: launch_mel<2>(p, grid, ldsw, s); break;
        case 3: launch_mel<3>(p, grid, ldsw, s); break;
        case 4: launch_mel<4>(p, grid, ldsw, s); break;
        case 5: launch_mel<5>(p, grid, ldsw, s); break;
        case 6: launch_mel<6>(p, grid, ldsw, s); break;
        case 7: launch_mel<7>(p, grid, ldsw, s); break;
        default: launch_mel<8>(p, grid, ldsw, s); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "mel launch");
    return SG_OK;
}

}  // extern "C"
