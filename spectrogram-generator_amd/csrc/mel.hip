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
    const float* w;          // [n_bins][n_mels]
    float* out;              // [n_frames][n_mels]
    int k_lo[kMaxTiles], k_hi[kMaxTiles];   // per mel tile: k range (multiples of 4, hi exclusive) with non-zero weights
};

// One wavefront per 16 consecutive frames.  A operand: S[f0 + (l&15)][k0 + (l>>4)], B operand:
// W[k0 + (l>>4)][16*t + (l&15)], accumulator: col = l&15 (mel), row = 4*(l>>4) + reg (frame).
__global__ __launch_bounds__(256) void mel_kernel(const MelParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int64_t f0 = tile * 16;
    if (f0 >= p.n_frames) return;
    const int i = lane & 15, kq = lane >> 4;
    int64_t fa = f0 + i;
    if (fa >= p.n_frames) fa = p.n_frames - 1;           // clamp: rows past the end are computed but not stored
    const float* const arow = p.spec + fa * p.n_bins;
    for (int t = 0; t < p.n_tiles; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int col = 16 * t + i;
        const bool col_ok = col < p.n_mels;
        for (int k0 = p.k_lo[t]; k0 < p.k_hi[t]; k0 += 4) {
            const int k = k0 + kq;
            const bool k_ok = k < p.n_bins;
            const float a = k_ok ? arow[k] : 0.f;
            const float b = (k_ok && col_ok) ? p.w[static_cast<int64_t>(k) * p.n_mels + col] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        if (col_ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t f = f0 + 4 * kq + r;
                if (f < p.n_frames) {
                    float v = acc[r];
                    if (p.log_scale) v = 10.0f * log10f(fmaxf(v, 1e-10f));
                    p.out[f * p.n_mels + col] = v;
                }
            }
        }
    }
}

}  // namespace
}  // namespace sg

using namespace sg;

extern "C" {

int sg_mel_weights(int nfft, double fs, int n_mels, double fmin, double fmax, double* weights_host) {
    if (!weights_host || nfft < 2 || n_mels < 1 || !(fs > 0) || !(fmax > fmin) || fmin < 0) {
        set_error("bad mel filterbank arguments");
        return SG_ERR_ARG;
    }
    const int n_bins = nfft / 2 + 1;
    auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
    auto mel2hz = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
    const double m_lo = hz2mel(fmin), m_hi = hz2mel(fmax);
    std::vector<double> edges(n_mels + 2);
    for (int j = 0; j < n_mels + 2; ++j) edges[j] = mel2hz(m_lo + (m_hi - m_lo) * j / (n_mels + 1));
    for (int k = 0; k < n_bins; ++k) {
        const double f = static_cast<double>(k) * fs / nfft;
        for (int m = 0; m < n_mels; ++m) {
            const double l = edges[m], c = edges[m + 1], r = edges[m + 2];
            const double up = (f - l) / (c - l), down = (r - f) / (r - c);
            const double v = up < down ? up : down;
            weights_host[static_cast<size_t>(k) * n_mels + m] = v > 0.0 ? v : 0.0;
        }
    }
    return SG_OK;
}

int sg_mel_tile_ranges(const double* weights_host, int n_bins, int n_mels, int* k_lo, int* k_hi) {
    if (!weights_host || !k_lo || !k_hi || n_bins < 1 || n_mels < 1) { set_error("bad argument"); return SG_ERR_ARG; }
    const int n_tiles = (n_mels + 15) / 16;
    for (int t = 0; t < n_tiles; ++t) {
        int lo = n_bins, hi = 0;
        for (int k = 0; k < n_bins; ++k)
            for (int m = 16 * t; m < 16 * t + 16 && m < n_mels; ++m)
                if (weights_host[static_cast<size_t>(k) * n_mels + m] != 0.0) { if (k < lo) lo = k; if (k + 1 > hi) hi = k + 1; }
        if (hi <= lo) { lo = 0; hi = 0; }
        k_lo[t] = lo & ~3;
        k_hi[t] = (hi + 3) & ~3;
    }
    return SG_OK;
}

int sg_mel(const float* spec_dev, int64_t n_frames, int n_bins, const float* weights_dev, int n_mels,
           const int* tile_k_lo, const int* tile_k_hi, int log_scale, float* mel_dev, void* stream) {
    if (!spec_dev || !weights_dev || !mel_dev) { set_error("null pointer"); return SG_ERR_ARG; }
    if (n_frames < 0 || n_bins < 1 || n_mels < 1 || n_mels > 16 * kMaxTiles) { set_error("bad mel sizes (n_mels <= 128)"); return SG_ERR_ARG; }
    if (n_frames == 0) return SG_OK;
    auto s = static_cast<hipStream_t>(stream);
    MelParams p{};
    p.spec = spec_dev; p.n_frames = n_frames; p.n_bins = n_bins; p.n_mels = n_mels;
    p.n_tiles = (n_mels + 15) / 16; p.log_scale = log_scale; p.w = weights_dev; p.out = mel_dev;
    for (int t = 0; t < p.n_tiles; ++t) {
        p.k_lo[t] = tile_k_lo ? tile_k_lo[t] & ~3 : 0;
        p.k_hi[t] = tile_k_hi ? (tile_k_hi[t] + 3) & ~3 : (n_bins + 3) & ~3;
        if (p.k_lo[t] < 0) p.k_lo[t] = 0;
    }
    const int64_t tiles = (n_frames + 15) / 16;
    hipLaunchKernelGGL(mel_kernel, dim3(static_cast<unsigned>((tiles + 3) / 4)), dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "mel launch");
    return SG_OK;
}

}  // extern "C"
