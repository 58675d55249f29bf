// stft_r8x3.hip -- the headline kernel: fused STFT -> one-sided PSD for nperseg = nfft = 1024, f32.
//
// Replaces, per frame, scipy's _fft_helper + PSD epilogue
// (scipy/signal/_spectral_py.py:2180-2202 framing/detrend/window/rfft, :2125-2134 |X|^2*scale
// with interior bins doubled) as called from the reference at PlotEngine.py:113 / :232.
//
// CDNA4 mapping (gfx950, wave64):
//   * one wavefront owns one frame at a time and walks FPW consecutive frames of one clip, so the
//     hop-overlapped samples of neighbouring frames are L1/L2 hits and HBM sees each sample once;
//   * the real 1024-point FFT is a 512-point complex FFT of z[n] = x[2n] + i*x[2n+1] (even/odd packing)
//     followed by a split pass.  512 = 8*8*8: every lane keeps 8 complex values in VGPRs and runs three
//     register-resident radix-8 butterflies; the two transposes in between and the k <-> 512-k pairing of the
//     split pass go through a 4.5 KiB per-wave LDS slab with padded strides (72 / 66 / 1 elements) that
//     make every ds_write_b64 / ds_read_b64 bank-conflict free (checked by tools/sim_r8x3.py);
//   * a wave only ever talks to its own slab, so there is not a single s_barrier in the kernel;
//   * window (16 floats/lane) and all twiddles (18 complex/lane) are loaded once per wave into VGPRs;
//   * the epilogue computes |X|^2 * scale (x2 for interior bins) and streams 513 contiguous floats per
//     frame; the spectrum is never staged in HBM in complex form.
//
// Algorithmic HBM bytes per frame: hop*4 read + 513*4 written (3076 B at hop = 256).
#include "spectro_internal.h"
#include "fft_wave.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

#ifndef SG_R8_PRIO
#define SG_R8_PRIO 1          // wave priority rises along a frame (0 pass 1, 1 pass 2, 2 pass 3, 3 split + stores + next prefetch); 0 = off
#endif

namespace sg {
namespace {

using namespace wavefft;

constexpr int kN = 1024;        // samples per frame
constexpr int kM = 512;         // complex points
constexpr int kBins = 513;
constexpr int kS1 = 72;         // LDS stride (elements) of exchange 1: [b][l2]
constexpr int kS2 = 66;         // LDS stride of exchange 2: [j0][l3]
constexpr int kSlab = 8 * kS1;  // 576 complex = 4608 B per wave
#ifndef SG_R8_WPW
#define SG_R8_WPW 4
#endif
constexpr int kWavesPerWg = SG_R8_WPW;
#ifndef SG_TW_LDS
// 0: twiddles in VGPRs (4 waves/SIMD); 1: in a workgroup-shared LDS table (5 waves/SIMD).  Sustained A/B on MI355X
// (bench.py, 400 steps): VGPR 96-97 us vs LDS 99.5 us per launch -- the 18 extra ds_read_b64 per frame cost more
// (LDS time and power) than the fifth wave buys, so VGPR is the default.
#define SG_TW_LDS 0
#endif
#ifndef SG_R8_OCC
// waves per SIMD the persistent grid fills.  The kernel needs 113 VGPRs, so four would fit; three are faster on every box
// tried in round 2 (same-box A/B of two builds, 4 repetitions: 85.3-88.1 us against 88.0-90.4 us; two waves 93 us;
// profiles/r02_ab_occupancy.txt): fewer, longer runs (39 frames instead of 29 per wave: fewer row streams for the memory system, fewer
// prologues, fewer halo re-reads) are worth more than the fourth wave's latency hiding.
#define SG_R8_OCC (SG_TW_LDS ? 5 : 3)
#endif
constexpr int kOccupancy = SG_R8_OCC;            // waves per SIMD the kernel is built for
constexpr int kMinRun = 4;       // shortest run of frames worth a wave's prologue

template <typename TIn, bool ALIGNED>
__device__ __forceinline__ float2 load_pair(const TIn* p);

template <>
__device__ __forceinline__ float2 load_pair<float, true>(const float* p) {
    return *reinterpret_cast<const float2*>(p);
}
template <>
__device__ __forceinline__ float2 load_pair<float, false>(const float* p) {
    return make_float2(p[0], p[1]);
}
template <>
__device__ __forceinline__ float2 load_pair<int16_t, true>(const int16_t* p) {
    const short2 s = *reinterpret_cast<const short2*>(p);
    return make_float2(static_cast<float>(s.x), static_cast<float>(s.y));
}
template <>
__device__ __forceinline__ float2 load_pair<int16_t, false>(const int16_t* p) {
    return make_float2(static_cast<float>(p[0]), static_cast<float>(p[1]));
}

struct R8Params {
    const void* x;
    int64_t clip_stride;
    int n_frames;          // per clip
    int hop;
    int sub;               // hop * sub == 128 (hops 64, 32): a clip's frames are walked as `sub` interleaved sequences of hop 128
                           // (frames v, v + sub, v + 2 sub, ...), each of which slides its window in registers; 1 otherwise
    int64_t total_frames;  // n_frames * n_clips, flattened index g = clip * n_frames + j, j = position in that walk
    int n_waves;           // waves in the grid; wave w owns run_len (+ 1 for w < run_rem) consecutive g, in wave order
    int64_t run_len;       // total_frames / n_waves
    int run_rem;           // total_frames % n_waves
    double inv_n_frames;   // 1.0 / n_frames: a run's first (clip, position) by one multiply and a fix-up (see the kernel)
    float* out;            // [clip][frame][513] (OUT_PSD / OUT_MAG), [clip][frame] (OUT_BAND), [clip][frame][k_hi-k_lo+1] (OUT_DB*)
    int64_t out_clip_stride;
    const float2* win2;    // [512]  (w[2n], w[2n+1]) * sqrt(scale / 2) or sqrt(scale / 4): sg_plan::r8_win_dev
    const float2* tw;      // [18][64]
    float scale;
    int k_lo, k_hi;        // OUT_BAND, OUT_DB_BAND
    float inv_base;        // OUT_DB*: 1 / (global_max + 1e-20)
    float2* mm_parts;      // OUT_DB*: [n_waves] (min, max) of the dB values a wave wrote
    // OUT_MEL: the band-sparse mel bank of sg_mel_sparse_pack (host_shim.cpp)
    const int* mel_start;      // [64*IPL]     first bin of work item i (item = up to 8 consecutive bins of one band)
    const float* mel_w;        // [8][64*IPL]  its weights, zero padded
    const int* mel_first;      // [n_mels]     first item of band j
    const int* mel_count;      // [n_mels]     items of band j
    int n_mels, log_scale;
#ifdef SG_R8_STAMP
    unsigned long long* stamps;   // diagnostic build only (tools/limiter.py): [n_waves][16] clock stamps; no output depends on them
#endif
};

// What a frame leaves in HBM.
//   OUT_PSD / OUT_MAG  the 513-bin row (scipy mode 'psd' / 'magnitude')
//   OUT_BAND           sum_{k_lo..k_hi} of the PSD row (A11, PlotEngine.py:238-239): 4 B per frame
//   OUT_DB_FULL/_BAND  10*log10(clip(S/(gmax+1e-20), 0, 1) + 1e-12) of bins [k_lo, k_hi] (PlotEngine.py:126-129 with a
//                      caller-supplied global_max, :110) plus the wave's min / max of what it wrote, so that the min-max
//                      rescale of :130-131 needs no further pass over the spectrum
//   OUT_MEL1..4        the mel spectrum [n_mels] of the PSD row through a band-sparse bank (cfg3; IPL = 1..4 work items per lane):
//                      a triangular bank touches every bin with exactly two bands, so the contraction is ~1000 multiply-adds per
//                      frame, not the 41 040 of the dense product (or the ~20 000 a block-sparse MFMA form issues).  The wave
//                      leaves its PSD row in LDS, every lane gathers the <= 8 bins of its work items against weights it keeps in
//                      registers, the partial sums of a band (adjacent items) are added by the lane that owns the band.
//                      No workgroup barrier, no 16-frame tile: the kernel keeps the occupancy and the run structure of OUT_BAND.
enum { OUT_PSD = 0, OUT_MAG = 1, OUT_BAND = 2, OUT_DB_FULL = 3, OUT_DB_BAND = 4, OUT_MEL1 = 5, OUT_MEL2 = 6, OUT_MEL3 = 7, OUT_MEL4 = 8 };
constexpr int kMelRow = 544;     // floats of a wave's PSD row in LDS: 513 bins + the 7 a last item may read past them, rounded up
constexpr int kMelPart = 256;    // partial sums: one per work item (64 * IPL <= 256)

__device__ __forceinline__ float wave_min_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// OUT: see the enum above.
// H: hop / 128 when the hop is a multiple of 128 samples (1..8), else 0.  With H > 0 consecutive frames of a
// wave share registers: lane l keeps samples 2l + 128k (k = 0..7), the next frame needs k + H, so only H new
// float2 per lane are fetched per frame and every sample is loaded once per wave.  In both cases the loads of
// frame f+1 are issued before the FFT of frame f, which hides the HBM/L2 latency behind ~1000 VALU cycles.
//
// Work split: the grid is persistent (a few workgroups per CU); wave w owns a contiguous run of the flattened
// (clip, frame) index space, runs differ by at most one frame (the first total % n_waves waves hold the longer ones).
// Window and twiddles stay in VGPRs (SG_TW_LDS=0, ~118 VGPRs = 4 waves/SIMD); SG_TW_LDS=1 moves the twiddles to a
// 9 KiB workgroup-shared LDS table (83 VGPRs = 5 waves/SIMD).
constexpr int occupancy_for(int out) { return out >= OUT_MEL1 ? (kOccupancy < 3 ? kOccupancy : 3) : kOccupancy; }   // OUT_MEL: +8*IPL weight registers

template <typename TIn, bool ALIGNED, bool DETREND, int OUT, int H>
__global__ __launch_bounds__(64 * kWavesPerWg, occupancy_for(OUT)) void stft1024_r8x3_kernel(const R8Params p) {
    constexpr bool BAND = OUT == OUT_BAND;
    constexpr bool DB = OUT == OUT_DB_FULL || OUT == OUT_DB_BAND;
    constexpr bool MEL = OUT >= OUT_MEL1;
    constexpr int IPL = MEL ? OUT - OUT_MEL1 + 1 : 1;        // work items per lane
    constexpr int MODE = OUT == OUT_MAG ? 1 : 0;
    __shared__ __attribute__((aligned(16))) float mel_lds[MEL ? kWavesPerWg * (kMelRow + kMelPart) : 1];
    __shared__ __attribute__((aligned(16))) float2 lds[kWavesPerWg * kSlab + (SG_TW_LDS ? 18 * 64 : 0)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2* const buf = lds + wave * kSlab;
#if SG_TW_LDS
    float2* const twl = lds + kWavesPerWg * kSlab;          // [18][64] twiddles, lane-linear rows
#endif

#if SG_TW_LDS
    for (int i = threadIdx.x; i < 18 * 64; i += 64 * kWavesPerWg) twl[i] = p.tw[i];
    __syncthreads();                                        // the only barrier: before any wave may exit
#endif

    const int lw = xcd_remap(blockIdx.x, gridDim.x) * kWavesPerWg + wave;     // logical wave index
    if (lw >= p.n_waves) return;
#ifdef SG_R8_STAMP
    // MI355X_MICROARCH.md, DVFS give-back item 6: shader-clock (s_memtime) and 100 MHz (s_memrealtime) stamps around the frame
    // loop of a DIAGNOSTIC build (tools/build_variant.sh stamp ... -DSG_R8_STAMP=1), written to a buffer of their own
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_c1 = 0, st_r1 = 0;
#endif

    float2 w[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) w[a] = p.win2[lane + 64 * a];
#if SG_TW_LDS
    const float2* const t1 = twl + lane;                    // + 64*(r-1),  r = 1..7
    const float2* const t2 = twl + 7 * 64 + lane;           // + 64*(s-1),  s = 1..7
    const float2* const t3 = twl + 14 * 64 + lane;          // + 64*m,      m = 0..3
#define SG_TW(tab, i) lds_get((tab) + 64 * (i))
#else
    float2 t1[7], t2[7], t3[4];
#pragma unroll
    for (int r = 0; r < 7; ++r) { t1[r] = p.tw[r * 64 + lane]; t2[r] = p.tw[(7 + r) * 64 + lane]; }
#pragma unroll
    for (int m = 0; m < 4; ++m) t3[m] = p.tw[(14 + m) * 64 + lane];
#define SG_TW(tab, i) (tab)[i]
#endif

    const int j0 = lane & 7, hi = lane >> 3;
    float2* const x1w = buf + hi * kS1 + j0;        // + 8*r
    float2* const x1r = buf + lane;                 // + b*kS1
    float2* const x2w = buf + j0 * kS2 + hi;        // + 8*s
    float2* const x2r = buf + lane;                 // + j0*kS2
    float2* const x3w = buf + lane;                 // + 64*t
    const float2* const x3b = buf + (kM - lane);    // - 64*m

    // |X|^2 * q: q = scale/2 for the interior bins of a one-sided PSD (doubled), scale/4 for bins 0 and 512 and for every
    // bin of a magnitude spectrum.  sqrt(q) rides on the window table (build_r8x3_tables: no multiply here, so nothing in the
    // prologue waits for a table before the first frame's samples are requested); what is left per frame is the factor 1/2 on
    // lane 0's bins 0 and 512 of a PSD.
    const float r0 = (MODE == 0 && lane == 0) ? 0.5f : 1.0f;
    float vmin = INFINITY, vmax = -INFINITY;         // OUT_DB*: this lane's extrema of the dB values written
    const int row_len = DB ? p.k_hi - p.k_lo + 1 : (MEL ? p.n_mels : kBins);
    // OUT_MEL: this wave's PSD row and partial sums in LDS; the lane's work items (first bin, 8 weights) and the items of the
    // bands it owns (band = lane + 64 * pass) stay in registers for the whole run
    float* const mrow = mel_lds + wave * (kMelRow + kMelPart);
    float* const mpart = mrow + kMelRow;
    float mw[IPL][8];
    const float* mgather[IPL];
    int bfirst[2] = {0, 0}, bcount[2] = {0, 0};
    if (MEL) {
#pragma unroll
        for (int i = 0; i < IPL; ++i) {
            mgather[i] = mrow + p.mel_start[lane + 64 * i];
#pragma unroll
            for (int c = 0; c < 8; ++c) mw[i][c] = p.mel_w[(c * IPL + i) * 64 + lane];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = lane + 64 * q;
            if (j < p.n_mels) { bfirst[q] = p.mel_first[j]; bcount[q] = p.mel_count[j]; }
        }
        if (lane < kMelRow - kBins) mrow[kBins + lane] = 0.f;        // the slots behind bin 512 meet zero weights: keep them finite
    }

    // This wave's run [g, g_end) of the flattened space, and the (clip, position in the clip's walk) it starts at -- without an integer
    // division: the three 64-bit divisions this used to take are ~700 scalar instructions, and the three waves of a SIMD issue their
    // scalar instructions one every four cycles between them: 3.5 us of a 3.6-3.9 us prologue (in-kernel stamps, profiles/r03_limiter.txt).
    int64_t g = p.run_len * lw + min(lw, p.run_rem);
    const int64_t g_end = g + p.run_len + (lw < p.run_rem ? 1 : 0);
    int clip = static_cast<int>(static_cast<double>(g) * p.inv_n_frames);          // within one of the quotient: fixed up below
    int pos;
    {
        int64_t r = g - static_cast<int64_t>(clip) * p.n_frames;
        if (r < 0) { --clip; r += p.n_frames; } else if (r >= p.n_frames) { ++clip; r -= p.n_frames; }
        pos = static_cast<int>(r);
    }
    while (g < g_end) {                              // one iteration per stretch: frames of one clip (and one sequence) that follow each other
        const int f0 = pos;
        const int f1 = static_cast<int>(min(static_cast<int64_t>(p.n_frames), f0 + (g_end - g)));

        const TIn* const xclip = static_cast<const TIn*>(p.x) + static_cast<int64_t>(clip) * p.clip_stride + 2 * lane;
        // positions [f0, f1) of the clip's walk -> one stretch of one sequence: frames fs, fs + sub, ..., `count` of them
        // (sub == 1: the frames f0 ... f1 - 1 themselves).  Sequence v holds ceil((n_frames - v) / sub) frames.
        int fs = f0, count = f1 - f0;
        if (p.sub > 1) {
            int v = 0, first = 0, len = (p.n_frames + p.sub - 1) / p.sub;
            while (f0 >= first + len) { first += len; ++v; len = (p.n_frames - v + p.sub - 1) / p.sub; }
            fs = (f0 - first) * p.sub + v;
            count = min(f1 - f0, first + len - f0);         // the rest of [f0, f1) belongs to the next sequence: next trip
        }
        g += count;
        pos += count;
        const int clip_now = clip;
        if (pos == p.n_frames) { pos = 0; ++clip; }        // the next stretch opens the next clip
        const int fstep = p.sub;
        // (OUT_DB_BAND: orow[k] is bin k's slot, i.e. the row start minus k_lo)
        float* orow = BAND ? p.out + static_cast<int64_t>(clip_now) * p.out_clip_stride + fs
                           : p.out + static_cast<int64_t>(clip_now) * p.out_clip_stride + static_cast<int64_t>(fs) * row_len -
                                 (OUT == OUT_DB_BAND ? p.k_lo : 0);
        const int row_step = (BAND ? 1 : row_len) * fstep;
        const int src_step = p.hop * fstep;
        const TIn* src = xclip + static_cast<int64_t>(fs) * p.hop;

        float2 raw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) raw[k] = load_pair<TIn, ALIGNED>(src + 128 * k);
#ifdef SG_R8_STAMP
        if (st_c1 == 0) {      // prologue over: tables in registers, first frame's samples here
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_c1 = __builtin_amdgcn_s_memtime(); st_r1 = __builtin_amdgcn_s_memrealtime();
        }
#endif

        // A3 + A4 on the frame held in a[] (reads only; the sample registers stay intact)
        auto prep = [&](float2 (&a)[8]) {
            if (DETREND) {   // A3: subtract the frame mean (scipy:2191, detrend type 'constant')
                float s = a[0].x + a[0].y;
#pragma unroll
                for (int k = 1; k < 8; ++k) s += a[k].x + a[k].y;
                const float mean = wave_sum(s) * (1.0f / kN);
#pragma unroll
                for (int k = 0; k < 8; ++k) { a[k].x -= mean; a[k].y -= mean; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { a[k].x *= w[k].x; a[k].y *= w[k].y; }   // A4 window
        };
        // the spectrum of the windowed frame in a[] -> row orow (or one band sum), then orow advances
        auto process = [&](float2 (&a)[8]) {
            if (SG_R8_PRIO) __builtin_amdgcn_s_setprio(0);
            // ---- pass 1: DFT over a (stride-64 elements), twiddle w512^(lane*r)
            radix8(a);
#pragma unroll
            for (int r = 1; r < 8; ++r) a[r] = cmul(a[r], SG_TW(t1, r - 1));
#pragma unroll
            for (int r = 0; r < 8; ++r) lds_put(x1w + 8 * r, a[r]);
            wave_lds_fence();
#pragma unroll
            for (int b = 0; b < 8; ++b) a[b] = lds_get(x1r + b * kS1);
            wave_lds_fence();

            if (SG_R8_PRIO) __builtin_amdgcn_s_setprio(1);
            // ---- pass 2: lane = j0 + 8r, DFT over b, twiddle w64^(j0*s)
            radix8(a);
#pragma unroll
            for (int s = 1; s < 8; ++s) a[s] = cmul(a[s], SG_TW(t2, s - 1));
#pragma unroll
            for (int s = 0; s < 8; ++s) lds_put(x2w + 8 * s, a[s]);
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = lds_get(x2r + j * kS2);
            wave_lds_fence();

            if (SG_R8_PRIO) __builtin_amdgcn_s_setprio(2);
            // ---- pass 3: lane = r + 8s, DFT over j0 -> Z[lane + 64t]
            radix8(a);
            // Split pass pairs k = lane + 64m (m = 0..3, still in registers a[0..3]) with 512-k = Z[(64-lane) + 64(7-m)],
            // which sits in registers a[4..7] of lane 64-lane: only the upper half crosses lanes (4 LDS writes, not 8).
#pragma unroll
            for (int t = 4; t < 8; ++t) lds_put(x3w + 64 * t, a[t]);
            lds_put(buf + kM + lane, a[0]);                 // lane 0: Z[512] := Z[0] closes the k <-> 512-k pairing (the slab's
                                                            // last 63 slots take the other lanes' copies: no exec-mask branch)
            wave_lds_fence();

            // ---- split pass + |X|^2 epilogue (A5 tail + A6) ----------------------
            if (SG_R8_PRIO) __builtin_amdgcn_s_setprio(3);
            float band = 0.f;
            auto emit = [&](int k, float v) {          // bin k of this frame
                if (BAND) {
                    band += (k >= p.k_lo && k <= p.k_hi) ? v : 0.f;
                } else if (DB) {
                    // PlotEngine.py:126-129; S >= 0, so only the upper clip acts; fminf also maps a NaN bin to 1 -> ~0 dB,
                    // which is what np.nan_to_num leaves of it
                    const float d = 3.01029995663981195f * __log2f(fminf(v * p.inv_base, 1.0f) + 1e-12f);
                    if (OUT == OUT_DB_FULL || (k >= p.k_lo && k <= p.k_hi)) {
                        orow[k] = d;
                        vmin = fminf(vmin, d);
                        vmax = fmaxf(vmax, d);
                    }
                } else if (MEL) {
                    mrow[k] = v;
                } else {
                    orow[k] = v;
                }
            };
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float2 A = a[m];
                const float2 B = lds_get(x3b - 64 * m);             // Z[512-k]; conj applied below
                const float2 cs = SG_TW(t3, m);                     // cos, sin of 2*pi*k/1024
                const float2 S = make_float2(A.x + B.x, A.y - B.y); // A + conj(B)
                const float2 D = make_float2(A.x - B.x, A.y + B.y); // A - conj(B)
                const float2 T = make_float2(fmaf(cs.y, D.x, -cs.x * D.y), fmaf(cs.x, D.x, cs.y * D.y));   // i*W^k*D
                const float2 Xk = csub(S, T), Xm = cadd(S, T);      // 2*X[k], 2*conj(X[512-k])
                // (|S|^2 + |D|^2 -+ 2Re(S conj T) would save two ops but cancels catastrophically for weak bins)
                float pk = fmaf(Xk.x, Xk.x, Xk.y * Xk.y);
                float pm = fmaf(Xm.x, Xm.x, Xm.y * Xm.y);
                if (MODE == 0 && m == 0) { pk *= r0; pm *= r0; }
                if (MODE == 1) { pk = sqrtf(pk); pm = sqrtf(pm); }
                const int k = lane + 64 * m;
                emit(k, pk);
                emit(kM - k, pm);
            }
            {   // k = 256 pairs with itself: X[256] = conj(Z[256]); Z[256] is lane 0's a[4]
                const float zx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[4].x), 0));
                const float zy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[4].y), 0));
                float pq = fmaf(zx, zx, zy * zy) * 4.0f;
                if (MODE == 1) pq = sqrtf(pq);
                if (BAND) {
                    band += (lane == 0 && 256 >= p.k_lo && 256 <= p.k_hi) ? pq : 0.f;
                } else {
                    // every lane holds the same pq: a wave-uniform store keeps the loop branch-free, so the
                    // compiler's s_waitcnt for the prefetched loads stays exact (vmcnt = stores issued since)
                    emit(256, pq);
                }
            }
            if (BAND) {
                const float tot = wave_sum(band);
                if (lane == 0) *orow = tot;
                orow += row_step;
            } else if (MEL) {
                wave_lds_fence();                                  // the row is complete
#pragma unroll
                for (int i = 0; i < IPL; ++i) {                    // <= 8 bins of one band per work item
                    float acc = mw[i][0] * mgather[i][0];
#pragma unroll
                    for (int c = 1; c < 8; ++c) acc = fmaf(mw[i][c], mgather[i][c], acc);
                    mpart[lane + 64 * i] = acc;
                }
                wave_lds_fence();
#pragma unroll
                for (int q = 0; q < 2; ++q) {                      // the lane that owns band lane + 64 q adds its items
                    if (q == 1 && p.n_mels <= 64) break;
                    float v = 0.f;
                    for (int t = 0; t < bcount[q]; ++t) v += mpart[bfirst[q] + t];
                    if (p.log_scale) v = 3.01029995663981195f * __log2f(fmaxf(v, 1e-10f));      // 10 log10 through v_log_f32
                    if (lane + 64 * q < p.n_mels) orow[lane + 64 * q] = v;
                }
                orow += row_step;
            } else {
                orow += row_step;
            }
            wave_lds_fence();    // next frame's exchange-1 writes stay behind these reads
        };

        // (fetching two frames ahead, so that two frames of stores may be in flight behind a request, measured slower:
        //  94.3 vs 90.0 us)
        // (rotating the sample window through its registers instead of shifting it -- the frame loop unrolled over the
        //  8 / gcd(8, H) rotations, no v_mov -- measured twice: round 1 83.6-84.7 vs 84.3-85.1 us at 142 VGPRs, round 2
        //  97.8 us at 144 VGPRs / 3 waves per SIMD and 119.7 us with the spills of 128 VGPRs against 89.2 us rolled;
        //  profiles/r02_ab_rotation.txt)
        for (int i = 0; i < count; ++i) {
            float2 a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = raw[k];
            prep(a);
            src += src_step;
            if (i + 1 < count) {   // prefetch the stretch's next frame (wave-uniform branch)
                if (H > 0) {
#pragma unroll
                    for (int k = 0; k + H < 8; ++k) raw[k] = raw[k + H];
#pragma unroll
                    for (int k = (H > 0 ? 8 - H : 0); k < 8; ++k) raw[k] = load_pair<TIn, ALIGNED>(src + 128 * k);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) raw[k] = load_pair<TIn, ALIGNED>(src + 128 * k);
                }
            }
            process(a);
        }
    }
    if (DB) {
        vmin = wave_min_f(vmin);
        vmax = wave_max_f(vmax);
        if (lane == 0) p.mm_parts[lw] = make_float2(vmin, vmax);
    }
#ifdef SG_R8_STAMP
    if (p.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the last rows have left the wave
        const unsigned long long st_c2 = __builtin_amdgcn_s_memtime(), st_r2 = __builtin_amdgcn_s_memrealtime();
        unsigned int hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        if (lane == 0) {
            unsigned long long* q = p.stamps + static_cast<size_t>(lw) * 16;
            q[0] = st_c0; q[1] = st_r0; q[2] = st_c1; q[3] = st_r1; q[4] = st_c2; q[5] = st_r2;
            q[6] = static_cast<unsigned long long>(p.run_len + (lw < p.run_rem ? 1 : 0));
            q[7] = (static_cast<unsigned long long>(xcc_id) << 32) | hw_id;
        }
    }
#endif
}

template <typename TIn, bool ALIGNED, bool DETREND, int OUT, int H>
int launch_h(const R8Params& prm, int n_wg, hipStream_t stream) {
    hipLaunchKernelGGL((stft1024_r8x3_kernel<TIn, ALIGNED, DETREND, OUT, H>), dim3(n_wg), dim3(64 * kWavesPerWg), 0,
                       stream, prm);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "stft1024_r8x3 launch");
    return SG_OK;
}

// register-rotating variants exist for aligned f32 input at the hops that matter (128, 256, 512, 896 = the
// reference default n - n/8); everything else takes the H = 0 prefetch path.
template <typename TIn, bool ALIGNED, bool DETREND, int OUT>
int launch_one(const R8Params& prm, int n_wg, hipStream_t stream) {
    if constexpr (std::is_same<TIn, float>::value && ALIGNED) {
        switch (prm.hop * prm.sub) {
            case 128: return launch_h<TIn, ALIGNED, DETREND, OUT, 1>(prm, n_wg, stream);
            case 256: return launch_h<TIn, ALIGNED, DETREND, OUT, 2>(prm, n_wg, stream);
            case 512: return launch_h<TIn, ALIGNED, DETREND, OUT, 4>(prm, n_wg, stream);
            case 896: return launch_h<TIn, ALIGNED, DETREND, OUT, 7>(prm, n_wg, stream);
            default: break;
        }
    }
    return launch_h<TIn, ALIGNED, DETREND, OUT, 0>(prm, n_wg, stream);
}

template <typename TIn, bool ALIGNED, bool DETREND>
int launch_out(const R8Params& prm, int n_wg, hipStream_t s, int out) {
    switch (out) {
        case OUT_PSD: return launch_one<TIn, ALIGNED, DETREND, OUT_PSD>(prm, n_wg, s);
        case OUT_MAG: return launch_one<TIn, ALIGNED, DETREND, OUT_MAG>(prm, n_wg, s);
        case OUT_BAND: return launch_one<TIn, ALIGNED, DETREND, OUT_BAND>(prm, n_wg, s);
        default: break;
    }
    if constexpr (std::is_same<TIn, float>::value) {      // the dB image is wired for float input
        if (out == OUT_DB_FULL) return launch_one<TIn, ALIGNED, DETREND, OUT_DB_FULL>(prm, n_wg, s);
        if (out == OUT_DB_BAND) return launch_one<TIn, ALIGNED, DETREND, OUT_DB_BAND>(prm, n_wg, s);
    }
    if constexpr (std::is_same<TIn, float>::value && ALIGNED) {      // the mel form: aligned float input, hop 256 slides, others reload
        const bool slide = prm.hop == 256;
        switch (out) {
            case OUT_MEL1: return slide ? launch_h<TIn, ALIGNED, DETREND, OUT_MEL1, 2>(prm, n_wg, s) : launch_h<TIn, ALIGNED, DETREND, OUT_MEL1, 0>(prm, n_wg, s);
            case OUT_MEL2: return slide ? launch_h<TIn, ALIGNED, DETREND, OUT_MEL2, 2>(prm, n_wg, s) : launch_h<TIn, ALIGNED, DETREND, OUT_MEL2, 0>(prm, n_wg, s);
            case OUT_MEL3: return slide ? launch_h<TIn, ALIGNED, DETREND, OUT_MEL3, 2>(prm, n_wg, s) : launch_h<TIn, ALIGNED, DETREND, OUT_MEL3, 0>(prm, n_wg, s);
            case OUT_MEL4: return slide ? launch_h<TIn, ALIGNED, DETREND, OUT_MEL4, 2>(prm, n_wg, s) : launch_h<TIn, ALIGNED, DETREND, OUT_MEL4, 0>(prm, n_wg, s);
            default: break;
        }
    }
    set_error("r8x3: output form %d is not built for this input type", out);
    return SG_ERR_UNSUPPORTED;
}

template <typename TIn>
int launch_in(const R8Params& prm, int n_wg, hipStream_t s, bool aligned, bool detrend, int out) {
    if (aligned) {
        return detrend ? launch_out<TIn, true, true>(prm, n_wg, s, out) : launch_out<TIn, true, false>(prm, n_wg, s, out);
    }
    return detrend ? launch_out<TIn, false, true>(prm, n_wg, s, out) : launch_out<TIn, false, false>(prm, n_wg, s, out);
}

}  // namespace

// persistent grid: kOccupancy waves per SIMD on every CU, but never runs shorter than kMinRun frames
int r8x3_grid_waves(const sg_plan& p, int64_t total_frames, bool mel) {
    int occ = mel ? occupancy_for(OUT_MEL1) : kOccupancy;
    // short batches (the reference's own call: hop 896, 34 240 frames per 64 clips = 11 frames per wave at three waves per SIMD): two waves
    // per SIMD with runs half as long again are 2-5 % faster (32.3 vs 33.1-34.2 us, same box, two rounds; one wave: 42 us)
    if (occ > 2 && total_frames > static_cast<int64_t>(p.n_cu) * 4 * occ && total_frames < static_cast<int64_t>(p.n_cu) * 4 * occ * 14) occ = 2;
    if (const char* e = SG_TUNE_ENV("SPECTRO_R8_OCC")) { const int v = atoi(e); if (v >= 1 && v <= kOccupancy) occ = v; }   // tuning aid
    int64_t n_waves = static_cast<int64_t>(p.n_cu) * 4 * occ;
    if (const char* e = SG_TUNE_ENV("SPECTRO_R8_WAVES")) { const long v = atol(e); if (v >= 64 && v <= 65536) n_waves = v; }          // tuning aid
    // GUI-sized calls (fewer frames than waves the chip holds): one frame per wave, latency before efficiency
    const int64_t by_work = total_frames <= n_waves ? total_frames : (total_frames + kMinRun - 1) / kMinRun;
    if (n_waves > by_work) n_waves = by_work;
    return static_cast<int>(n_waves);
}

int launch_r8x3(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    if (a.n_frames > INT32_MAX) { set_error("r8x3: more than 2^31 frames per clip"); return SG_ERR_ARG; }
    R8Params prm{};
    prm.x = a.x;
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.sub = 1;
    prm.total_frames = a.n_frames * a.n_clips;
    prm.n_waves = r8x3_grid_waves(p, prm.total_frames, a.mel_ipl > 0);
    prm.run_len = prm.total_frames / prm.n_waves;
    prm.run_rem = static_cast<int>(prm.total_frames % prm.n_waves);
    prm.inv_n_frames = 1.0 / static_cast<double>(a.n_frames);
    const int n_wg = (prm.n_waves + kWavesPerWg - 1) / kWavesPerWg;
    prm.out = static_cast<float*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.win2 = static_cast<const float2*>(p.r8_win_dev);
    prm.tw = static_cast<const float2*>(p.r8_tw_dev);
    prm.scale = static_cast<float>(p.scale);
    prm.k_lo = a.k_lo;
    prm.k_hi = a.k_hi;
#ifdef SG_R8_STAMP
    if (const char* e = SG_TUNE_ENV("SPECTRO_R8_STAMP_PTR")) prm.stamps = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    int out = a.band_mode ? OUT_BAND : (p.mode == SG_MODE_PSD ? OUT_PSD : OUT_MAG);
    if (a.db_mode) {
        if (p.mode != SG_MODE_PSD || !a.mm_parts) { set_error("r8x3: dB image needs a psd plan and a partials buffer"); return SG_ERR_ARG; }
        out = (a.k_lo == 0 && a.k_hi == kBins - 1) ? OUT_DB_FULL : OUT_DB_BAND;
        prm.inv_base = a.inv_base;
        prm.mm_parts = static_cast<float2*>(a.mm_parts);
    }
    if (a.mel_ipl > 0) {
        if (p.mode != SG_MODE_PSD || a.mel_ipl > 4 || !a.mel_start || !a.mel_w || !a.mel_first || !a.mel_count || a.in_i16) {
            set_error("r8x3: the mel form needs a psd plan, float input and a band-sparse bank of at most 256 work items");
            return SG_ERR_ARG;
        }
        out = OUT_MEL1 + a.mel_ipl - 1;
        prm.mel_start = a.mel_start; prm.mel_w = a.mel_w; prm.mel_first = a.mel_first; prm.mel_count = a.mel_count;
        prm.n_mels = a.n_mels; prm.log_scale = a.log_scale;
    }
    const bool detrend = p.detrend == SG_DETREND_CONSTANT;
    if (a.in_i16) {
        const bool aligned = (p.hop % 2 == 0) && (a.clip_stride % 2 == 0) && (reinterpret_cast<uintptr_t>(a.x) % 4 == 0);
        return launch_in<int16_t>(prm, n_wg, a.stream, aligned, detrend, out);
    }
    const bool aligned = (p.hop % 2 == 0) && (a.clip_stride % 2 == 0) && (reinterpret_cast<uintptr_t>(a.x) % 8 == 0);
    // hops 64, 32, 16: two / four / eight interleaved hop-128 sequences, so the window slides in registers (one 8-byte load per
    // lane and frame instead of eight); rows of one sequence are 2 / 4 / 8 rows apart in the output
    if (aligned && a.mel_ipl == 0 && (p.hop == 64 || p.hop == 32 || p.hop == 16) && !SG_TUNE_ENV("SPECTRO_R8_NO_SUB")) prm.sub = 128 / p.hop;
    // tuning aid (tools/ab_sub.py): walk ANY hop as `sub` interleaved sequences, e.g. hop 256 as two hop-512 sequences whose rows alternate in the output
    if (const char* e = SG_TUNE_ENV("SPECTRO_R8_SUB")) { const int v = atoi(e); if (aligned && a.mel_ipl == 0 && v >= 1 && v <= 8) prm.sub = v; }
    if (a.mel_ipl > 0 && !aligned) { set_error("r8x3: the mel form needs an even hop / clip stride and 8-byte aligned input"); return SG_ERR_UNSUPPORTED; }
    return launch_in<float>(prm, n_wg, a.stream, aligned, detrend, out);
}

// Per-lane twiddle table [18][64] (float2), computed in double:
//   rows 0..6   t1[r-1][j]  = exp(-2*pi*i*j*r/512)            r = 1..7, j  = lane
//   rows 7..13  t2[s-1][j]  = exp(-2*pi*i*(j&7)*s/64)         s = 1..7
//   rows 14..17 t3[m][j]    = (cos, sin)(2*pi*(j+64m)/1024)   m = 0..3
int build_r8x3_tables(sg_plan& p, const std::vector<double>& /*window*/) {
    std::vector<float2> tw(18 * 64);
    const double two_pi = 6.283185307179586476925286766559;
    for (int j = 0; j < 64; ++j) {
        for (int r = 1; r < 8; ++r) {
            const double ang = -two_pi * static_cast<double>((j * r) % 512) / 512.0;
            tw[(r - 1) * 64 + j] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
            const double ang2 = -two_pi * static_cast<double>(((j & 7) * r) % 64) / 64.0;
            tw[(7 + r - 1) * 64 + j] = make_float2(static_cast<float>(std::cos(ang2)), static_cast<float>(std::sin(ang2)));
        }
        for (int m = 0; m < 4; ++m) {
            const double ang = two_pi * static_cast<double>(j + 64 * m) / 1024.0;
            tw[(14 + m) * 64 + j] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
        }
    }
    SG_HIP(hipMalloc(&p.r8_tw_dev, tw.size() * sizeof(float2)));
    SG_HIP(hipMemcpy(p.r8_tw_dev, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    // the window as the kernel wants it: times sqrt(q), q = scale / 2 (psd: interior bins doubled) or scale / 4 (magnitude), in float
    // arithmetic (what the kernel itself did per wave until round 3: same roundings, same bits)
    std::vector<float> w(kN);
    SG_HIP(hipMemcpy(w.data(), p.win_dev, kN * sizeof(float), hipMemcpyDeviceToHost));
    const float sc = static_cast<float>(p.scale);
    const float sq = sqrtf(p.mode == SG_MODE_MAGNITUDE ? sc * 0.25f : sc * 0.5f);
    for (float& v : w) v *= sq;
    if (p.r8_win_dev) { (void)hipFree(p.r8_win_dev); p.r8_win_dev = nullptr; }
    SG_HIP(hipMalloc(&p.r8_win_dev, kN * sizeof(float)));
    SG_HIP(hipMemcpy(p.r8_win_dev, w.data(), kN * sizeof(float), hipMemcpyHostToDevice));
    return SG_OK;
}

}  // namespace sg
