// stft_rsmall.hip -- register FFT kernel for the small transforms of the parameter sweep (BASELINE cfg4):
// nperseg = nfft = 128*R with R = 2 (256) or R = 4 (512), f32, detrend none|constant, psd|magnitude -- and R = 1 (128, eight frames per
// wave, no pass 1: round 4; the spin box starts at 32 and steps by 32, /root/reference/GUI.py:87-89).
//
// Same machine mapping as stft_r8x3.hip (one wavefront, 8 complex values per lane, three register passes, two
// padded LDS transposes, split pass with only the upper half crossing lanes, no s_barrier in the frame loop), but a
// wave carries G = 8/R frames at once so that all 64 lanes stay busy:
//   pass 1: lane j holds z_g[j + 64a] (a < R) of every frame g -> G independent R-point DFTs, twiddle w_M^(j*r)
//   pass 2/3: the radix-8 passes of the 1024 kernel with the register index v = g*R + r
//   after pass 3 lane l = lu + L*g (L = 8R lanes per frame) holds Z_g[lu + L*t], t = 0..7.
// Index maps and LDS bank behaviour are replayed in tools/sim_rsmall.py.
// (A register sliding window as in stft_r8x3.hip / stft_rbig.hip -- a group's G frames span R + G - 1 blocks of 128 samples, the
//  next group G blocks further, G loads per group instead of 8 -- was built and measured in round 2 at hops 128 / 64: 0.121 /
//  0.065 ms against 0.120 / 0.066 ms (nfft 256) and 0.230 / 0.118 against 0.217 / 0.118 ms (nfft 512) per 64-clip batch.  These
//  kernels run below the power cap at 2.4 GHz: the re-read samples are not what limits them.  Not kept.)
// Algorithmic HBM bytes per frame: hop*4 + (64R+1)*4.
#include "spectro_internal.h"
#include "fft_wave.h"

#include <cmath>
#include <cstdlib>

#ifndef SG_RSMALL_PRIO
#define SG_RSMALL_PRIO 1        // wave priority rises along a group of frames (pass 1 -> stores), as in stft_r8x3; 0 = off
#endif

namespace sg {
namespace {

using namespace wavefft;

#ifndef SG_RSMALL_X4
#define SG_RSMALL_X4 1          // the staged rows of a group leave as 16-byte stores (ds_read_b128 from the slab, global_store_dwordx4): a quarter of the store
                                // instructions: -1 ... -4 % (profiles/r04_rsmall_whatif.txt).  Round 4's what-ifs: this kernel's time is its global loads and stores (removing
                                // either saves 30-35 %), not its exchanges or arithmetic (removing any of them changes nothing); a second group of prefetch and a round-robin deal
                                // of the groups do not help (+4 %, +-2 %): it runs at 3.5-3.7 TB/s of a 1 : 2 read : write mix from 4 096 streams.
                                // (the rows start on 4-byte boundaries: the 16-byte stores are unaligned, which the hardware takes.)  0 = dword stores
#endif
#ifndef SG_RSMALL_WPW
#define SG_RSMALL_WPW 4
#endif
constexpr int kS1 = 72, kS2 = 66, kSlab = 8 * kS1, kWaves = SG_RSMALL_WPW;

struct SmallParams {
    const float* x;
    int64_t clip_stride;
    int n_frames, hop;
    int groups_per_clip;      // ceil(n_frames / G)
    int64_t total_groups;
    int n_waves;
    float* out;
    int64_t out_clip_stride;
    const float2* win2;       // [M] pairs (w[2n], w[2n+1])
    const float2* tw;         // [(R-1) + 7 + 4][64]
    float scale;
    int k_lo, k_hi;           // MODE 2: bins of the band
};

template <int L> __device__ __forceinline__ float group_sum(float v) {      // over the L = 8R lanes that share a frame
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int R> __device__ __forceinline__ void radix_small(float2* a);
template <> __device__ __forceinline__ void radix_small<1>(float2*) {}
template <> __device__ __forceinline__ void radix_small<2>(float2* a) {
    const float2 s = cadd(a[0], a[1]), d = csub(a[0], a[1]);
    a[0] = s; a[1] = d;
}
template <> __device__ __forceinline__ void radix_small<4>(float2* a) {
    const float2 s02 = cadd(a[0], a[2]), d02 = csub(a[0], a[2]);
    const float2 s13 = cadd(a[1], a[3]), d13 = mul_mi(csub(a[1], a[3]));
    a[0] = cadd(s02, s13); a[2] = csub(s02, s13);
    a[1] = cadd(d02, d13); a[3] = csub(d02, d13);
}

// MODE 0 psd, 1 magnitude, 2 band power: out[clip][frame] = sum of PSD bins [k_lo, k_hi] (A11, the spectrum is never written)
template <int R, bool DETREND, int MODE>
__global__ __launch_bounds__(64 * kWaves) void stft_rsmall_kernel(const SmallParams p) {
    constexpr int G = 8 / R, M = 64 * R, L = 8 * R, NB = M + 1, RS = M + 8;
    static_assert(G * RS <= kSlab, "split regions must fit the slab");
    __shared__ __attribute__((aligned(16))) float2 lds[kWaves * kSlab];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2* const buf = lds + wave * kSlab;

    const int lw = xcd_remap(blockIdx.x, gridDim.x) * kWaves + wave;
    if (lw >= p.n_waves) return;
    int64_t q = p.total_groups * lw / p.n_waves;
    const int64_t q_end = p.total_groups * (lw + 1) / p.n_waves;

    // per-lane constants
    float2 w[R], t1[R > 1 ? R - 1 : 1], t2[7], t3[4];
#pragma unroll
    for (int a = 0; a < R; ++a) w[a] = p.win2[lane + 64 * a];
#pragma unroll
    for (int r = 0; r < R - 1; ++r) t1[r] = p.tw[r * 64 + lane];
#pragma unroll
    for (int s = 0; s < 7; ++s) t2[s] = p.tw[(R - 1 + s) * 64 + lane];
#pragma unroll
    for (int t = 0; t < 4; ++t) t3[t] = p.tw[(R - 1 + 7 + t) * 64 + lane];

    const int j0 = lane & 7, v_ = lane >> 3;
    float2* const x1w = buf + v_ * kS1 + j0;                          // + 8*v      (v_ here is b = lane>>3)
    float2* const x1r = buf + lane;                                   // + b*kS1
    float2* const x2w = buf + j0 * kS2 + (v_ % R) + L * (v_ / R);     // + R*s      (v_ here is v = g*R + r)
    float2* const x2r = buf + lane;                                   // + j0*kS2
    const int g3 = lane / L, lu = lane - g3 * L;
    float2* const x3w = buf + g3 * RS + lu;                           // + L*t
    const float2* const x3b = buf + g3 * RS + (M - lu);               // - L*t

    // sqrt of the PSD scale rides on the window registers (stft_r8x3.hip); per group only the 1/2 on bins 0 and M is left
    const float q_in = MODE != 1 ? p.scale * 0.5f : p.scale * 0.25f;
    {
        const float sq = sqrtf(q_in);
#pragma unroll
        for (int a = 0; a < R; ++a) { w[a].x *= sq; w[a].y *= sq; }
    }
    const float r0 = (MODE != 1 && lu == 0) ? 0.5f : 1.0f;

    // loads of group q+1 are issued before the FFT of group q (one group of register prefetch)
    auto load_group = [&](int clip, int gi, float2 (&dst)[8]) {
        const int fg = gi * G;
        const float* const xclip = p.x + static_cast<int64_t>(clip) * p.clip_stride + 2 * lane;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = min(fg + g, p.n_frames - 1);                 // partial last group: recompute the last frame
            const float* const src = xclip + static_cast<int64_t>(f) * p.hop;
#pragma unroll
            for (int k = 0; k < R; ++k) dst[g * R + k] = *reinterpret_cast<const float2*>(src + 128 * k);
        }
    };
    // (clip, group) of the run's first group by one division; after that they advance incrementally (a 64-bit
    // division is ~200 scalar instructions, and the scalar unit is shared by the CU)
    int clip = static_cast<int>(q / p.groups_per_clip);
    int gi = static_cast<int>(q - static_cast<int64_t>(clip) * p.groups_per_clip);
    float2 nxt[8];
    if (q < q_end) load_group(clip, gi, nxt);

    for (; q < q_end; ++q) {
        const int fg = gi * G;
        const int clip_n = gi + 1 == p.groups_per_clip ? clip + 1 : clip, gi_n = gi + 1 == p.groups_per_clip ? 0 : gi + 1;
        const bool more = q + 1 < q_end;                               // the run's last group fetches itself again

        float2 a[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) a[v] = nxt[v];
        load_group(more ? clip_n : clip, more ? gi_n : gi, nxt);
        if (SG_RSMALL_PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (DETREND) {
                float s = a[g * R].x + a[g * R].y;
#pragma unroll
                for (int k = 1; k < R; ++k) s += a[g * R + k].x + a[g * R + k].y;
                const float mean = wave_sum(s) * (1.0f / (2 * M));
#pragma unroll
                for (int k = 0; k < R; ++k) { a[g * R + k].x -= mean; a[g * R + k].y -= mean; }
            }
#pragma unroll
            for (int k = 0; k < R; ++k) { a[g * R + k].x *= w[k].x; a[g * R + k].y *= w[k].y; }
            radix_small<R>(a + g * R);
#pragma unroll
            for (int r = 1; r < R; ++r) a[g * R + r] = cmul(a[g * R + r], t1[r - 1]);
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) lds_put(x1w + 8 * v, a[v]);
        wave_lds_fence();
#pragma unroll
        for (int b = 0; b < 8; ++b) a[b] = lds_get(x1r + b * kS1);
        wave_lds_fence();

        if (SG_RSMALL_PRIO) __builtin_amdgcn_s_setprio(1);
        radix8(a);
#pragma unroll
        for (int s = 1; s < 8; ++s) a[s] = cmul(a[s], t2[s - 1]);
#pragma unroll
        for (int s = 0; s < 8; ++s) lds_put(x2w + R * s, a[s]);
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = lds_get(x2r + j * kS2);
        wave_lds_fence();

        if (SG_RSMALL_PRIO) __builtin_amdgcn_s_setprio(2);
        radix8(a);
#pragma unroll
        for (int t = 4; t < 8; ++t) lds_put(x3w + L * t, a[t]);
        if (lu == 0) lds_put(buf + g3 * RS + M, a[0]);                 // Z_g[M] := Z_g[0]
        wave_lds_fence();

        if (SG_RSMALL_PRIO) __builtin_amdgcn_s_setprio(3);
        const int f = fg + g3;
        const bool live = f < p.n_frames;
        float bsum = 0.f;
        float pk[4], pm[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float2 A = a[t];
            const float2 B = lds_get(x3b - L * t);
            const float2 cs = t3[t];
            const float2 S = make_float2(A.x + B.x, A.y - B.y);
            const float2 D = make_float2(A.x - B.x, A.y + B.y);
            const float2 T = make_float2(fmaf(cs.y, D.x, -cs.x * D.y), fmaf(cs.x, D.x, cs.y * D.y));
            const float2 Xk = csub(S, T), Xm = cadd(S, T);
            pk[t] = fmaf(Xk.x, Xk.x, Xk.y * Xk.y);
            pm[t] = fmaf(Xm.x, Xm.x, Xm.y * Xm.y);
            if (MODE != 1 && t == 0) { pk[t] *= r0; pm[t] *= r0; }
            if (MODE == 1) { pk[t] = sqrtf(pk[t]); pm[t] = sqrtf(pm[t]); }
            const int k = lu + L * t;
            if (MODE == 2) {
                if (k >= p.k_lo && k <= p.k_hi) bsum += pk[t];
                if (M - k >= p.k_lo && M - k <= p.k_hi) bsum += pm[t];
            }
        }
        float pq = fmaf(a[4].x, a[4].x, a[4].y * a[4].y) * 4.0f;      // k = M/2 pairs with itself: lane lu == 0 holds Z[M/2]
        if (MODE == 1) pq = sqrtf(pq);
        if (MODE == 2) {
            if (lu == 0 && M / 2 >= p.k_lo && M / 2 <= p.k_hi) bsum += pq;
            bsum = group_sum<L>(bsum);
            if (live && lu == 0) p.out[static_cast<int64_t>(clip) * p.out_clip_stride + f] = bsum;
        } else {
            // The G rows of a group are G * NB consecutive floats in HBM (consecutive frames of one clip).  A lane holds bins of ONE
            // frame L apart, so storing from here writes G segments of 4L bytes per instruction, none of them aligned (a row is
            // 4 * NB = 516 / 1028 bytes): twice the 64-byte write requests of a contiguous store.  The rows go through the slab
            // instead (the split pass has read what it needs from it) and leave as G * NB consecutive floats, 256 bytes per
            // instruction: nfft 256 / hop 64 127.7 -> see profiles/r03_rsmall_rows.txt.
            wave_lds_fence();
            float* const stage = reinterpret_cast<float*>(buf);
            float* const mine = stage + g3 * NB + lu;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                mine[L * t] = pk[t];
                stage[g3 * NB + M - lu - L * t] = pm[t];
            }
            if (lu == 0) stage[g3 * NB + M / 2] = pq;
            wave_lds_fence();
            const int n_live = min(G, p.n_frames - fg) * NB;            // (wave-uniform; a partial last group writes fewer rows)
            float* const obase = p.out + static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(fg) * NB;
#if SG_RSMALL_X4
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            typedef v4f_ v4f_u __attribute__((aligned(4)));
#pragma unroll
            for (int i = 0; i < (G * NB + 255) / 256; ++i) {           // 4 floats per lane and instruction; the last quad of the live rows may be partial
                const int idx = 4 * (lane + 64 * i);
                if (idx + 3 < n_live) {
                    *reinterpret_cast<v4f_u*>(obase + idx) = *(const __attribute__((address_space(3))) volatile v4f_*)(stage + idx);
                } else {
#pragma unroll
                    for (int e = 0; e < 3; ++e) if (idx + e < n_live) obase[idx + e] = stage[idx + e];
                }
            }
#else
#pragma unroll
            for (int i = 0; i < (G * NB + 63) / 64; ++i) {
                const int idx = lane + 64 * i;
                if (idx < n_live) obase[idx] = stage[idx];
            }
#endif
        }
        wave_lds_fence();
        clip = clip_n;
        gi = gi_n;
    }
}

template <int R, bool DETREND>
int launch_rd(const SmallParams& prm, int n_wg, hipStream_t s, int mode, bool band) {
    if (band) hipLaunchKernelGGL((stft_rsmall_kernel<R, DETREND, 2>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    else if (mode == SG_MODE_PSD) hipLaunchKernelGGL((stft_rsmall_kernel<R, DETREND, 0>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    else hipLaunchKernelGGL((stft_rsmall_kernel<R, DETREND, 1>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft_rsmall launch");
}

template <int R>
int launch_r(const sg_plan& p, const StftArgs& a) {
    constexpr int G = 8 / R;
    SmallParams prm{};
    prm.x = static_cast<const float*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.groups_per_clip = static_cast<int>((a.n_frames + G - 1) / G);
    prm.total_groups = static_cast<int64_t>(prm.groups_per_clip) * a.n_clips;
    int occ = 4;
    if (const char* e = SG_TUNE_ENV("SPECTRO_RSMALL_OCC")) { const int v = atoi(e); if (v >= 1 && v <= 8) occ = v; }     // tuning aid
    int64_t n_waves = static_cast<int64_t>(p.n_cu) * 4 * occ;
    const int64_t by_work = prm.total_groups <= n_waves ? prm.total_groups : (prm.total_groups + 1) / 2;     // small calls: a group per wave
    if (n_waves > by_work) n_waves = by_work;
    if (n_waves < 1) n_waves = 1;
    prm.n_waves = static_cast<int>(n_waves);
    prm.out = static_cast<float*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.win2 = static_cast<const float2*>(p.win_dev);
    prm.tw = static_cast<const float2*>(p.r8_tw_dev);
    prm.scale = static_cast<float>(p.scale);
    prm.k_lo = a.k_lo; prm.k_hi = a.k_hi;
    const int n_wg = static_cast<int>((n_waves + kWaves - 1) / kWaves);
    const bool band = a.band_mode != 0;                    // run_stft has checked: psd plan, 0 <= k_lo <= k_hi < n_bins
    return p.detrend == SG_DETREND_CONSTANT ? launch_rd<R, true>(prm, n_wg, a.stream, p.mode, band)
                                            : launch_rd<R, false>(prm, n_wg, a.stream, p.mode, band);
}

}  // namespace

// The register path needs 8-byte aligned float2 loads; everything else (int16 input, odd hops) is served by the
// Stockham kernel of the same plan.
bool rsmall_can_run(const sg_plan& p, const StftArgs& a) {
    return !a.in_i16 && (p.hop % 2 == 0) && (a.clip_stride % 2 == 0 || a.n_clips == 1) &&
           (reinterpret_cast<uintptr_t>(a.x) % 8 == 0) && a.n_frames <= INT32_MAX;
}

int launch_rsmall(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    return p.nfft == 128 ? launch_r<1>(p, a) : p.nfft == 256 ? launch_r<2>(p, a) : launch_r<4>(p, a);
}

// Per-lane twiddle table [(R-1) + 7 + 4][64] float2 (R = nfft/128), computed in double:
//   rows 0..R-2      t1[r-1][j] = exp(-2*pi*i*j*r/M)                  M = 64R
//   rows R-1..R+5    t2[s-1][j] = exp(-2*pi*i*(j&7)*s/64)
//   rows R+6..R+9    t3[t][j]   = (cos, sin)(2*pi*k/(2M)), k = (j % 8R) + 8R*t
int build_rsmall_tables(sg_plan& p) {
    const int R = p.nfft / 128, M = 64 * R, L = 8 * R;
    std::vector<float2> tw(static_cast<size_t>(R - 1 + 7 + 4) * 64);
    const double two_pi = 6.283185307179586476925286766559;
    for (int j = 0; j < 64; ++j) {
        for (int r = 1; r < R; ++r) {
            const double ang = -two_pi * static_cast<double>((j * r) % M) / M;
            tw[(r - 1) * 64 + j] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
        }
        for (int s = 1; s < 8; ++s) {
            const double ang = -two_pi * static_cast<double>(((j & 7) * s) % 64) / 64.0;
            tw[(R - 1 + s - 1) * 64 + j] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
        }
        for (int t = 0; t < 4; ++t) {
            const double ang = two_pi * static_cast<double>((j % L) + L * t) / (2.0 * M);
            tw[(R - 1 + 7 + t) * 64 + j] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
        }
    }
    SG_HIP(hipMalloc(&p.r8_tw_dev, tw.size() * sizeof(float2)));
    SG_HIP(hipMemcpy(p.r8_tw_dev, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    return SG_OK;
}

}  // namespace sg
