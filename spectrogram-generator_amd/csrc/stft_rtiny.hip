// stft_rtiny.hip -- register kernel for the smallest transforms: nperseg = nfft = 64 (Q = 4) or 32 (Q = 2), f32 ("rtiny") and f64 ("rtinyd"),
// detrend none|constant, psd|magnitude, fused band power.  (The GUI's nperseg spin box starts at 32 and steps by 32,
// /root/reference/GUI.py:87-89; until round 4 these two sizes ran on the Stockham LDS kernel at 0.8 TB/s algorithmic.)
//
// The real frame is packed into N2 = n/2 = 8 Q complex points.  Q lanes own a frame -- a wave carries G = 64 / Q = 16 / 32 frames per step --
// and lane j of them holds z[j + Q r], r = 0..7, in registers:
//     Z[k1 + 8 k2] = sum_j W_Q^(j k2) * ( W_N2^(j k1) * sum_r z[j + Q r] W_8^(r k1) )
// = a radix-8 pass in registers, seven per-lane twiddles, and a Q-point DFT ACROSS the Q lanes of a quad, which DPP quad permutes do without
// LDS (Q = 4: two butterfly stages, lane j ends with k2 = bit-reversed j; Q = 2: one).  The real-input split pairs bin k with N2 - k =
// (8 - k1) + 8 (Q - 1 - k2): the mirror lane of the same quad -- DPP again.  So the transform itself touches no LDS and has no barrier; LDS
// only stages the G finished rows (G * (N2 + 1) consecutive values in HBM) so that they leave as contiguous 16-byte stores, as in
// stft_rsmall.hip.  Lane-level model of the index maps: tools/sim_rtiny.py.
//
// nperseg 96 / 160 / 192 / 224 (Q = 6 / 10 / 12 / 14: the spin box's sizes below 256 that are no power of two; the chirp-z kernel ran
// them on 512-point transforms): the same frame layout on Q lanes (floor(64 / Q) frames per wave step, the last 4 or 8 lanes idle), the Q-point
// DFT across the frame's lanes as a direct sum through LDS (Q complex multiply-adds per bin, the Q coefficients W_Q^(j k2) in registers), the
// split's mirror lane by ds_bpermute.
// Algorithmic HBM bytes per frame: hop * s + (n/2 + 1) * s, s = 4 / 8.
#include "spectro_internal.h"

#include <cmath>
#include <vector>

namespace sg {
namespace {

constexpr int kWaves = 4;                                    // per workgroup

template <typename R> struct cx { R x, y; };
template <typename R> __device__ __forceinline__ cx<R> cadd(cx<R> a, cx<R> b) { return {a.x + b.x, a.y + b.y}; }
template <typename R> __device__ __forceinline__ cx<R> csub(cx<R> a, cx<R> b) { return {a.x - b.x, a.y - b.y}; }
template <typename R> __device__ __forceinline__ cx<R> cmul(cx<R> a, cx<R> w) { return {fma(a.x, w.x, -a.y * w.y), fma(a.x, w.y, a.y * w.x)}; }
template <typename R> __device__ __forceinline__ cx<R> mul_mi(cx<R> a) { return {a.y, -a.x}; }

template <typename R> __device__ __forceinline__ void radix8(cx<R> (&a)[8]) {      // forward 8-point DFT in registers (fft_wave.h)
    const R h = static_cast<R>(0.70710678118654752440);
    const cx<R> b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]);
    const cx<R> b1 = cadd(a[1], a[5]), b5 = csub(a[1], a[5]);
    const cx<R> b2 = cadd(a[2], a[6]), b6 = csub(a[2], a[6]);
    const cx<R> b3 = cadd(a[3], a[7]), b7 = csub(a[3], a[7]);
    const cx<R> t5 = {b5.x + b5.y, b5.y - b5.x};
    const cx<R> t6 = mul_mi(b6);
    const cx<R> t7 = {b7.y - b7.x, -(b7.x + b7.y)};
    const cx<R> c0 = cadd(b0, b2), c2 = csub(b0, b2);
    const cx<R> c1 = cadd(b1, b3), c3 = mul_mi(csub(b1, b3));
    const cx<R> c4 = cadd(b4, t6), c6 = csub(b4, t6);
    const cx<R> c5 = cadd(t5, t7), c7 = mul_mi(csub(t5, t7));
    a[0] = cadd(c0, c1); a[4] = csub(c0, c1);
    a[2] = cadd(c2, c3); a[6] = csub(c2, c3);
    a[1] = {fma(h, c5.x, c4.x), fma(h, c5.y, c4.y)};
    a[5] = {fma(-h, c5.x, c4.x), fma(-h, c5.y, c4.y)};
    a[3] = {fma(h, c7.x, c6.x), fma(h, c7.y, c6.y)};
    a[7] = {fma(-h, c7.x, c6.x), fma(-h, c7.y, c6.y)};
}

// DPP quad permutes: lane l reads from lane (l & ~3) + perm[l & 3]; CTRL = perm[0] | perm[1] << 2 | perm[2] << 4 | perm[3] << 6
constexpr int kXor1 = 0xB1, kXor2 = 0x4E, kMirror = 0x1B, kSwapHi = 0xB4;      // [1,0,3,2] [2,3,0,1] [3,2,1,0] [0,1,3,2]
template <int CTRL> __device__ __forceinline__ float quad(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double quad(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(u), CTRL, 0xf, 0xf, true));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(u >> 32), CTRL, 0xf, 0xf, true));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}
template <int CTRL, typename R> __device__ __forceinline__ cx<R> quad(cx<R> v) { return {quad<CTRL>(v.x), quad<CTRL>(v.y)}; }

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {                     // fft_wave.h
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename R> struct TinyParams {
    const R* x;
    int64_t clip_stride;
    int n_frames, hop;
    int groups_per_clip;      // ceil(n_frames / G)
    int64_t total_groups;
    int n_waves;
    R* out;
    int64_t out_clip_stride;
    const R* win;             // [n]
    const R* tw;              // [16 (+ Q)][64][2]: rows 0..7 W_N2^(j k1) (row 0 = 1), rows 8..15 exp(-2 pi i (k1 + 8 k2) / n), rows 16.. W_Q^(jj k2), per lane
    R scale;
    int k_lo, k_hi;           // MODE 2: bins of the band
};

// MODE 0 psd, 1 magnitude, 2 band power: out[clip][frame] = sum of PSD bins [k_lo, k_hi] (A11, the spectrum is never written)
template <typename R, int Q> constexpr int kOcc = (sizeof(R) == 4 && (Q & (Q - 1)) == 0) ? 4 : 2;      // waves per SIMD: 116-130 VGPRs in f32 (held to 128), 176-204 in f64; two where Q coefficients join them

template <typename R, int Q, bool DETREND, int MODE>
__global__ __launch_bounds__(64 * kWaves, (kOcc<R, Q>)) void stft_rtiny_kernel(const TinyParams<R> p) {
    constexpr int G = 64 / Q, N2 = 8 * Q, NB = N2 + 1;
    constexpr bool kGen = (Q & (Q - 1)) != 0;                // Q no power of two: LDS / bpermute instead of DPP quad permutes
    constexpr int kStage = kGen ? 16 * G * Q + 8 : ((G * NB + 3) / 4) * 4 + 4;       // values per wave (kGen: the cross-lane DFT's 8 G Q complex values, then the rows)
    constexpr int kVec = 16 / sizeof(R);                     // values per 16-byte store
    __shared__ __attribute__((aligned(16))) R lds[kWaves * kStage];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    R* const stage = lds + wave * kStage;

    const int lw = xcd_remap(blockIdx.x, gridDim.x) * kWaves + wave;
    if (lw >= p.n_waves) return;
    int64_t q = p.total_groups * lw / p.n_waves;
    const int64_t q_end = p.total_groups * (lw + 1) / p.n_waves;

    const bool live = lane < G * Q;                          // (kGen: the last lanes of the wave own no frame; they shadow frame G - 1 and store nothing)
    const int j = lane % Q, g = live ? lane / Q : G - 1;
    const int k2 = Q == 4 ? ((j & 1) << 1) | (j >> 1) : j;   // the bin block this lane ends with
    // per-lane constants; sqrt of the PSD scale rides on the window (stft_r8x3.hip), bins 0 and N2 get 1/2 below
    const R sq = sqrt(MODE != 1 ? p.scale * static_cast<R>(0.5) : p.scale * static_cast<R>(0.25));
    cx<R> w[8], t1[8], st[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        w[r] = {p.win[2 * (j + Q * r)] * sq, p.win[2 * (j + Q * r) + 1] * sq};
        t1[r] = {p.tw[2 * (r * 64 + lane)], p.tw[2 * (r * 64 + lane) + 1]};
        st[r] = {p.tw[2 * ((8 + r) * 64 + lane)], p.tw[2 * ((8 + r) * 64 + lane) + 1]};
    }
    cx<R> tq[kGen ? Q : 1];                                  // kGen: W_Q^(jj k2), jj < Q
#pragma unroll
    for (int jj = 0; jj < (kGen ? Q : 0); ++jj) tq[jj] = {p.tw[2 * ((16 + jj) * 64 + lane)], p.tw[2 * ((16 + jj) * 64 + lane) + 1]};
    const int mirror = g * Q + (Q - 1 - j), mirror0 = g * Q + (Q - j) % Q;      // kGen: the lanes that hold bins N2 - k (k1 >= 1 / k1 = 0)
    // the quad DFT's per-lane coefficients: Q = 4 stage A u = partner + sA own; stage B out = A u + B partner(u);  Q = 2: out = partner + sA own
    const R sA = Q == 4 ? ((j & 2) ? -1 : 1) : (j ? -1 : 1);
    // lane 0: A = 1, B = 1;  lane 1: A = -1, B = 1;  lane 2: A = 1, B = -i;  lane 3: A = i, B = 1
    const cx<R> cA = {static_cast<R>(j == 3 ? 0 : j == 1 ? -1 : 1), static_cast<R>(j == 3 ? 1 : 0)};
    const cx<R> cB = {static_cast<R>(j == 2 ? 0 : 1), static_cast<R>(j == 2 ? -1 : 0)};
    const R r0 = (MODE != 1 && k2 == 0) ? static_cast<R>(0.5) : static_cast<R>(1);      // bin 0 (register 0 of the k2 = 0 lane); bin N2 likewise

    auto frame_sum = [&](R v) {                              // over the Q lanes of a frame, the same order in every lane
        if (!kGen) {
            v += quad<kXor1>(v);
            if (Q == 4) v += quad<kXor2>(v);
            return v;
        }
        if (live) stage[g * Q + j] = v;
        wave_lds_fence();
        R s = stage[g * Q];
#pragma unroll
        for (int jj = 1; jj < Q; ++jj) s += stage[g * Q + jj];
        wave_lds_fence();
        return s;
    };
    auto load_group = [&](int clip, int gi, cx<R> (&dst)[8]) {
        const int f = min(gi * G + g, p.n_frames - 1);       // partial last group: recompute the last frame
        const R* const src = p.x + static_cast<int64_t>(clip) * p.clip_stride + static_cast<int64_t>(f) * p.hop + 2 * j;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            typedef R v2 __attribute__((ext_vector_type(2)));
            const v2 v = *reinterpret_cast<const v2*>(src + 2 * Q * r);
            dst[r] = {v.x, v.y};
        }
    };
    int clip = static_cast<int>(q / p.groups_per_clip);
    int gi = static_cast<int>(q - static_cast<int64_t>(clip) * p.groups_per_clip);
    cx<R> nxt[8];
    if (q < q_end) load_group(clip, gi, nxt);

    for (; q < q_end; ++q) {
        const int fg = gi * G;
        const int clip_n = gi + 1 == p.groups_per_clip ? clip + 1 : clip, gi_n = gi + 1 == p.groups_per_clip ? 0 : gi + 1;
        const bool more = q + 1 < q_end;                     // the run's last group fetches itself again
        cx<R> a[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = nxt[r];
        load_group(more ? clip_n : clip, more ? gi_n : gi, nxt);
        if (DETREND) {                                       // A3: the frame's mean over its Q lanes
            R s = a[0].x + a[0].y;
#pragma unroll
            for (int r = 1; r < 8; ++r) s += a[r].x + a[r].y;
            s = frame_sum(s);
            const R mean = kGen ? s / static_cast<R>(2 * N2) : s * (static_cast<R>(1) / (2 * N2));     // (a true division where n is no power of two: a constant clip detrends to 0, as in scipy)
#pragma unroll
            for (int r = 0; r < 8; ++r) { a[r].x -= mean; a[r].y -= mean; }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) { a[r].x *= w[r].x; a[r].y *= w[r].y; }
        radix8(a);                                           // over r -> k1
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) a[k1] = cmul(a[k1], t1[k1]);
        if (kGen) {                                          // Q-point DFT across the frame's lanes, a direct sum through LDS
            cx<R>* const cbuf = reinterpret_cast<cx<R>*>(stage);
            if (live) {
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) cbuf[(g * 8 + k1) * Q + j] = a[k1];
            }
            wave_lds_fence();
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) {
                cx<R> acc = cbuf[(g * 8 + k1) * Q];          // jj = 0: coefficient 1
#pragma unroll
                for (int jj = 1; jj < (kGen ? Q : 1); ++jj) {
                    const cx<R> b = cbuf[(g * 8 + k1) * Q + jj];
                    acc.x = fma(b.x, tq[jj].x, fma(-b.y, tq[jj].y, acc.x));
                    acc.y = fma(b.x, tq[jj].y, fma(b.y, tq[jj].x, acc.y));
                }
                a[k1] = acc;
            }
            wave_lds_fence();
        }
#pragma unroll
        for (int k1 = 0; k1 < (kGen ? 0 : 8); ++k1) {        // Q-point DFT across the quad's lanes -> this lane holds Z[k1 + 8 k2]
            const cx<R> pa = Q == 4 ? quad<kXor2>(a[k1]) : quad<kXor1>(a[k1]);
            const cx<R> u = {fma(sA, a[k1].x, pa.x), fma(sA, a[k1].y, pa.y)};
            if (Q == 4) {
                const cx<R> pb = quad<kXor1>(u);
                a[k1] = cadd(cmul(u, cA), cmul(pb, cB));
            } else {
                a[k1] = u;
            }
        }
        // ---- split + epilogue: bin k = k1 + 8 k2 pairs with N2 - k: register 8 - k1 of the mirror lane (k1 = 0: register 0 of the lane with (Q - k2) % Q)
        R pk[8], bsum = 0;
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            const cx<R> A = a[k1];
            const cx<R> B = kGen ? cx<R>{__shfl(a[(8 - k1) % 8].x, k1 == 0 ? mirror0 : mirror), __shfl(a[(8 - k1) % 8].y, k1 == 0 ? mirror0 : mirror)}
                          : k1 == 0 ? (Q == 4 ? quad<kSwapHi>(a[0]) : a[0]) : (Q == 4 ? quad<kMirror>(a[8 - k1]) : quad<kXor1>(a[8 - k1]));
            const cx<R> S = {A.x + B.x, A.y - B.y};
            const cx<R> D = {A.x - B.x, A.y + B.y};
            const cx<R> X = {S.x + fma(st[k1].x, D.y, st[k1].y * D.x), S.y + fma(st[k1].y, D.y, -st[k1].x * D.x)};
            R v = fma(X.x, X.x, X.y * X.y);
            if (k1 == 0) v *= r0;
            if (MODE == 1) v = sqrt(v);
            pk[k1] = v;
            if (MODE == 2) { const int k = k1 + 8 * k2; if (k >= p.k_lo && k <= p.k_hi) bsum += v; }
        }
        R pn = (a[0].x - a[0].y) * (a[0].x - a[0].y) * static_cast<R>(MODE != 1 ? 2 : 4);      // bin N2 from Z[0]: X = 2 (Re - Im), |X|^2 / 2 (meaningful on the k2 = 0 lane)
        if (MODE == 1) pn = sqrt(pn);
        const int f = fg + g;
        if (MODE == 2) {
            if (k2 == 0 && N2 >= p.k_lo && N2 <= p.k_hi) bsum += pn;
            bsum = frame_sum(bsum);
            if (live && j == 0 && f < p.n_frames) p.out[static_cast<int64_t>(clip) * p.out_clip_stride + f] = bsum;
        } else {
            // the G rows of a group are G * NB consecutive values in HBM: through the slab, out as contiguous 16-byte stores (stft_rsmall.hip)
            R* const mine = stage + g * NB + 8 * k2;
            if (live) {
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) mine[k1] = pk[k1];
                if (k2 == 0) stage[g * NB + N2] = pn;
            }
            wave_lds_fence();
            const int n_live = min(G, p.n_frames - fg) * NB;  // (wave-uniform; a partial last group writes fewer rows)
            R* const obase = p.out + static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(fg) * NB;
            typedef R vv __attribute__((ext_vector_type(kVec)));
            typedef vv vv_u __attribute__((aligned(sizeof(R))));
#pragma unroll
            for (int i = 0; i < (G * NB + 64 * kVec - 1) / (64 * kVec); ++i) {
                const int idx = kVec * (lane + 64 * i);
                if (idx + kVec - 1 < n_live) {
                    *reinterpret_cast<vv_u*>(obase + idx) = *reinterpret_cast<const vv*>(stage + idx);
                } else {
#pragma unroll
                    for (int e = 0; e < kVec - 1; ++e) if (idx + e < n_live) obase[idx + e] = stage[idx + e];
                }
            }
            wave_lds_fence();
        }
        clip = clip_n;
        gi = gi_n;
    }
}

template <typename R, int Q, bool DETREND>
int launch_mode(const TinyParams<R>& prm, int n_wg, hipStream_t s, int mode, bool band) {
    if (band) hipLaunchKernelGGL((stft_rtiny_kernel<R, Q, DETREND, 2>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    else if (mode == SG_MODE_PSD) hipLaunchKernelGGL((stft_rtiny_kernel<R, Q, DETREND, 0>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    else hipLaunchKernelGGL((stft_rtiny_kernel<R, Q, DETREND, 1>), dim3(n_wg), dim3(64 * kWaves), 0, s, prm);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft_rtiny launch");
}

template <typename R, int Q>
int launch_q(const sg_plan& p, const StftArgs& a) {
    constexpr int G = 64 / Q;
    TinyParams<R> prm{};
    prm.x = static_cast<const R*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.groups_per_clip = static_cast<int>((a.n_frames + G - 1) / G);
    prm.total_groups = static_cast<int64_t>(prm.groups_per_clip) * a.n_clips;
    int64_t n_waves = static_cast<int64_t>(p.n_cu) * 4 * kOcc<R, Q>;
    const int64_t by_work = prm.total_groups <= n_waves ? prm.total_groups : (prm.total_groups + 1) / 2;     // small calls: a group per wave
    if (n_waves > by_work) n_waves = by_work;
    if (n_waves < 1) n_waves = 1;
    prm.n_waves = static_cast<int>(n_waves);
    prm.out = static_cast<R*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.win = static_cast<const R*>(p.win_dev);
    prm.tw = static_cast<const R*>(p.r8_tw_dev);
    prm.scale = static_cast<R>(p.scale);
    prm.k_lo = a.k_lo; prm.k_hi = a.k_hi;
    const int n_wg = static_cast<int>((n_waves + kWaves - 1) / kWaves);
    const bool band = a.band_mode != 0;                    // run_stft has checked: psd plan, 0 <= k_lo <= k_hi < n_bins
    return p.detrend == SG_DETREND_CONSTANT ? launch_mode<R, Q, true>(prm, n_wg, a.stream, p.mode, band)
                                            : launch_mode<R, Q, false>(prm, n_wg, a.stream, p.mode, band);
}

}  // namespace

// The register path needs aligned two-sample loads (8 bytes in f32, 16 in f64); everything else (int16 input, odd hops) is served by the
// Stockham kernel of the same plan.
bool rtiny_can_run(const sg_plan& p, const StftArgs& a) {
    const size_t pair = p.dtype == SG_F64 ? 16 : 8;
    return !a.in_i16 && !a.db_mode && a.mel_ipl == 0 && (p.hop % 2 == 0) && (a.clip_stride % 2 == 0 || a.n_clips == 1) &&
           (reinterpret_cast<uintptr_t>(a.x) % pair == 0) && a.n_frames <= INT32_MAX;
}

int launch_rtiny(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    if (p.dtype == SG_F64) {
        switch (p.nfft) {
            case 32: return launch_q<double, 2>(p, a);
            case 64: return launch_q<double, 4>(p, a);
            case 96: return launch_q<double, 6>(p, a);
            case 160: return launch_q<double, 10>(p, a);
            case 192: return launch_q<double, 12>(p, a);
            default: return launch_q<double, 14>(p, a);      // 224
        }
    }
    switch (p.nfft) {
        case 32: return launch_q<float, 2>(p, a);
        case 64: return launch_q<float, 4>(p, a);
        case 96: return launch_q<float, 6>(p, a);
        case 160: return launch_q<float, 10>(p, a);
        case 192: return launch_q<float, 12>(p, a);
        default: return launch_q<float, 14>(p, a);           // 224
    }
}

// Per-lane twiddle table [16 (+ Q)][64] complex (computed in long double), lane = Q g + j, k2 = the bin block of lane j:
//   rows 0..7    exp(-2 pi i j k1 / N2)
//   rows 8..15   exp(-2 pi i (k1 + 8 k2) / n)
//   rows 16..    (Q no power of two) exp(-2 pi i jj k2 / Q), jj < Q
int build_rtiny_tables(sg_plan& p) {
    const int n = p.nfft, N2 = n / 2, Q = N2 / 8;
    const bool gen = (Q & (Q - 1)) != 0;
    const int rows = 16 + (gen ? Q : 0);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    std::vector<long double> t(static_cast<size_t>(rows) * 64 * 2);
    for (int lane = 0; lane < 64; ++lane) {
        const int j = lane % Q, k2 = Q == 4 ? ((j & 1) << 1) | (j >> 1) : j;
        for (int k1 = 0; k1 < 8; ++k1) {
            const long double a1 = -two_pi * static_cast<long double>((j * k1) % N2) / N2;
            const long double a2 = -two_pi * static_cast<long double>(k1 + 8 * k2) / n;
            t[2 * (k1 * 64 + lane)] = cosl(a1); t[2 * (k1 * 64 + lane) + 1] = sinl(a1);
            t[2 * ((8 + k1) * 64 + lane)] = cosl(a2); t[2 * ((8 + k1) * 64 + lane) + 1] = sinl(a2);
        }
        for (int jj = 0; gen && jj < Q; ++jj) {
            const long double a3 = -two_pi * static_cast<long double>((jj * k2) % Q) / Q;
            t[2 * ((16 + jj) * 64 + lane)] = cosl(a3); t[2 * ((16 + jj) * 64 + lane) + 1] = sinl(a3);
        }
    }
    if (p.dtype == SG_F64) {
        std::vector<double> h(t.begin(), t.end());
        SG_HIP(hipMalloc(&p.r8_tw_dev, h.size() * sizeof(double)));
        SG_HIP(hipMemcpy(p.r8_tw_dev, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    } else {
        std::vector<float> h(t.size());
        for (size_t i = 0; i < t.size(); ++i) h[i] = static_cast<float>(t[i]);
        SG_HIP(hipMalloc(&p.r8_tw_dev, h.size() * sizeof(float)));
        SG_HIP(hipMemcpy(p.r8_tw_dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return SG_OK;
}

}  // namespace sg
