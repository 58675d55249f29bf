// stft_rbig.hip -- register FFT kernel for the large transforms of the parameter sweep (BASELINE cfg4):
// nperseg = nfft = 1024*T with T = 2 (2048) or T = 4 (4096), f32, detrend none|constant, psd|magnitude.
//
// One wavefront = one frame; the packed signal has M = 512*T complex points, lane j keeps z[j + 64a] for
// a < R = 8T in VGPRs (d[a0][a1], a = a0 + T*a1).  Three register passes, two padded LDS transposes:
//   pass 1: R-point DFT in registers = T radix-8 butterflies over a1, constant twiddles w_R^(a0*r1), 8 radix-T
//           butterflies over a0; then the lane twiddle w_M^(j*r)              (r = r1 + 8*r0 lives in d[r0][r1])
//   pass 2: lane j0 + 8*r1 runs T radix-8 butterflies over b (j = j0 + 8b), twiddle w_64^(j0*s)
//   pass 3: lane l runs T radix-8 butterflies over j0 and ends with Z[l + 64c], c = q3 + T*t in d[q3][t]
// and the split pass pairs k = l + 64c (c < R/2, registers) with M - k (upper half, through LDS).
// Window and twiddle tables live in LDS (one copy per workgroup).  Index maps and bank behaviour: tools/sim_rbig.py.
// Every exchange moves 8 registers per lane at a time through ONE 4.5 KiB slab per wave (a wave's DS operations
// execute in issue order, so the slab is reused back to back without waiting): 12 (T = 4) / 16 (T = 2) waves per CU
// instead of the 6 a whole-frame slab allowed -- the kernel was latency-bound at 1.5 waves per SIMD.
// Round 2: at hops 128 / 256 (and 64 / 32 / 16 walked as interleaved hop-128 sequences, stft_r8x3.hip) the frame's samples stay in
// registers and slide by one or two 128-sample blocks per frame: one or two 8-byte loads per lane and frame instead of 8T from
// L1 / L2 -- a reloaded sample costs about what an HBM byte costs on this part (nfft 2048: -25...30 %; nfft 4096, at 2 waves per
// SIMD for the 64 extra VGPRs: -13...16 %).  The fused band power (A11) is one more output form.
// Algorithmic HBM bytes per frame: hop*4 + (512T+1)*4.
#include "spectro_internal.h"
#include "fft_wave.h"

#include <cmath>
#include <cstdlib>

namespace sg {
namespace {

using namespace wavefft;

constexpr int kS1 = 72, kS2 = 66;
typedef float v4f __attribute__((ext_vector_type(4)));
// two neighbouring float2 of the tables in one 16-byte LDS read (volatile for the reason lds_get is, fft_wave.h)
__device__ __forceinline__ v4f lds_get2(const float2* p) { return *(__attribute__((address_space(3))) volatile v4f*)(p); }
// waves per workgroup (the LDS tables are shared by the workgroup): T = 4 -> one workgroup of 12 waves per CU
// (43 KiB of tables + 12 x 4.5 KiB, 142 VGPRs = 3 waves/SIMD); T = 2 -> two workgroups of 8 (21 + 36 KiB each)
#ifndef SG_RBIG_PRIO
#define SG_RBIG_PRIO 1          // wave priority rises along a frame (pass 1 -> stores), as in stft_r8x3; 0 = off
#endif
#ifndef SG_RBIG_TW2_REG
#define SG_RBIG_TW2_REG 1          // the seven pass-2 twiddles in VGPRs where the occupancy has room (T = 4 sliding: 236 -> 254 of 256): 28 LDS reads fewer per
                                   // frame; nfft 4096 hop 64 / 128 / 256: -0.3 / -1.3 / -2.4 % (profiles/r03_rbig_tw2_registers.txt); 0 = all from LDS
#endif
#ifndef SG_RBIG_B128
#define SG_RBIG_B128 1             // window, t1 and t3 tables keep rows 2m, 2m+1 of a lane side by side: one ds_read_b128 per two rows (a b128 read moves twice
                                   // the bytes in 1.3x the time); -1 ... -2.7 % on every shape (profiles/r03_rbig_b128_tables.txt); 0 = one ds_read_b64 per row
#endif
#ifndef SG_RBIG_SADDR
#define SG_RBIG_SADDR 1            // the row pointer is made wave-uniform (v_readfirstlane of its offset) so that the 8T + 1 stores of a row take the
                                   // `global_store_dword v_offset, v_data, s[base:base+1] offset:imm` form: one address VGPR per store instead of a 64-bit
                                   // pair and no 64-bit VALU address arithmetic (timing what-ifs: the row stores are 27 % of the nfft-4096 launch)
#endif
__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(static_cast<uint64_t>(v) >> 32));
    return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}
#ifndef SG_RBIG4_OCC
#define SG_RBIG4_OCC 3            // waves per SIMD the T = 4 kernel is compiled for
#endif
// H > 0 (the sample window slides in registers, see the frame loop): T = 4 keeps 32 more float2 per lane -> 2 waves per SIMD
template <int T, int H> struct OccFor { static constexpr int value = T == 4 ? (H > 0 ? 2 : SG_RBIG4_OCC) : 4; };
template <int T, int H> struct WavesFor { static constexpr int value = T == 4 ? 4 * OccFor<T, H>::value : 8; };
constexpr int kSlabElems = 8 * kS1;                          // one 8-register exchange group (576 float2)

struct BigParams {
    const float* x;
    int64_t clip_stride;
    int n_frames, hop;
    int sub;                  // hop * sub is a multiple of 128 (H > 0): a clip's frames are walked as `sub` interleaved sequences
                              // (frames v, v + sub, ...) whose window slides in registers (stft_r8x3.hip); 1 otherwise
    int64_t total_frames;
    int n_waves;
    float* out;
    int64_t out_clip_stride;
    const float2* win2;       // [M]
    const float2* tw;         // [(R-1) + 7 + R/2][64]
    float scale;
    int k_lo, k_hi;           // MODE 2: bins of the band
};

template <int T> __device__ __forceinline__ void radix_t(float2 (&v)[T]);
template <> __device__ __forceinline__ void radix_t<2>(float2 (&v)[2]) {
    const float2 s = cadd(v[0], v[1]), d = csub(v[0], v[1]);
    v[0] = s; v[1] = d;
}
template <> __device__ __forceinline__ void radix_t<4>(float2 (&v)[4]) {
    const float2 s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
    const float2 s13 = cadd(v[1], v[3]), d13 = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(s02, s13); v[2] = csub(s02, s13);
    v[1] = cadd(d02, d13); v[3] = csub(d02, d13);
}

// exp(-2*pi*i*n/R) for the in-register twiddles of pass 1; indices are compile-time after unrolling, so these
// fold into instruction immediates.
__device__ constexpr float kW16[16][2] = {{1.000000000e+00f, -0.000000000e+00f}, {9.238795325e-01f, -3.826834324e-01f}, {7.071067812e-01f, -7.071067812e-01f}, {3.826834324e-01f, -9.238795325e-01f}, {6.123233996e-17f, -1.000000000e+00f}, {-3.826834324e-01f, -9.238795325e-01f}, {-7.071067812e-01f, -7.071067812e-01f}, {-9.238795325e-01f, -3.826834324e-01f}, {-1.000000000e+00f, -1.224646799e-16f}, {-9.238795325e-01f, 3.826834324e-01f}, {-7.071067812e-01f, 7.071067812e-01f}, {-3.826834324e-01f, 9.238795325e-01f}, {-1.836970199e-16f, 1.000000000e+00f}, {3.826834324e-01f, 9.238795325e-01f}, {7.071067812e-01f, 7.071067812e-01f}, {9.238795325e-01f, 3.826834324e-01f}};
__device__ constexpr float kW32[32][2] = {{1.000000000e+00f, -0.000000000e+00f}, {9.807852804e-01f, -1.950903220e-01f}, {9.238795325e-01f, -3.826834324e-01f}, {8.314696123e-01f, -5.555702330e-01f}, {7.071067812e-01f, -7.071067812e-01f}, {5.555702330e-01f, -8.314696123e-01f}, {3.826834324e-01f, -9.238795325e-01f}, {1.950903220e-01f, -9.807852804e-01f}, {6.123233996e-17f, -1.000000000e+00f}, {-1.950903220e-01f, -9.807852804e-01f}, {-3.826834324e-01f, -9.238795325e-01f}, {-5.555702330e-01f, -8.314696123e-01f}, {-7.071067812e-01f, -7.071067812e-01f}, {-8.314696123e-01f, -5.555702330e-01f}, {-9.238795325e-01f, -3.826834324e-01f}, {-9.807852804e-01f, -1.950903220e-01f}, {-1.000000000e+00f, -1.224646799e-16f}, {-9.807852804e-01f, 1.950903220e-01f}, {-9.238795325e-01f, 3.826834324e-01f}, {-8.314696123e-01f, 5.555702330e-01f}, {-7.071067812e-01f, 7.071067812e-01f}, {-5.555702330e-01f, 8.314696123e-01f}, {-3.826834324e-01f, 9.238795325e-01f}, {-1.950903220e-01f, 9.807852804e-01f}, {-1.836970199e-16f, 1.000000000e+00f}, {1.950903220e-01f, 9.807852804e-01f}, {3.826834324e-01f, 9.238795325e-01f}, {5.555702330e-01f, 8.314696123e-01f}, {7.071067812e-01f, 7.071067812e-01f}, {8.314696123e-01f, 5.555702330e-01f}, {9.238795325e-01f, 3.826834324e-01f}, {9.807852804e-01f, 1.950903220e-01f}};
template <int R> __device__ __forceinline__ float2 const_tw(int n) {
    return R == 16 ? make_float2(kW16[n & 15][0], kW16[n & 15][1]) : make_float2(kW32[n & 31][0], kW32[n & 31][1]);
}

// H: hop * sub == 128 * H and the sample window slides in registers (H blocks of 128 samples per frame); 0: every frame reloads
template <int T, bool DETREND, int MODE, int H>
__global__ __launch_bounds__((64 * WavesFor<T, H>::value), (OccFor<T, H>::value)) void stft_rbig_kernel(const BigParams p) {
    constexpr int R = 8 * T, M = 64 * R, NB = M + 1, kWaves = WavesFor<T, H>::value;
    constexpr int kSlab = kSlabElems;                        // complex elements per wave
    constexpr int kTw1 = M, kTw2 = kTw1 + (SG_RBIG_B128 ? R : R - 1) * 64, kTw3 = kTw2 + 7 * 64, kTabs = kTw3 + (R / 2) * 64;   // B128: t1 padded to an even row count
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2* const buf = lds + kTabs + wave * kSlab;

    // sqrt of the PSD scale rides on the window table (stft_r8x3.hip); per frame only the 1/2 on bins 0 and M is left
    const float q_in = MODE != 1 ? p.scale * 0.5f : p.scale * 0.25f;
    {
        const float sq = sqrtf(q_in);
        for (int i = threadIdx.x; i < M; i += 64 * kWaves) {
            const float2 wv = p.win2[i];
            const int r = i >> 6, l = i & 63;                // row r of lane l; SG_RBIG_B128: rows 2m, 2m+1 of a lane side by side
            lds[SG_RBIG_B128 ? (r >> 1) * 128 + 2 * l + (r & 1) : i] = make_float2(wv.x * sq, wv.y * sq);
        }
    }
    for (int i = threadIdx.x; i < (R - 1 + 7 + R / 2) * 64; i += 64 * kWaves) {
        const int r = i >> 6, l = i & 63;                    // p.tw: rows [0, R-1) t1, [R-1, R+6) t2, [R+6, ...) t3
        int dst = M + i;
        if (SG_RBIG_B128) {                                  // t1 and t3: rows 2m, 2m+1 of a lane side by side (one ds_read_b128); t2 as it was
            if (r < R - 1) dst = kTw1 + (r >> 1) * 128 + 2 * l + (r & 1);
            else if (r < R + 6) dst = kTw2 + (r - (R - 1)) * 64 + l;
            else dst = kTw3 + ((r - (R + 6)) >> 1) * 128 + 2 * l + ((r - (R + 6)) & 1);
        }
        lds[dst] = p.tw[i];
    }
    __syncthreads();

    const int lw = xcd_remap(blockIdx.x, gridDim.x) * kWaves + wave;
    if (lw >= p.n_waves) return;
    int64_t g = p.total_frames * lw / p.n_waves;
    const int64_t g_end = p.total_frames * (lw + 1) / p.n_waves;

    // (SG_RBIG_B128: the window, t1 and t3 tables are read in pairs from their paired layout, see the table fill above)
    [[maybe_unused]] const float2* const wtab = lds + lane;                   // + 64*a
    [[maybe_unused]] const float2* const t1 = lds + kTw1 + lane;              // + 64*(r-1)
    const float2* const t2 = lds + kTw2 + lane;                               // + 64*(s-1)
    [[maybe_unused]] const float2* const t3 = lds + kTw3 + lane;              // + 64*c
    const int j0 = lane & 7, hi = lane >> 3;
    float2* const x1w = buf + hi * kS1 + j0;                 // + 8*r1            (group q)
    float2* const x1r = buf + lane;                          // + b*kS1
    float2* const x2w = buf + j0 * kS2 + hi;                 // + ((8q + R*s) % 64)   (group q3 = (8q + R*s) / 64)
    float2* const x2r = buf + lane;                          // + j*kS2
    float2* const x3w = buf + lane;                          // + 64*slot, 4 upper blocks per group
    const float2* const x3b = buf + (256 - lane);            // - 64*(c - 4i)

    const float r0 = (MODE != 1 && lane == 0) ? 0.5f : 1.0f;
    constexpr bool kTw2Reg = SG_RBIG_TW2_REG && T == 4 && H > 0;
    float2 t2r[kTw2Reg ? 7 : 1];
    if (kTw2Reg) {
#pragma unroll
        for (int s = 0; s < 7; ++s) t2r[s] = lds_get(t2 + 64 * s);
    }

    // one frame: samples d[][] (destroyed) -> row orow
    auto frame = [&](float2 (&d)[T][8], float* const orow_in) {
#if SG_RBIG_SADDR
        float* const orow = p.out + uniform_i64(orow_in - p.out);          // (an offset: the pointer keeps its global address space)
#else
        float* const orow = orow_in;
#endif
        float bsum = 0.f;                                    // MODE 2: this lane's share of the band sum (A11)
        if (DETREND) {
            float s = d[0][0].x + d[0][0].y;
#pragma unroll
            for (int a0 = 0; a0 < T; ++a0)
#pragma unroll
                for (int a1 = 0; a1 < 8; ++a1)
                    if (a0 + a1 > 0) s += d[a0][a1].x + d[a0][a1].y;
            const float mean = wave_sum(s) * (1.0f / (2 * M));
#pragma unroll
            for (int a0 = 0; a0 < T; ++a0)
#pragma unroll
                for (int a1 = 0; a1 < 8; ++a1) { d[a0][a1].x -= mean; d[a0][a1].y -= mean; }
        }
#if SG_RBIG_B128
#pragma unroll
        for (int a0 = 0; a0 < T; a0 += 2)
#pragma unroll
            for (int a1 = 0; a1 < 8; ++a1) {                 // rows a0 + T*a1 (even) and the next one in one 16-byte read
                const v4f w = lds_get2(lds + ((a0 + T * a1) >> 1) * 128 + 2 * lane);
                d[a0][a1].x *= w.x; d[a0][a1].y *= w.y;
                d[a0 + 1][a1].x *= w.z; d[a0 + 1][a1].y *= w.w;
            }
#else
#pragma unroll
        for (int a0 = 0; a0 < T; ++a0)
#pragma unroll
            for (int a1 = 0; a1 < 8; ++a1) {
                const float2 w = lds_get(wtab + 64 * (a0 + T * a1));
                d[a0][a1].x *= w.x; d[a0][a1].y *= w.y;
            }
#endif

        if (SG_RBIG_PRIO) __builtin_amdgcn_s_setprio(0);
        // ---- pass 1: R-point DFT over a = a0 + T*a1 --------------------------------------------------------
#pragma unroll
        for (int a0 = 0; a0 < T; ++a0) {
            radix8(d[a0]);                                 // over a1 -> r1
            if (a0 > 0) {
#pragma unroll
                for (int r1 = 1; r1 < 8; ++r1) d[a0][r1] = cmul(d[a0][r1], const_tw<R>(a0 * r1));
            }
        }
#pragma unroll
        for (int r1 = 0; r1 < 8; ++r1) {                     // over a0 -> r0 ; r = r1 + 8*r0
            float2 v[T];
#pragma unroll
            for (int a0 = 0; a0 < T; ++a0) v[a0] = d[a0][r1];
            radix_t<T>(v);
#pragma unroll
            for (int r0 = 0; r0 < T; ++r0) d[r0][r1] = v[r0];
        }
#if SG_RBIG_B128
#pragma unroll
        for (int i = 0; i < R - 1; i += 2) {                 // table rows i, i + 1 <-> r = i + 1, i + 2
            const v4f w = lds_get2(lds + kTw1 + (i >> 1) * 128 + 2 * lane);
            d[(i + 1) / 8][(i + 1) % 8] = cmul(d[(i + 1) / 8][(i + 1) % 8], make_float2(w.x, w.y));
            if (i + 2 < R) d[(i + 2) / 8][(i + 2) % 8] = cmul(d[(i + 2) / 8][(i + 2) % 8], make_float2(w.z, w.w));
        }
#else
#pragma unroll
        for (int q = 0; q < T; ++q)
#pragma unroll
            for (int r1 = 0; r1 < 8; ++r1)
                if (q + r1 > 0) d[q][r1] = cmul(d[q][r1], lds_get(t1 + 64 * (r1 + 8 * q - 1)));
#endif
#pragma unroll
        for (int q = 0; q < T; ++q) {                        // exchange 1, one group of 8 at a time through the slab
#pragma unroll
            for (int r1 = 0; r1 < 8; ++r1) lds_put(x1w + 8 * r1, d[q][r1]);
            wave_lds_fence();
#pragma unroll
            for (int b = 0; b < 8; ++b) d[q][b] = lds_get(x1r + b * kS1);
            wave_lds_fence();
        }

        if (SG_RBIG_PRIO) __builtin_amdgcn_s_setprio(1);
        // ---- pass 2 ----------------------------------------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < T; ++q) {
            radix8(d[q]);
#pragma unroll
            for (int s = 1; s < 8; ++s) d[q][s] = cmul(d[q][s], kTw2Reg ? t2r[kTw2Reg ? s - 1 : 0] : lds_get(t2 + 64 * (s - 1)));
        }
        float2 e[T][8];                                      // pass-3 operands: e[q3][j]
#pragma unroll
        for (int q3 = 0; q3 < T; ++q3) {                     // exchange 2: group q3 collects the (q, s) with (8q + R*s) / 64 == q3
#pragma unroll
            for (int q = 0; q < T; ++q)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int uu = 8 * q + R * s;            // + r1 (= hi) < 8 never carries into the next 64
                    if (uu / 64 == q3) lds_put(x2w + (uu % 64), d[q][s]);
                }
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 8; ++j) e[q3][j] = lds_get(x2r + j * kS2);
            wave_lds_fence();
        }

        if (SG_RBIG_PRIO) __builtin_amdgcn_s_setprio(2);
        // ---- pass 3: d[q3][t] = Z[lane + 64*(q3 + T*t)] ---------------------------------------------------
#pragma unroll
        for (int q3 = 0; q3 < T; ++q3) radix8(e[q3]);
#define SG_Z(c) e[(c) % T][(c) / T]                          // Z[lane + 64*c]

        if (SG_RBIG_PRIO) __builtin_amdgcn_s_setprio(3);
        // ---- split pass + epilogue: lower blocks c (registers) pair with upper blocks R-1-c (and element 0 of
        //      block R-c) of the mirrored lane; four blocks per trip through the slab ------------------------------
#pragma unroll
        for (int i = 0; i < R / 8; ++i) {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) lds_put(x3w + 64 * sl, SG_Z(R - 4 * i - 4 + sl));
            if (lane == 0) lds_put(buf + 256, i == 0 ? SG_Z(0) : SG_Z(R - 4 * i));   // i = 0: Z[M] := Z[0]
            wave_lds_fence();
            v4f cs2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = 4 * i + cc;
                const float2 A = SG_Z(c);
                const float2 B = lds_get(x3b - 64 * cc);
#if SG_RBIG_B128
                if (cc % 2 == 0) cs2 = lds_get2(lds + kTw3 + (c >> 1) * 128 + 2 * lane);
                const float2 cs = cc % 2 == 0 ? make_float2(cs2.x, cs2.y) : make_float2(cs2.z, cs2.w);
#else
                const float2 cs = lds_get(t3 + 64 * c);
#endif
                const float2 S = make_float2(A.x + B.x, A.y - B.y);
                const float2 D = make_float2(A.x - B.x, A.y + B.y);
                const float2 Tt = make_float2(fmaf(cs.y, D.x, -cs.x * D.y), fmaf(cs.x, D.x, cs.y * D.y));
                const float2 Xk = csub(S, Tt), Xm = cadd(S, Tt);
                float pk = fmaf(Xk.x, Xk.x, Xk.y * Xk.y);
                float pm = fmaf(Xm.x, Xm.x, Xm.y * Xm.y);
                if (MODE != 1 && c == 0) { pk *= r0; pm *= r0; }
                if (MODE == 1) { pk = sqrtf(pk); pm = sqrtf(pm); }
                const int k = lane + 64 * c;
                if (MODE == 2) {
                    if (k >= p.k_lo && k <= p.k_hi) bsum += pk;
                    if (M - k >= p.k_lo && M - k <= p.k_hi) bsum += pm;
                } else {
                    orow[k] = pk;
                    orow[M - k] = pm;
                }
            }
            wave_lds_fence();
        }
        {   // k = M/2: lane 0, block c = R/2
            const float zx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, SG_Z(R / 2).x), 0));
            const float zy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, SG_Z(R / 2).y), 0));
            float pq = fmaf(zx, zx, zy * zy) * 4.0f;
            if (MODE == 1) pq = sqrtf(pq);
            if (MODE == 2) {
                if (lane == 0 && M / 2 >= p.k_lo && M / 2 <= p.k_hi) bsum += pq;
                bsum = wave_sum(bsum);
                if (lane == 0) orow[0] = bsum;
            } else {
                orow[M / 2] = pq;                            // wave-uniform store
            }
        }
#undef SG_Z
    };

    if constexpr (H == 0) {
        // Every frame reloads its samples.  T = 2 fetches the samples of frame g+1 at the top of frame g (+32 VGPRs throughout);
        // T = 4 has no registers to spare and loads at the top of the frame (fetching late, as the split pass frees registers,
        // or whole frames ahead at 2 waves/SIMD, measured slower: 1.83-2.26 ms against 1.78 ms per 64-clip batch at hop 64).
        constexpr bool kPrefetch = T == 2;
        auto load_frame = [&](int clip, int f, float2 (&dst)[T][8]) {
            const float* const src = p.x + static_cast<int64_t>(clip) * p.clip_stride + static_cast<int64_t>(f) * p.hop + 2 * lane;
#pragma unroll
            for (int a0 = 0; a0 < T; ++a0)
#pragma unroll
                for (int a1 = 0; a1 < 8; ++a1) dst[a0][a1] = *reinterpret_cast<const float2*>(src + 128 * (a0 + T * a1));
        };
        // (clip, frame) of the run's first frame by one division; after that they advance incrementally
        int clip = static_cast<int>(g / p.n_frames);
        int f = static_cast<int>(g - static_cast<int64_t>(clip) * p.n_frames);
        float2 nxt[kPrefetch ? T : 1][8];
        if (kPrefetch && g < g_end) load_frame(clip, f, reinterpret_cast<float2 (&)[T][8]>(nxt));

        for (; g < g_end; ++g) {
            float* const orow = p.out + static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(f) * (MODE == 2 ? 1 : NB);
            const int clip_n = f + 1 == p.n_frames ? clip + 1 : clip, f_n = f + 1 == p.n_frames ? 0 : f + 1;

            float2 d[T][8];
            if (kPrefetch) {
#pragma unroll
                for (int a0 = 0; a0 < T; ++a0)
#pragma unroll
                    for (int a1 = 0; a1 < 8; ++a1) d[a0][a1] = nxt[kPrefetch ? a0 : 0][a1];
                // the run's last frame fetches itself again: an unconditional fetch keeps the old registers out of the loop's live set
                const bool more = g + 1 < g_end;
                load_frame(more ? clip_n : clip, more ? f_n : f, reinterpret_cast<float2 (&)[T][8]>(nxt));
            } else {
                load_frame(clip, f, d);
            }
            frame(d, orow);
            clip = clip_n;
            f = f_n;
        }
    } else {
        // The frame's samples live in raw[][] across frames; the next frame shifts them by H blocks of 128 samples (register moves)
        // and loads the H new blocks -- one or two 8-byte loads per lane and frame instead of 8T from L2 (measured on r8x3: a
        // reloaded sample costs about what an HBM byte costs).  T = 4 pays 64 more VGPRs for it: 2 waves per SIMD (OccFor).
        // blocks k >= first of the frame that starts `off` samples into the clip at xclip (both wave-uniform: the per-lane part of
        // the address is the 32-bit 2*lane only)
        auto load_blocks = [&](const float* xclip, int64_t off, float2 (&dst)[T][8], int first) {
            const float* const src = xclip + off + 2 * lane;
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (k >= first) dst[k % T][k / T] = *reinterpret_cast<const float2*>(src + 128 * k);
        };
        while (g < g_end) {                                  // one trip per stretch: frames fs, fs + sub, ... (`count` of them) of one clip
            const int clip = static_cast<int>(g / p.n_frames);
            const int f0 = static_cast<int>(g - static_cast<int64_t>(clip) * p.n_frames);
            const int f1 = static_cast<int>(min(static_cast<int64_t>(p.n_frames), f0 + (g_end - g)));
            int fs = f0, count = f1 - f0;
            if (p.sub > 1) {                                 // sequence v holds ceil((n_frames - v) / sub) frames (stft_r8x3.hip)
                int v = 0, first = 0, len = (p.n_frames + p.sub - 1) / p.sub;
                while (f0 >= first + len) { first += len; ++v; len = (p.n_frames - v + p.sub - 1) / p.sub; }
                fs = (f0 - first) * p.sub + v;
                count = min(f1 - f0, first + len - f0);
            }
            g += count;
            const float* const xclip = p.x + static_cast<int64_t>(clip) * p.clip_stride;
            int64_t soff = static_cast<int64_t>(fs) * p.hop;
            float* orow = p.out + static_cast<int64_t>(clip) * p.out_clip_stride + static_cast<int64_t>(fs) * (MODE == 2 ? 1 : NB);
            const int row_step = (MODE == 2 ? 1 : NB) * p.sub, src_step = p.hop * p.sub;
            float2 raw[T][8];
            load_blocks(xclip, soff, raw, 0);
            for (int i = 0; i < count; ++i) {
                float2 d[T][8];
#pragma unroll
                for (int a0 = 0; a0 < T; ++a0)
#pragma unroll
                    for (int a1 = 0; a1 < 8; ++a1) d[a0][a1] = raw[a0][a1];
                soff += src_step;
                if (i + 1 < count) {                         // wave-uniform
#pragma unroll
                    for (int k = 0; k + H < R; ++k) raw[k % T][k / T] = raw[(k + H) % T][(k + H) / T];
                    load_blocks(xclip, soff, raw, R - H);
                }
                frame(d, orow);
                orow += row_step;
            }
        }
    }
}

template <int T, bool DETREND, int H>
int launch_tdh(const BigParams& prm, hipStream_t s, int mode, bool band, int n_cu) {
    constexpr int R = 8 * T, M = 64 * R, kWaves = WavesFor<T, H>::value;
    auto k0 = stft_rbig_kernel<T, DETREND, 0, H>;
    auto k1 = stft_rbig_kernel<T, DETREND, 1, H>;
    auto k2 = stft_rbig_kernel<T, DETREND, 2, H>;
    auto kern = band ? k2 : mode == SG_MODE_PSD ? k0 : k1;
    const size_t lds = (static_cast<size_t>(M) + ((SG_RBIG_B128 ? R : R - 1) + 7 + R / 2) * 64 + static_cast<size_t>(kWaves) * kSlabElems) * sizeof(float2);
    const int wg_per_cu = static_cast<int>((160 * 1024) / lds) < 1 ? 1 : static_cast<int>((160 * 1024) / lds);
    BigParams p = prm;
    int64_t n_waves = static_cast<int64_t>(n_cu) * wg_per_cu * kWaves;
    if (n_waves > p.total_frames) n_waves = p.total_frames;
    p.n_waves = static_cast<int>(n_waves);
    const int n_wg = static_cast<int>((n_waves + kWaves - 1) / kWaves);
    if (lds > 64 * 1024)
        SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * kWaves), lds, s, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SG_OK : hip_fail(e, "stft_rbig launch");
}

template <int T, bool DETREND>
int launch_td(const BigParams& prm, hipStream_t s, int mode, bool band, int n_cu, int h) {
    switch (h) {
        case 1: return launch_tdh<T, DETREND, 1>(prm, s, mode, band, n_cu);
        case 2: return launch_tdh<T, DETREND, 2>(prm, s, mode, band, n_cu);
        default: return launch_tdh<T, DETREND, 0>(prm, s, mode, band, n_cu);
    }
}

template <int T>
int launch_t(const sg_plan& p, const StftArgs& a) {
    BigParams prm{};
    prm.x = static_cast<const float*>(a.x);
    prm.clip_stride = a.clip_stride;
    prm.n_frames = static_cast<int>(a.n_frames);
    prm.hop = p.hop;
    prm.sub = 1;
    prm.total_frames = a.n_frames * a.n_clips;
    prm.out = static_cast<float*>(a.out);
    prm.out_clip_stride = a.out_clip_stride;
    prm.win2 = static_cast<const float2*>(p.win_dev);
    prm.tw = static_cast<const float2*>(p.r8_tw_dev);
    prm.scale = static_cast<float>(p.scale);
    prm.k_lo = a.k_lo; prm.k_hi = a.k_hi;
    // sliding window: hops 128 and 256 directly, hops 64 / 32 / 16 as 2 / 4 / 8 interleaved hop-128 sequences
    int h = 0;
    const char* off = SG_TUNE_ENV("SPECTRO_RBIG_NO_SLIDE");      // A/B aid: "1" = never slide, "4" = not for T = 4
    if (!(off && (off[0] == '1' || (off[0] == '4' && T == 4)))) {
        if (p.hop == 128 || p.hop == 256) h = p.hop / 128;
        else if (p.hop == 64 || p.hop == 32 || p.hop == 16) { h = 1; prm.sub = 128 / p.hop; }
    }
    const bool band = a.band_mode != 0;                    // run_stft has checked: psd plan, 0 <= k_lo <= k_hi < n_bins
    // T = 4 slides at 2 waves/SIMD (64 more VGPRs): -13...16 % with the rows written.  For the band sums alone (no stores to hide
    // behind) sliding lost by 4 % in round 2 (1.34 against 1.29 ms per 64-clip batch at hop 64); since the table reads went to
    // ds_read_b128 and the pass-2 twiddles to registers it wins: 1.185 / 0.611 / 0.320 against 1.277 / 0.651 / 0.335 ms at hops
    // 64 / 128 / 256 (profiles/r03_rbig_band_slide.txt).  SPECTRO_RBIG_BAND_RELOAD=1 restores the reloading form for an A/B.
    if (T == 4 && band && SG_TUNE_ENV("SPECTRO_RBIG_BAND_RELOAD")) { h = 0; prm.sub = 1; }
    return p.detrend == SG_DETREND_CONSTANT ? launch_td<T, true>(prm, a.stream, p.mode, band, p.n_cu, h)
                                            : launch_td<T, false>(prm, a.stream, p.mode, band, p.n_cu, h);
}

}  // namespace

bool rbig_can_run(const sg_plan& p, const StftArgs& a) {
    return !a.in_i16 && (p.hop % 2 == 0) && (a.clip_stride % 2 == 0 || a.n_clips == 1) &&
           (reinterpret_cast<uintptr_t>(a.x) % 8 == 0) && a.n_frames <= INT32_MAX;
}

int launch_rbig(const sg_plan& p, const StftArgs& a) {
    if (a.n_frames <= 0 || a.n_clips <= 0) return SG_OK;
    return p.nfft == 2048 ? launch_t<2>(p, a) : launch_t<4>(p, a);
}

// Per-lane twiddle table [(R-1) + 7 + R/2][64] float2 (R = nfft/128), computed in double:
//   t1[r-1][j] = exp(-2*pi*i*j*r/M), r = 1..R-1;  t2[s-1][j] = exp(-2*pi*i*(j&7)*s/64);  t3[c][j] = (cos, sin)(2*pi*(j+64c)/(2M))
int build_rbig_tables(sg_plan& p) {
    const int R = p.nfft / 128, M = 64 * R;
    std::vector<float2> tw(static_cast<size_t>(R - 1 + 7 + R / 2) * 64);
    const double two_pi = 6.283185307179586476925286766559;
    for (int j = 0; j < 64; ++j) {
        for (int r = 1; r < R; ++r) {
            const double ang = -two_pi * static_cast<double>((static_cast<long long>(j) * r) % M) / M;
            tw[(r - 1) * 64 + j] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
        }
        for (int s = 1; s < 8; ++s) {
            const double ang = -two_pi * static_cast<double>(((j & 7) * s) % 64) / 64.0;
            tw[(R - 1 + s - 1) * 64 + j] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
        }
        for (int c = 0; c < R / 2; ++c) {
            const double ang = two_pi * static_cast<double>(j + 64 * c) / (2.0 * M);
            tw[(R - 1 + 7 + c) * 64 + j] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
        }
    }
    SG_HIP(hipMalloc(&p.r8_tw_dev, tw.size() * sizeof(float2)));
    SG_HIP(hipMemcpy(p.r8_tw_dev, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    return SG_OK;
}

}  // namespace sg
