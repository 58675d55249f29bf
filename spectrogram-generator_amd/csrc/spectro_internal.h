// Internal declarations shared by the translation units of libspectro.so.
// Not part of the ABI (that is include/spectro.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <mutex>
#include <vector>
#include "spectro.h"
#include "host_shim.h"

namespace sg {

int hip_fail(hipError_t e, const char* what);   // records message, returns SG_ERR_HIP

#define SG_HIP(call)                                                   \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) return ::sg::hip_fail(e__, #call);      \
    } while (0)

// Tuning aids (SPECTRO_R8_OCC, SPECTRO_RBIG_NO_SLIDE ...: environment variables the A/B tools under tools/ flip between launches) exist only
// in a -DSG_TUNING=1 build (`SG_TUNING=1 python build.py --force`, or tools/build_variant.sh <name> <file> -DSG_TUNING=1).  The product
// library reads no environment variable on a launch path: getenv is not safe against a concurrent setenv, and the Python shim calls in
// with the GIL released while another thread may be writing os.environ; a stray variable cannot change a release build's grid either.
#ifdef SG_TUNING
#include <cstdlib>
#define SG_TUNE_ENV(name) getenv(name)
#else
#define SG_TUNE_ENV(name) (static_cast<const char*>(nullptr))
#endif

enum class Kernel { R8X3, R8X3D, RSMALL, RBIG, RBIGD, STOCKHAM, BLUESTEIN, RBLUE, RBLUED, RBLUEW, RBLUEWD, RTINY };

}  // namespace sg

// The plan: argument triage done once, device tables owned here.
struct sg_plan {
    int nperseg = 0, nfft = 0, hop = 0;
    int detrend = 0, scaling = 0, mode = 0, dtype = 0;
    double fs = 1.0;
    double scale = 1.0;          // 1/(fs*sum w^2) or 1/(sum w)^2, in dtype precision (scipy:2086-2089)
    sg::Kernel kernel = sg::Kernel::STOCKHAM;
    int device = 0;
    int n_cu = 256;
    // device tables
    void* win_dev = nullptr;     // nperseg reals of dtype
    void* tw_dev = nullptr;      // nfft/2 complex of dtype: exp(-2*pi*i*k/nfft), k < nfft/2
    void* r8_tw_dev = nullptr;   // R8X3: [18][64] (R8X3D: double2), RSMALL: [(R-1)+11][64] float2, RBIG / RBIGD: [(R-1)+7+R/2][64] per-lane twiddles
    void* r8_win_dev = nullptr;  // R8X3: the window times sqrt(scale / 2) (psd) or sqrt(scale / 4) (magnitude), so that a wave's prologue is loads only
    // Bluestein tables (complex of dtype)
    int bs_len = 0;              // padded pow2 length L >= 2*nfft-1
    void* bs_chirp_dev = nullptr;   // b[n] = exp(-i*pi*n^2/nfft), n < nfft
    void* bs_filter_dev = nullptr;  // FFT_L of conj-chirp filter, bit-reversed order, pre-scaled by 1/L
    void* bs_tw_dev = nullptr;      // exp(-2*pi*i*k/L), k < L/2
    // register chirp-z tables (RBLUE, stft_rblue.hip): window pairs + chirp, filter spectrum, split twiddles, per-lane FFT twiddles
    void* rb_wc_dev = nullptr;      // (RBLUED, stft_rblue_f64.hip: ONE table of (re, im) double pairs holds all of them; RBLUEW, stft_rbluew.hip: one float2 table)
    void* rb_filt_dev = nullptr;
    void* rb_stw_dev = nullptr;
    void* rb_tw_dev = nullptr;
};

namespace sg {

// launchers (each returns an sg_status) ---------------------------------------
struct StftArgs {
    const void* x; int in_i16;              // input pointer; 1 = int16 samples
    int64_t n_samples, clip_stride; int n_clips;
    void* out; int64_t out_clip_stride;     // spectrum out (may be null for band mode)
    int64_t n_frames;
    int band_mode, k_lo, k_hi;              // band-power variant: only per-frame sums are written
    int db_mode; float inv_base; void* mm_parts;   // r8x3 dB-image variant: bins [k_lo,k_hi] as 10*log10(clip(S*inv_base,0,1)+1e-12),
                                                   // per-wave (min, max) pairs into mm_parts[r8x3_grid_waves]
    int mel_ipl;                                   // r8x3 mel variant: work items per lane (1..4) of a band-sparse bank, 0 = off
    const int* mel_start; const float* mel_w; const int* mel_first; const int* mel_count; int n_mels, log_scale;
    hipStream_t stream;
};

int launch_r8x3(const sg_plan& p, const StftArgs& a);
int launch_r8x3_f64(const sg_plan& p, const StftArgs& a);
bool r8x3_f64_can_run(const sg_plan& p, const StftArgs& a);
int r8x3_grid_waves(const sg_plan& p, int64_t total_frames, bool mel = false);   // waves (= min/max partials of the dB variant) of a launch
// epilogue.hip: per-(device, stream) scratch of 256 KiB for reduction partials; fold of n (min, max) float pairs into mm[2]
void* reduction_scratch(hipStream_t s);
// Sequences of launches that hand data to each other through reduction_scratch / stream_workspace (partials -> fold, convert ->
// transform) hold the stream's lock while they are being SUBMITTED, so that two host threads using the same stream cannot interleave
// their launches (the stream then runs each sequence back to back).  Recursive: a sequence may contain another.
std::recursive_mutex& launch_sequence_mutex(hipStream_t s);   // one per (device, stream): other streams never wait
void drop_stream_state(hipStream_t s);                 // sg_stream_destroy: frees the stream's scratch + workspace, forgets its lock
void* stream_workspace(hipStream_t s, size_t bytes);   // grows on demand, per (device, stream); nullptr when out of memory
int device_cu_count();   // compute units of the current device, queried once per device (hipGetDeviceProperties costs 10s-100s of us)
int fold_minmax_f32(const float* parts, int n_parts, float* mm_dev, hipStream_t s);
int launch_rsmall(const sg_plan& p, const StftArgs& a);
bool rsmall_can_run(const sg_plan& p, const StftArgs& a);
int launch_rbig(const sg_plan& p, const StftArgs& a);
bool rbig_can_run(const sg_plan& p, const StftArgs& a);
int launch_rbig_f64(const sg_plan& p, const StftArgs& a);
bool rbig_f64_can_run(const sg_plan& p, const StftArgs& a);
int launch_stockham(const sg_plan& p, const StftArgs& a);
int launch_bluestein(const sg_plan& p, const StftArgs& a);
int launch_rblue(const sg_plan& p, const StftArgs& a);
bool rblue_can_run(const sg_plan& p, const StftArgs& a);
int launch_rblue_f64(const sg_plan& p, const StftArgs& a);
int launch_rbluew(const sg_plan& p, const StftArgs& a);
bool rbluew_can_run(const sg_plan& p, const StftArgs& a);
int rbluew_size(int nfft);
int launch_rbluew_f64(const sg_plan& p, const StftArgs& a);
int launch_rtiny(const sg_plan& p, const StftArgs& a);
bool rtiny_can_run(const sg_plan& p, const StftArgs& a);
bool rbluew_f64_can_run(const sg_plan& p, const StftArgs& a);
int rbluew_f64_size(int nfft);
bool rblue_f64_can_run(const sg_plan& p, const StftArgs& a);

int build_r8x3_tables(sg_plan& p, const std::vector<double>& window);
int build_r8x3_f64_tables(sg_plan& p);
int build_rsmall_tables(sg_plan& p);
int build_rbig_tables(sg_plan& p);
int build_rbig_f64_tables(sg_plan& p);
int build_bluestein_tables(sg_plan& p);
int build_rblue_tables(sg_plan& p, const std::vector<double>& window);
int build_rblue_f64_tables(sg_plan& p, const std::vector<double>& window);
int build_rbluew_tables(sg_plan& p, const std::vector<double>& window);
int build_rbluew_f64_tables(sg_plan& p, const std::vector<double>& window);
int build_rtiny_tables(sg_plan& p);
void host_fft_pow2(std::vector<double>& re, std::vector<double>& im);     // in place, forward, radix 2 (stft_rblue.hip)

}  // namespace sg
