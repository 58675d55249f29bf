/*
 * spectro.h -- C ABI of libspectro.so, the MI355X (gfx950) STFT/PSD engine.
 *
 * This is the drop-in boundary for ONE hot path of the reference application:
 * the call
 *
 *     f, t, Sxx = spectrogram(data, fs=fs, nperseg=nperseg,
 *                             scaling="density", mode="psd")
 *
 * made at /root/reference PlotEngine.py:113 and PlotEngine.py:232
 * (`spectrogram` = scipy.signal.spectrogram, PlotEngine.py:8), plus the numpy
 * post-processing that consumes its result (PlotEngine.py:114-131, 238-241,
 * 686-719).  The reference is pure Python, so a maintainer binds this library
 * with ctypes; nothing here mentions torch or numpy.
 *
 * Conventions
 *   - every entry point returns 0 on success or a negative sg_status;
 *     sg_last_error() returns a thread-local message for the last failure.
 *   - the caller owns every buffer; a plan owns only its device tables
 *     (window, twiddles).  Pointers named *_dev are device pointers of the
 *     current HIP device, `stream` is a hipStream_t passed as void* (NULL =
 *     the default stream).  No entry point synchronises unless it says so; one exception by construction: a call that needs a
 *     LARGER stream workspace than that stream has so far (int16 batches on nfft 256 / 512 / 2048 / 4096 plans, chirp-z sizes whose
 *     convolution buffer does not fit the LDS) synchronises THAT stream once while the block is replaced -- later calls of the
 *     same or a smaller size are asynchronous again.
 *   - spectra are FRAME-MAJOR on the device: out[clip][frame][bin]; the Python
 *     shim returns the transposed view [bin][frame] exactly like
 *     scipy/signal/_spectral_py.py:2153 (moveaxis of a frame-major result).
 *   - re-entrant per (plan, stream).  Besides the thread-local error string the library keeps per-(device, stream) scratch (a small
 *     reduction buffer, a workspace that grows on demand: sg_workspace_release) that the calls of a stream share in stream order;
 *     entry points that hand data from one launch to the next through it (min/max and band totals: partials -> fold; sg_stft_db;
 *     int16 batches: convert -> transform) submit their launches under that stream's lock, so host threads may share a stream;
 *     calls on different streams or devices never wait for each other on the host.
 */
#ifndef SPECTRO_H
#define SPECTRO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_VERSION 103 /* 0.1.1: sg_stft_mel takes weights_n_bins; sg_stft_db, sg_db_rescale, sg_colormap_db,
                          * sg_band_features_batch, sg_device_pci_bus_id added; 0.1.2: sg_convert_i16;
                          * 0.1.3: "rblue" / "rbluew" / "rtiny" plans (every nperseg the reference's spin box produces runs a register kernel), sg_stft_mel's log_scale is a flags word (SG_MEL_*),
                          * no environment variable is read on a launch path */

typedef enum sg_status {
    SG_OK = 0,
    SG_ERR_ARG = -1,         /* bad argument (shim raises ValueError)            */
    SG_ERR_HIP = -2,         /* HIP runtime failure (shim raises RuntimeError)   */
    SG_ERR_UNSUPPORTED = -3, /* valid request the device path does not cover     */
    SG_ERR_NO_DEVICE = -4
} sg_status;

/* scipy's `detrend=` (scipy/signal/_spectral_py.py:2070-2072) */
typedef enum sg_detrend { SG_DETREND_NONE = 0, SG_DETREND_CONSTANT = 1, SG_DETREND_LINEAR = 2 } sg_detrend;
/* scipy's `scaling=` (scipy:2086-2089) */
typedef enum sg_scaling { SG_SCALING_DENSITY = 0, SG_SCALING_SPECTRUM = 1 } sg_scaling;
/* scipy's `mode=` (scipy:960-985). PSD/MAGNITUDE/ANGLE write one real per bin,
 * COMPLEX writes interleaved (re, im). 'phase' = unwrap(ANGLE) is done by the caller. */
typedef enum sg_mode { SG_MODE_PSD = 0, SG_MODE_MAGNITUDE = 1, SG_MODE_COMPLEX = 2, SG_MODE_ANGLE = 3 } sg_mode;
/* arithmetic type of the path = input precision (scipy:1976-1981: f32 in -> f32 out) */
typedef enum sg_dtype { SG_F32 = 0, SG_F64 = 1 } sg_dtype;

typedef struct sg_plan sg_plan;

/* ---- library / device ------------------------------------------------- */
int sg_version(void);
const char* sg_last_error(void);
int sg_device_count(int* count);
/* Select `device` for the calling thread (hipSetDevice) and check it is gfx950. */
int sg_init(int device);
/* name (e.g. "gfx950"), CU count and HBM bytes of the current device */
int sg_device_info(char* arch, size_t arch_len, int* compute_units, uint64_t* hbm_bytes);
/* free and total device memory of the current device right now (hipMemGetInfo): what a caller sizes its batches against */
int sg_mem_info(uint64_t* free_bytes, uint64_t* total_bytes);
/* PCI address of the current device as sysfs spells it ("0000:05:00.0"): /sys/bus/pci/devices/<id>/hwmon/ holds its
 * clock and board-power sensors, which bench.py reads beside the timing (the headline kernel runs at the power cap). */
int sg_device_pci_bus_id(char* buf, size_t len /* >= 13 */);

/* ---- raw device memory / streams (so a ctypes host needs nothing else) -- */
int sg_malloc(void** dev_ptr, size_t bytes);
int sg_free(void* dev_ptr);
int sg_host_alloc(void** host_ptr, size_t bytes); /* pinned */
int sg_host_free(void* host_ptr);
/* Pin an existing host range in place (hipHostRegister) so that copies from/to it run at PCIe speed and truly
 * asynchronously; undo with sg_host_unregister.  Used by the shim for large numpy buffers. */
int sg_host_register(void* host_ptr, size_t bytes);
int sg_host_unregister(void* host_ptr);
int sg_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int sg_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int sg_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream); /* ranges may not overlap */
/* `height` rows of `width_bytes`, row pitches in bytes (streaming: one call moves a [channels][n] chunk into / out of the
 * per-channel device buffers); kind: 0 host->device, 1 device->host, 2 device->device */
int sg_memcpy2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t height,
                int kind, void* stream);
int sg_memset(void* dst_dev, int value, size_t bytes, void* stream);
int sg_stream_create(void** stream);
int sg_stream_destroy(void* stream); /* synchronises the stream, frees the scratch / workspace the library kept for it */
int sg_stream_sync(void* stream); /* blocks the caller */

/* ---- plan ------------------------------------------------------------- */
/*
 * Replaces the argument triage of scipy.signal.spectrogram /_spectral_helper
 * (scipy:967-970, 2031-2041, 2083-2091) for one (nperseg, nfft, hop, window).
 *   nperseg  samples per frame (>= 1); nfft >= nperseg (zero padded); 1 <= hop <= nperseg
 *            (hop = nperseg - noverlap; the reference's default is nperseg - nperseg/8)
 *   window   nperseg doubles on the HOST (the shim builds scipy's periodic Tukey(0.25)
 *            for the reference call); cast to `dtype` like scipy:2083-2084
 *   fs       sampling rate; scale = 1/(fs*sum(w^2)) or 1/sum(w)^2 (scipy:2086-2089),
 *            computed in `dtype` precision like scipy does
 * Supported on the device: every nfft from 2 to 2^20 in SG_F32 and SG_F64 -- register kernels for f32 nfft 256..4096, an LDS
 * Stockham kernel for the other powers of two up to 16384 (f32) / 8192 (f64), chirp-z (Bluestein) for everything else, in
 * LDS while the convolution buffer fits and in the stream's HBM workspace (see sg_workspace_release) above that (e.g. f64 nperseg 4097..8191,
 * which the reference GUI's spin box reaches).
 */
int sg_plan_create(sg_plan** plan, int nperseg, int nfft, int hop, const double* window,
                   int detrend, double fs, int scaling, int mode, int dtype);
int sg_plan_destroy(sg_plan* plan);
/* A2: (n_samples - nperseg)/hop + 1, 0 when n_samples < nperseg (scipy:2180-2188) */
int sg_plan_n_frames(const sg_plan* plan, int64_t n_samples, int64_t* n_frames);
/* nfft/2 + 1 */
int sg_plan_n_bins(const sg_plan* plan, int* n_bins);
/* the scale factor the kernels multiply |X|^2 with (as double) */
int sg_plan_scale(const sg_plan* plan, double* scale);
/* name of the kernel family the plan dispatches to: "r8x3", "r8x3d" / "rsmalld" (f64 nperseg = nfft = 1024 / 128, 256, 512), "rsmall" (128, 256, 512),
 * "rbig", "rbigd", "rtiny" / "rtinyd" (32, 64, 96, 160, 192, 224), "rblue" / "rblued" (f32 / f64, even nperseg = nfft <= 2048 / 1024 that is no power of two: register
 * chirp-z), "rbluew" / "rbluewd" (the same up to 8192, nperseg a multiple of 4 / 8 / 16: two to eight wavefronts per frame; 8192 itself too),
 * "stockham", "bluestein" */
const char* sg_plan_kernel(const sg_plan* plan);
/* Tests / benchmarks: route the plan to another family that can run it ("stockham" for an
 * r8x3 plan).  SG_ERR_UNSUPPORTED if that family cannot run this plan. */
int sg_plan_force_kernel(sg_plan* plan, const char* name);
/* A7 on the host, bit-exact with scipy:2115 and scipy:2136-2137 */
int sg_freqs(int nfft, double fs, double* f_out /* nfft/2+1 */);
int sg_times(int64_t n_samples, int nperseg, int hop, double fs, double* t_out /* n_frames */);

/* ---- the hot path (A2..A6) -------------------------------------------- */
/*
 * Framing + detrend + window + real FFT + |X|^2*scale with one-sided doubling for
 * n_clips signals of n_samples each (clip c starts at x_dev + c*clip_stride elements).
 * out_dev receives [n_clips][n_frames][n_bins] (x2 for SG_MODE_COMPLEX) elements of
 * the plan's dtype; clip c starts at out_dev + c*out_clip_stride elements
 * (out_clip_stride >= n_frames*n_bins*(complex?2:1)).  Asynchronous on `stream`.
 */
int sg_stft(const sg_plan* plan, const void* x_dev, int64_t n_samples, int64_t clip_stride,
            int n_clips, void* out_dev, int64_t out_clip_stride, void* stream);
/* int16 PCM input converted on the fly (WAV path); plan dtype must be SG_F32.
 * Samples are used as-is (no 1/32768 scaling), i.e. like numpy's int16 -> float cast. */
int sg_stft_i16(const sg_plan* plan, const int16_t* x_dev, int64_t n_samples, int64_t clip_stride,
                int n_clips, float* out_dev, int64_t out_clip_stride, void* stream);
/* Frees what the library keeps per (device, stream): the workspaces of int16 batches (float copy) and oversize chirp-z plans (they
 * grow on demand) and the small reduction scratch.  Streams made by sg_stream_create lose theirs in sg_stream_destroy; for the
 * default stream and for streams the caller owns this is the call (otherwise held until the library is unloaded).  Waits for the
 * devices concerned; nothing may be in flight on other host threads. */
int sg_workspace_release(void);
/* dst[i] = (float)src[i], i < n: int16 PCM to the f32 the plans compute in (exact).  sg_stft_i16 does this itself where a
 * kernel has no int16 loads of its own (batches on nperseg 256 / 512 / 2048 / 4096: the stream's workspace, then the
 * float kernel); exported for callers that keep a converted copy across many calls (spectro.engine.DeviceClips). */
int sg_convert_i16(const int16_t* src_dev, float* dst_dev, int64_t n, void* stream);
/*
 * Same framing/FFT but the spectrum never reaches HBM: per frame only
 * p[frame] = sum_{k in [k_lo, k_hi]} Sxx[k, frame] is written (A11 band sum,
 * PlotEngine.py:238-239).  band_out_dev: [n_clips][n_frames] of the plan's dtype.
 * Needs a psd plan; fused in every register family (f32 and f64, every power of two 32...4096 and the chirp-z families) and in the LDS kernel;
 * SG_ERR_UNSUPPORTED on an LDS chirp-z ("bluestein") plan (use sg_stft + sg_band_sum there).
 */
int sg_stft_band_power(const sg_plan* plan, const void* x_dev, int64_t n_samples,
                       int64_t clip_stride, int n_clips, int k_lo, int k_hi,
                       void* band_out_dev, int64_t out_clip_stride, void* stream);

/*
 * The log display of PlotEngine.py:126-131 fused into the STFT kernel for a caller-supplied global_max (the batch-global
 * normalisation base of PlotEngine.py:110,126): for every frame only
 *     db[frame][k - k_lo] = 10*log10(clip(Sxx[k]/(global_max + 1e-20), 0, 1) + 1e-12),   k_lo <= k <= k_hi,
 * is written (db_dev: [n_clips][n_frames][k_hi-k_lo+1] f32, clip c at db_dev + c*out_clip_stride) and mm_dev[0..1]
 * receives the minimum and maximum of the dB values over the whole call (per-wave partials folded by one small kernel on
 * the same stream), so the min-max rescale of :130-131 costs no further pass over the spectrum: fold it into the consumer
 * (sg_colormap_db) or apply it in place (sg_db_rescale).  Needs an f32 nperseg = nfft = 1024 PSD plan
 * (sg_plan_kernel == "r8x3") and global_max > 0; SG_ERR_UNSUPPORTED / SG_ERR_ARG otherwise (other plans:
 * sg_stft + sg_normalise_image).  Asynchronous.
 */
int sg_stft_db(const sg_plan* plan, const float* x_dev, int64_t n_samples, int64_t clip_stride, int n_clips,
               int k_lo, int k_hi, double global_max, float* db_dev, int64_t out_clip_stride, float* mm_dev,
               void* stream);
/* db <- (db - mm[0]) / (mm[1] - mm[0]) in place, zeros when mm[1] - mm[0] <= 1e-6 (PlotEngine.py:130-131); n elements. */
int sg_db_rescale(float* db_dev, int64_t n, const float* mm_dev, void* stream);

/* ---- epilogues over a frame-major spectrum (A8..A13) ------------------ */
/* `dtype` (sg_dtype) is the element type of every spectrum / image / band buffer below.
 * The reductions (sg_minmax, sg_normalise_image, sg_band_totals) run in two stages through a 256 KiB scratch that the
 * library keeps per (device, stream) from its first use until it is unloaded; calls on one stream are ordered, calls on
 * different streams do not share it. */
/* mm_dev[0] = min, mm_dev[1] = max over rows [0,n_frames) x bins [k_lo,k_hi] of spec (row stride n_bins).
 * mm_dev: 2 elements of dtype. */
int sg_minmax(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int k_lo, int k_hi,
              void* mm_dev, void* stream);
/*
 * A9/A10 (PlotEngine.py:126-131): img = clip(S/(base+1e-20),0,1); if log_scale:
 * db = 10*log10(img+1e-12), img = (db-db_min)/(db_max-db_min) or 0 when range <= 1e-6.
 * base = global_max if > 0 else max(S) over the band; db_min/db_max follow from the
 * band's min/max because the map is monotone.  Writes img_dev[n_frames][k_hi-k_lo+1].
 * mm_dev: scratch of 2 elements.  Asynchronous.
 */
int sg_normalise_image(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int k_lo, int k_hi,
                       int log_scale, double global_max, void* img_dev, void* mm_dev, void* stream);
/* A11 (PlotEngine.py:239-241): feat[f] = (log10(p[f]+1e-20), diff with prepend) from band sums p. */
int sg_band_features(const void* band_dev, int dtype, int64_t n_frames, void* feat_dev /* [n_frames][2] */,
                     void* stream);
/* The same for n_clips clips of n_frames each in one launch (band_dev [n_clips][n_frames], feat_dev [n_clips][n_frames][2]):
 * the difference restarts at every clip. */
int sg_band_features_batch(const void* band_dev, int dtype, int n_clips, int64_t n_frames, void* feat_dev, void* stream);
/* A11 without the fused kernel: p[f] = sum_{k_lo..k_hi} spec[f][k] */
int sg_band_sum(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int k_lo, int k_hi,
                void* band_dev, void* stream);
/* A12/A13 (PlotEngine.py:686-719): sums_dev[b] = sum over frames of sum_{k in [lo_b, hi_b)} max(0, spec[f][k])
 * for n_bands (<= 16) half-open bin ranges given on the host; accumulates in double. */
int sg_band_totals(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int n_bands,
                   const int* k_lo_host, const int* k_hi_host, double* sums_dev, void* stream);
/* Copy bins [k_lo,k_hi] of every frame into dst_dev[n_frames][k_hi-k_lo+1] (A8 mask + store). */
int sg_slice_bins(const void* spec_dev, int dtype, int64_t n_frames, int n_bins, int k_lo, int k_hi,
                  void* dst_dev, void* stream);

/* ---- display epilogue: normalised image -> RGBA (replaces the colour mapping matplotlib's pcolormesh(cmap='jet',
 *      vmin=0, vmax=1) does on the host at PlotEngine.py:134-135) ------------------------------------------------ */
/* 256-entry 'jet' table as matplotlib builds it (segment data sampled at i/255), rgba_host[256][4] in 0..255. */
int sg_jet_lut(uint8_t* rgba_host);
/* rgba_dev[i] = lut_dev[clamp(int(img[i]*256), 0, 255)] (NaN -> transparent black), img f32 in [0,1]; n elements. */
int sg_colormap(const float* img_dev, int64_t n, const uint8_t* lut_dev /* [256][4] */, uint8_t* rgba_dev, void* stream);

/* sg_colormap of the min-max rescaled dB image without materialising it: v = (db - mm[0])/(mm[1] - mm[0]) (0 when the
 * range is <= 1e-6), rgba = lut[clamp(int(v*256), 0, 255)]. */
int sg_colormap_db(const float* db_dev, int64_t n, const float* mm_dev, const uint8_t* lut_dev, uint8_t* rgba_dev,
                   void* stream);

/* ---- mel filterbank (BASELINE cfg3; NOT in the reference: definition is this library's own) ---- */
/*
 * HTK mel scale m = 2595*log10(1 + f/700), n_mels triangular filters with peak 1 (no area normalisation)
 * between fmin and fmax, evaluated at the nfft/2+1 bin frequencies k*fs/nfft.  Writes the dense weight
 * matrix weights_host[n_bins][n_mels] (row-major doubles); the caller uploads it in the plan's dtype.
 */
int sg_mel_weights(int nfft, double fs, int n_mels, double fmin, double fmax, double* weights_host);
/* Per tile of 16 mel bands t: the bin range [k_lo[t], k_hi[t]) (multiples of 4) outside which every weight of the
 * tile is zero (block sparsity of the triangular bank).  ceil(n_mels/16) entries each. */
int sg_mel_tile_ranges(const double* weights_host, int n_bins, int n_mels, int* k_lo, int* k_hi);
/* Device layout of the bank: transposed and zero padded, packed_host[16*ceil(n_mels/16)][round_up(n_bins,16)] f32
 * (bin index contiguous), so that both MFMA operands are fetched as aligned 16-byte vectors. */
int sg_mel_pack_weights(const double* weights_host, int n_bins, int n_mels, float* packed_host);
/*
 * mel_dev[n_frames][n_mels] = spec_dev[n_frames][n_bins] x W   (weights_dev = sg_mel_pack_weights layout; f32, contraction on the
 * matrix cores: v_mfma_f32_16x16x4_f32, exact f32 FMA chain); tile_k_lo/hi from sg_mel_tile_ranges (host arrays,
 * NULL = dense); log_scale != 0 applies 10*log10(max(x, 1e-10)).  n_mels <= 128.  Asynchronous.
 */
int sg_mel(const float* spec_dev, int64_t n_frames, int n_bins, const float* weights_dev, int n_mels,
           const int* tile_k_lo, const int* tile_k_hi, int log_scale, float* mel_dev, void* stream);

/*
 * BASELINE cfg3 fused: framing .. PSD as in sg_stft, then the mel contraction as an MFMA epilogue inside the same
 * kernel; only mel_dev[n_clips][n_frames][n_mels] is written (hop*4 + n_mels*4 algorithmic bytes per frame).
 * Needs an f32 nperseg = nfft = 1024 PSD plan (sg_plan_kernel == "r8x3"), an even hop and 8-byte aligned float input;
 * SG_ERR_UNSUPPORTED otherwise (use sg_stft + sg_mel).  Weights/ranges as for sg_mel; weights_n_bins is the n_bins the
 * bank was packed for (sg_mel_pack_weights) and must equal the plan's nfft/2+1 (SG_ERR_ARG otherwise: the kernel indexes
 * the bank with the plan's row pitch).  Asynchronous.
 * log_scale is a flags word here: SG_MEL_LOG (= 1, so a plain 0 / 1 keeps its meaning) | SG_MEL_FORM_WS (the wave-specialised
 * kernel form: producer waves transform while consumer waves contract the previous tile) | SG_MEL_FORM_CONS4 (that form with four
 * consumer waves instead of eight).  The forms compute the same values; the library itself reads no environment variable.
 */
#define SG_MEL_LOG 1
#define SG_MEL_FORM_WS 0x100
#define SG_MEL_FORM_CONS4 0x200
int sg_stft_mel(const sg_plan* plan, const float* x_dev, int64_t n_samples, int64_t clip_stride, int n_clips,
                const float* packed_weights_dev, int weights_n_bins, int n_mels, const int* tile_k_lo,
                const int* tile_k_hi, int log_scale, float* mel_dev, int64_t out_clip_stride, void* stream);

/*
 * The same product through a BAND-SPARSE form of the bank (the default of the Python shim for triangular banks): a
 * triangular filterbank touches every bin with exactly two bands, so a frame's mel spectrum is ~2 n_bins multiply-adds, not
 * the n_bins * n_mels of the dense product.  sg_mel_sparse_pack (host) cuts the hull of every band's non-zero bins into work
 * items of <= 8 bins: item_start[64*IPL], item_w[8][64*IPL] (zero padded), band_first / band_count[n_mels] (a band's items are
 * adjacent), *items_per_lane = IPL in 1..4; the caller provides room for IPL = 4 (256 ints, 2048 floats) and uploads the
 * first 64*IPL resp. 8*64*IPL entries.  SG_ERR_UNSUPPORTED when the bank needs more than 256 items (a dense bank) or has more
 * than 128 bands: use sg_stft_mel.  sg_stft_mel_sparse runs the nfft-1024 register kernel with that epilogue: every wave
 * leaves the PSD row in LDS, gathers its items against weights held in registers and writes mel_dev[n_clips][n_frames][n_mels]
 * (10*log10(max(x, 1e-10)) with log_scale); no workgroup barrier, no frame tile.  Same plan / input requirements as sg_stft_mel.
 */
int sg_mel_sparse_pack(const double* weights_host, int n_bins, int n_mels, int* items_per_lane, int32_t* item_start,
                       float* item_w, int32_t* band_first, int32_t* band_count);
/* sg_mel through the band-sparse bank: mel_dev[n_frames][n_mels] from spec_dev[n_frames][n_bins] (f32), one wavefront per row. */
int sg_mel_sparse(const float* spec_dev, int64_t n_frames, int n_bins, const int32_t* item_start_dev, const float* item_w_dev,
                  const int32_t* band_first_dev, const int32_t* band_count_dev, int items_per_lane, int n_mels, int log_scale,
                  float* mel_dev, void* stream);
int sg_stft_mel_sparse(const sg_plan* plan, const float* x_dev, int64_t n_samples, int64_t clip_stride, int n_clips,
                       const int32_t* item_start_dev, const float* item_w_dev, const int32_t* band_first_dev,
                       const int32_t* band_count_dev, int items_per_lane, int n_mels, int log_scale, float* mel_dev,
                       int64_t out_clip_stride, void* stream);

/* ---- timing helper used by bench.py (HIP events on `stream`) ---------- */
/* Runs sg_stft `iters` times back to back between two hipEvents and returns the
 * average milliseconds per launch.  Synchronises `stream`. */
int sg_time_stft(const sg_plan* plan, const void* x_dev, int64_t n_samples, int64_t clip_stride,
                 int n_clips, void* out_dev, int64_t out_clip_stride, void* stream, int iters,
                 float* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* SPECTRO_H */
