// host_shim.cpp -- the parts of the C ABI that never touch the device: error plumbing, argument triage of sg_plan_create,
// the bit-exact f / t vectors (A7), the mel bank construction and the jet lookup table.  Plain C++17 with no HIP header,
// so the same file is compiled into libspectro.so (by hipcc) AND, with g++ -fsanitize=address,undefined, into the
// sanitizer driver of tests/test_host_logic.py (GPU sanitizers are not available on this pool).
#include "host_shim.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

namespace sg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_plan_args(int nperseg, int nfft, int hop, int detrend, double fs, int scaling, int mode, int dtype) {
    // messages follow scipy's (scipy/signal/_spectral_py.py:2031-2041, 960-962, 2091) where scipy has one
    if (nperseg < 1) { set_error("nperseg must be a positive integer"); return SG_ERR_ARG; }
    if (nfft < nperseg) { set_error("nfft must be greater than or equal to nperseg."); return SG_ERR_ARG; }
    if (hop < 1 || hop > nperseg) { set_error("noverlap must be less than nperseg."); return SG_ERR_ARG; }
    if (detrend < 0 || detrend > 2) { set_error("Trend type must be 'linear' or 'constant'."); return SG_ERR_ARG; }
    if (scaling < 0 || scaling > 1) { set_error("Unknown scaling: %d", scaling); return SG_ERR_ARG; }
    if (mode < 0 || mode > 3) { set_error("unknown value for mode %d", mode); return SG_ERR_ARG; }
    if (dtype != SG_F32 && dtype != SG_F64) { set_error("bad dtype %d", dtype); return SG_ERR_ARG; }
    if (!(fs > 0.0) || !std::isfinite(fs)) { set_error("fs must be positive and finite"); return SG_ERR_ARG; }
    return SG_OK;
}

const char* last_error_cstr() { return g_err; }

}  // namespace sg

using namespace sg;

extern "C" {

int sg_version(void) { return SG_VERSION; }

const char* sg_last_error(void) { return last_error_cstr(); }

// A7 -- must reproduce numpy's arithmetic exactly:
//   rfftfreq(n, d): val = 1.0/(n*d); results = arange(0, n//2+1) * val         (scipy:2115, d = 1/fs)
//   time = arange(nperseg/2, N - nperseg/2 + 1, step) / float(fs)              (scipy:2136-2137)
// numpy's arange(start, stop, step) for doubles fills start + i*step.
int sg_freqs(int nfft, double fs, double* f_out) {
    if (!f_out || nfft < 1 || !(fs > 0)) { set_error("bad argument"); return SG_ERR_ARG; }
    const double d = 1.0 / fs;
    const double val = 1.0 / (static_cast<double>(nfft) * d);
    for (int k = 0; k <= nfft / 2; ++k) f_out[k] = static_cast<double>(k) * val;
    return SG_OK;
}

int sg_times(int64_t n_samples, int nperseg, int hop, double fs, double* t_out) {
    if (nperseg < 1 || hop < 1 || !(fs > 0)) { set_error("bad argument"); return SG_ERR_ARG; }
    if (n_samples < nperseg) return SG_OK;
    if (!t_out) { set_error("null pointer"); return SG_ERR_ARG; }
    const int64_t n = (n_samples - nperseg) / hop + 1;
    const double start = static_cast<double>(nperseg) / 2.0;
    // numpy arange: first two values are start and start+step, the rest start + i*delta with delta = (start+step)-start
    const double delta = (start + static_cast<double>(hop)) - start;
    for (int64_t i = 0; i < n; ++i) t_out[i] = (start + static_cast<double>(i) * delta) / fs;
    return SG_OK;
}

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

int sg_mel_pack_weights(const double* weights_host, int n_bins, int n_mels, float* packed_host) {
    if (!weights_host || !packed_host || n_bins < 1 || n_mels < 1) { set_error("bad argument"); return SG_ERR_ARG; }
    const int k_pad = (n_bins + 15) & ~15, m_pad = ((n_mels + 15) / 16) * 16;
    for (int m = 0; m < m_pad; ++m)
        for (int k = 0; k < k_pad; ++k)
            packed_host[static_cast<size_t>(m) * k_pad + k] =
                (m < n_mels && k < n_bins) ? static_cast<float>(weights_host[static_cast<size_t>(k) * n_mels + m]) : 0.f;
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

int sg_mel_sparse_pack(const double* weights_host, int n_bins, int n_mels, int* items_per_lane, int32_t* item_start,
                       float* item_w, int32_t* band_first, int32_t* band_count) {
    if (!weights_host || !items_per_lane || !item_start || !item_w || !band_first || !band_count || n_bins < 1 || n_mels < 1) {
        set_error("sg_mel_sparse_pack: bad argument");
        return SG_ERR_ARG;
    }
    *items_per_lane = 0;
    if (n_mels > 128) { set_error("sg_mel_sparse_pack: more than 128 bands"); return SG_ERR_UNSUPPORTED; }
    // work items: the hull [lo, hi) of a band's non-zero bins cut into pieces of 8 bins
    struct Item { int start, band; };
    std::vector<Item> items;
    for (int j = 0; j < n_mels; ++j) {
        int lo = n_bins, hi = 0;
        for (int k = 0; k < n_bins; ++k)
            if (weights_host[static_cast<size_t>(k) * n_mels + j] != 0.0) { if (k < lo) lo = k; hi = k + 1; }
        band_first[j] = static_cast<int32_t>(items.size());
        int n = 0;
        for (int s = lo; s < hi; s += 8, ++n) items.push_back({s, j});
        band_count[j] = n;
        if (items.size() > 256) {
            set_error("sg_mel_sparse_pack: the bank needs more than 256 work items (a dense bank: use sg_mel / sg_stft_mel)");
            return SG_ERR_UNSUPPORTED;
        }
    }
    const int ipl = items.empty() ? 1 : static_cast<int>((items.size() + 63) / 64);
    const int slots = 64 * ipl;
    for (int i = 0; i < slots; ++i) {
        const bool live = i < static_cast<int>(items.size());
        item_start[i] = live ? items[i].start : 0;
        for (int c = 0; c < 8; ++c) {
            const int k = live ? items[i].start + c : 0;
            item_w[static_cast<size_t>(c) * slots + i] =
                (live && k < n_bins) ? static_cast<float>(weights_host[static_cast<size_t>(k) * n_mels + items[i].band]) : 0.f;
        }
    }
    *items_per_lane = ipl;
    return SG_OK;
}

int sg_jet_lut(uint8_t* rgba_host) {
    if (!rgba_host) { set_error("null pointer"); return SG_ERR_ARG; }
    // matplotlib _cm.py 'jet' segment data: (x, y) breakpoints, linear in between
    static const double R[][2] = {{0, 0}, {0.35, 0}, {0.66, 1}, {0.89, 1}, {1, 0.5}};
    static const double G[][2] = {{0, 0}, {0.125, 0}, {0.375, 1}, {0.64, 1}, {0.91, 0}, {1, 0}};
    static const double B[][2] = {{0, 0.5}, {0.11, 1}, {0.34, 1}, {0.65, 0}, {1, 0}};
    // same arithmetic as matplotlib.colors._create_lookup_table (N = 256, gamma = 1): breakpoints scaled by N-1,
    // sample i at (N-1) * (i * (1/(N-1))), left searchsorted, linear blend; end points taken verbatim
    auto interp = [](const double (*seg)[2], int n, int i) {
        if (i == 0) return seg[0][1];
        if (i == 255) return seg[n - 1][1];
        const double xind = 255.0 * (static_cast<double>(i) * (1.0 / 255.0));
        int ind = 0;
        while (ind < n && seg[ind][0] * 255.0 < xind) ++ind;
        const double x0 = seg[ind - 1][0] * 255.0, x1 = seg[ind][0] * 255.0;
        const double distance = (xind - x0) / (x1 - x0);
        return distance * (seg[ind][1] - seg[ind - 1][1]) + seg[ind - 1][1];
    };
    for (int i = 0; i < 256; ++i) {
        const double c[3] = {interp(R, 5, i), interp(G, 6, i), interp(B, 5, i)};
        for (int k = 0; k < 3; ++k) {
            double v = c[k] < 0 ? 0 : (c[k] > 1 ? 1 : c[k]);
            rgba_host[4 * i + k] = static_cast<uint8_t>(v * 255.0);      // matplotlib bytes=True: (lut * 255).astype(uint8)
        }
        rgba_host[4 * i + 3] = 255;
    }
    return SG_OK;
}

}  // extern "C"
