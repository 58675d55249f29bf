/* c_client.c -- a plain C caller of include/spectro.h (no Python, no torch, no HIP headers).
 *
 * What a non-Python host (a C/C++ acquisition program, or any FFI that speaks the C ABI) does to get the
 * reference's spectrogram:  scipy.signal.spectrogram(x, fs, nperseg, scaling="density", mode="psd") as
 * called at PlotEngine.py:113, i.e. periodic Tukey(0.25), noverlap = nperseg/8, detrend constant.
 *
 *   c_client <in.f32> <n_samples> <fs> <nperseg> <out.f32>      -> writes [n_frames][nperseg/2+1] float32
 *   c_client --probe                                             -> exit 0 and print the library version;
 *                                                                   exit 3 with the library's message if no GPU
 * build: gcc -O2 -std=c99 -I include examples/c_client.c -L spectrogram-generator_amd/lib -lspectro -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spectro.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc__ = (call);                                                              \
        if (rc__ != SG_OK) {                                                            \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, sg_last_error());      \
            return rc__ == SG_ERR_NO_DEVICE ? 3 : 2;                                    \
        }                                                                               \
    } while (0)

/* scipy.signal.windows.tukey(n, 0.25, sym=False): the symmetric window of n+1 points without its last one */
static void tukey_periodic(int n, double alpha, double* w) {
    const int m = n + 1;
    const double pi = 3.14159265358979323846;
    const int width = (int)floor(alpha * (m - 1) / 2.0);
    for (int i = 0; i < n; ++i) {
        if (i < width + 1) w[i] = 0.5 * (1.0 + cos(pi * (-1.0 + 2.0 * i / alpha / (m - 1))));
        else if (i < m - width - 1) w[i] = 1.0;
        else w[i] = 0.5 * (1.0 + cos(pi * (-2.0 / alpha + 1.0 + 2.0 * i / alpha / (m - 1))));
    }
}

int main(int argc, char** argv) {
    if (argc == 2 && strcmp(argv[1], "--probe") == 0) {
        printf("libspectro ABI version %d\n", sg_version());
        CHECK(sg_init(0));
        char arch[64];
        int cus = 0;
        uint64_t hbm = 0;
        CHECK(sg_device_info(arch, sizeof arch, &cus, &hbm));
        printf("device %s, %d CUs, %.0f GiB\n", arch, cus, (double)hbm / (1 << 30));
        return 0;
    }
    if (argc != 6) {
        fprintf(stderr, "usage: %s <in.f32> <n_samples> <fs> <nperseg> <out.f32> | --probe\n", argv[0]);
        return 1;
    }
    const long long n_samples = atoll(argv[2]);
    const double fs = atof(argv[3]);
    const int nperseg = atoi(argv[4]);
    const int hop = nperseg - nperseg / 8;

    float* x = (float*)malloc(sizeof(float) * (size_t)n_samples);
    FILE* fi = fopen(argv[1], "rb");
    if (!x || !fi || fread(x, sizeof(float), (size_t)n_samples, fi) != (size_t)n_samples) {
        fprintf(stderr, "cannot read %lld samples from %s\n", n_samples, argv[1]);
        return 1;
    }
    fclose(fi);

    CHECK(sg_init(0));
    double* win = (double*)malloc(sizeof(double) * (size_t)nperseg);
    tukey_periodic(nperseg, 0.25, win);
    sg_plan* plan = NULL;
    CHECK(sg_plan_create(&plan, nperseg, nperseg, hop, win, SG_DETREND_CONSTANT, fs, SG_SCALING_DENSITY, SG_MODE_PSD, SG_F32));
    int64_t n_frames = 0;
    int n_bins = 0;
    CHECK(sg_plan_n_frames(plan, n_samples, &n_frames));
    CHECK(sg_plan_n_bins(plan, &n_bins));
    fprintf(stderr, "kernel family %s: %lld frames x %d bins\n", sg_plan_kernel(plan), (long long)n_frames, n_bins);

    void *x_dev = NULL, *out_dev = NULL;
    const size_t out_bytes = sizeof(float) * (size_t)n_frames * (size_t)n_bins;
    CHECK(sg_malloc(&x_dev, sizeof(float) * (size_t)n_samples));
    CHECK(sg_malloc(&out_dev, out_bytes ? out_bytes : 4));
    CHECK(sg_memcpy_h2d(x_dev, x, sizeof(float) * (size_t)n_samples, NULL));
    CHECK(sg_stft(plan, x_dev, n_samples, n_samples, 1, out_dev, n_frames * n_bins, NULL));
    float* out = (float*)malloc(out_bytes ? out_bytes : 4);
    CHECK(sg_memcpy_d2h(out, out_dev, out_bytes, NULL));
    CHECK(sg_stream_sync(NULL));

    FILE* fo = fopen(argv[5], "wb");
    if (!fo || fwrite(out, 1, out_bytes, fo) != out_bytes) {
        fprintf(stderr, "cannot write %s\n", argv[5]);
        return 1;
    }
    fclose(fo);
    CHECK(sg_plan_destroy(plan));
    CHECK(sg_free(x_dev));
    CHECK(sg_free(out_dev));
    free(out);
    free(win);
    free(x);
    return 0;
}
