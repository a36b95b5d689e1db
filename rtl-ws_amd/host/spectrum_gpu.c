/* spectrum_gpu.c -- spectrum.h (drop-in boundary #1) over the HIP shim.
 *
 * Replaces reference src/spectrum.c:37-107.  The device does conversion,
 * DFT, |X|^2 and the fft-shift for one frame IN DOUBLE, as the reference does
 * (rtlws_spectra_batch_f64 with one frame, K = 1: src/spectrum.c:54-58 converts
 * to double, :21 is an f64 FFTW plan, :28 accumulates doubles); this file moves
 * the frame across PCIe and performs the read-modify-write into the caller's
 * host f64 buffer, including the order-dependent DC-slot rule of reference
 * src/spectrum.c:25-33, because that buffer belongs to the caller and lives in
 * host memory.  One frame per call is launch- and PCIe-bound whatever the
 * arithmetic costs, so nothing is gained by computing it in f32; the f32 fused
 * kernel is the batch API's (rtlws_hip.h).
 */
#include "spectrum.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "host_ctx.h"
#include "rtlws_hip.h"

struct spectrum {
    int N;
    rtlws_engine* eng;
    void* d_in;       /* N * 8 bytes: large enough for cmplx_s32 (NULL in zero-copy mode) */
    double* d_out;    /* N doubles (NULL in zero-copy mode) */
    void* h_in;       /* pinned */
    double* h_out;    /* pinned */
    int zero_copy;
};

struct spectrum* spectrum_alloc(int N)
{
    struct spectrum* s;
    rtlws_spectra_desc probe;
    memset(&probe, 0, sizeof probe);
    probe.n_fft = N;
    probe.k_avg = 1;
    if (rtlws_spectra_kernel_kind(&probe) == 0 || N > 8192) {
        fprintf(stderr, "rtlws: spectrum_alloc(%d): size not supported by the device engine\n", N);
        return NULL;
    }
    s = (struct spectrum*)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->N = N;
    s->eng = rtlws_engine_create(rtlws_host_device());
    if (!s->eng) {
        fprintf(stderr, "rtlws: spectrum_alloc: %s\n", rtlws_last_error());
        free(s);
        return NULL;
    }
    {
        const char* z = getenv("RTLWS_DROPIN_ZEROCOPY");
        s->zero_copy = !(z && z[0] == '0');
    }
    if (!s->zero_copy) {             /* device staging exists only in the two-copies mode */
        s->d_in = rtlws_dev_alloc(s->eng, (size_t)N * sizeof(cmplx_s32));
        s->d_out = (double*)rtlws_dev_alloc(s->eng, (size_t)N * sizeof(double));
    }
    s->h_in = rtlws_pinned_alloc((size_t)N * sizeof(cmplx_s32));
    s->h_out = (double*)rtlws_pinned_alloc((size_t)N * sizeof(double));
    if ((!s->zero_copy && (!s->d_in || !s->d_out)) || !s->h_in || !s->h_out) {
        fprintf(stderr, "rtlws: spectrum_alloc: %s\n", rtlws_last_error());
        spectrum_free(s);
        return NULL;
    }
    return s;
}

/* One frame through the device, then the reference's accumulation loop. */
static int add_frame(struct spectrum* s, const void* src, size_t sample_bytes, int input_kind,
                     double* power_spectrum, int len)
{
    rtlws_spectra_desc d;
    const int N = s->N;
    const int offset = N / 2;
    int i;

    if (len != N) return -1;                      /* reference src/spectrum.c:51-52 */

    memset(&d, 0, sizeof d);
    d.n_fft = N;
    d.k_avg = 1;
    d.input = input_kind;
    d.window = RTLWS_WIN_RECT;
    d.output = RTLWS_OUT_POWER_SUM;

    memcpy(s->h_in, src, (size_t)N * sample_bytes);
    /* One frame per call: the kernel reads the frame from, and stores the row into, the pinned
     * (device-mapped) staging buffers itself -- one launch and one synchronisation per call
     * instead of two copies around them (RTLWS_DROPIN_ZEROCOPY=0: the copies; A/B). */
    if (s->zero_copy
            ? (rtlws_spectra_batch_f64(s->eng, &d, s->h_in, 1, s->h_out, NULL) || rtlws_stream_sync(s->eng, NULL))
            : (rtlws_copy_h2d(s->eng, s->d_in, s->h_in, (size_t)N * sample_bytes, NULL) ||
               rtlws_spectra_batch_f64(s->eng, &d, s->d_in, 1, s->d_out, NULL) ||
               rtlws_copy_d2h(s->eng, s->h_out, s->d_out, (size_t)N * sizeof(double), NULL) ||
               rtlws_stream_sync(s->eng, NULL))) {
        fprintf(stderr, "rtlws: spectrum_add: device failure: %s\n", rtlws_last_error());
        return -3;
    }

    /* h_out[i] already is |X[(i + N/2) % N]|^2.  Walk the slots in increasing
     * order so the slot showing bin 0 picks up its left neighbour's updated
     * value (reference src/spectrum.c:25-33). */
    for (i = 0; i < len; i++) {
        if ((offset + i) % len > 0)
            power_spectrum[i] += s->h_out[i];
        else
            power_spectrum[i] += power_spectrum[i - 1];
    }
    return 0;
}

int spectrum_add_cmplx_u8(struct spectrum* s, const cmplx_u8* src, double* power_spectrum, int len)
{
    return add_frame(s, src, sizeof(cmplx_u8), RTLWS_IN_CU8, power_spectrum, len);
}

int spectrum_add_cmplx_s32(struct spectrum* s, const cmplx_s32* src, double* power_spectrum, int len)
{
    return add_frame(s, src, sizeof(cmplx_s32), RTLWS_IN_CS32, power_spectrum, len);
}

int spectrum_add_real_f32(struct spectrum* s, const float* src, double* power_spectrum, int len)
{
    return add_frame(s, src, sizeof(float), RTLWS_IN_RF32, power_spectrum, len);
}

void spectrum_free(struct spectrum* s)
{
    if (!s) return;
    if (s->eng) {
        rtlws_dev_free(s->eng, s->d_in);
        rtlws_dev_free(s->eng, s->d_out);
    }
    rtlws_pinned_free(s->h_in);
    rtlws_pinned_free(s->h_out);
    rtlws_engine_destroy(s->eng);
    free(s);
}
