/* dropin_demo.c -- a reference-style caller, strict C99, written only against
 * the drop-in headers.  It is what reference src/cbb_main.c:40-70 does with the
 * estimator, plus the decimator calls of src/rf_decimator.c's users, so that a
 * header or ABI regression shows up at build time (this file is compiled with
 * -std=c99 -pedantic -Werror) and at run time on the GPU (tests/test_dropin_demo_gpu.py).
 * Prints a few checksums; exit code 0 on success.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "resample.h"
#include "rf_decimator.h"
#include "spectrum.h"

#define FFT_POINTS 1024
#define FFT_AVERAGE 6

static long g_blocks = 0;
static long long g_sum = 0;

static void on_block(const cmplx_s32* iq, int len)
{
    int i;
    g_blocks++;
    for (i = 0; i < len; i++) g_sum += iq[i].p.re + 3 * (long long)iq[i].p.im;
}

int main(void)
{
    static cmplx_u8 signal[FFT_AVERAGE * FFT_POINTS];
    static double power_spectrum[FFT_POINTS];
    struct spectrum* spect;
    struct rf_decimator* decim;
    struct cic_delay_line delay;
    static cmplx_s32 dec[FFT_POINTS];
    float hb_in[64], hb_out[32], hb_delay[HALF_BAND_N - 1];
    int i, peak = 0;

    /* a tone at bin +100 and a little deterministic dither */
    for (i = 0; i < FFT_AVERAGE * FFT_POINTS; i++) {
        double ph = 2.0 * 3.14159265358979323846 * 100.0 * (double)i / FFT_POINTS;
        signal[i].re = (uint8_t)(128.0 + 100.0 * cos(ph) + (double)((i * 7) % 3) - 1.0);
        signal[i].im = (uint8_t)(128.0 + 100.0 * sin(ph) + (double)((i * 5) % 3) - 1.0);
    }

    spect = spectrum_alloc(FFT_POINTS);
    if (!spect) { fprintf(stderr, "spectrum_alloc failed (no HIP device?)\n"); return 2; }
    memset(power_spectrum, 0, sizeof power_spectrum);
    for (i = 0; i < FFT_AVERAGE; i++)
        if (spectrum_add_cmplx_u8(spect, &signal[i * FFT_POINTS], power_spectrum, FFT_POINTS)) return 3;
    if (spectrum_add_cmplx_u8(spect, signal, power_spectrum, FFT_POINTS - 1) != -1) return 4;
    for (i = 1; i < FFT_POINTS; i++)
        if (power_spectrum[i] > power_spectrum[peak]) peak = i;
    printf("peak_slot %d expected %d\n", peak, (100 + FFT_POINTS / 2) % FFT_POINTS);
    printf("dc_slot_equals_running_neighbour %d\n",
           power_spectrum[FFT_POINTS / 2] > power_spectrum[FFT_POINTS / 2 - 1]);
    spectrum_free(spect);
    if (peak != (100 + FFT_POINTS / 2) % FFT_POINTS) return 5;

    memset(&delay, 0, sizeof delay);
    if (cic_decimate(6, signal, FFT_AVERAGE * FFT_POINTS, dec, FFT_POINTS, &delay)) return 6;
    if (cic_decimate(6, signal, 100, dec, 7, &delay) != -1) return 7;
    printf("cic dec[0] %d %d state %d %d\n", dec[0].p.re, dec[0].p.im,
           delay.integrator_prev_out.p.re, delay.comb_prev_in.p.im);

    for (i = 0; i < 64; i++) hb_in[i] = (float)((i * 37) % 11) - 5.0f;
    memset(hb_delay, 0, sizeof hb_delay);
    halfband_decimate(hb_in, hb_out, 32, hb_delay);
    printf("halfband out[7] %.6f delay[9] %.1f\n", hb_out[7], hb_delay[9]);

    decim = rf_decimator_alloc();
    if (rf_decimator_decimate_cmplx_u8(decim, signal, 10) != -1) return 8;      /* unconfigured */
    rf_decimator_add_callback(decim, on_block);
    if (rf_decimator_set_parameters(decim, 10240.0, 4)) return 9;               /* 256-out / 1024-in blocks */
    for (i = 0; i < FFT_AVERAGE; i++)
        if (rf_decimator_decimate_cmplx_u8(decim, &signal[i * FFT_POINTS + (i ? 1 : 0)],
                                           i ? FFT_POINTS : FFT_POINTS + 1)) return 10;
    printf("rf_decimator blocks %ld checksum %lld\n", g_blocks, g_sum);
    rf_decimator_free(decim);
    return g_blocks == 6 ? 0 : 11;
}
