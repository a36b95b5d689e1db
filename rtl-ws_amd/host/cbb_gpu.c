/* cbb_gpu.c -- cbb_main.h (drop-in boundary #2) over the HIP engine.
 *
 * Replaces reference src/cbb_main.c:1-153.  Per sensor buffer (worker thread):
 * count samples, feed the decimator, and -- at most every 250 ms -- turn the
 * first min(len/1024, 6) frames into ONE launch of the f64 kernel (K = blocks;
 * double power sums, as reference src/cbb_main.c:27,50-59 keeps them, that
 * stay on the device).  On the server thread, cbb_get_spectrum_payload runs
 * the dB/clamp kernel -- in double, src/cbb_main.c:125 -- on the published
 * sums and brings back the 1024 bytes main.c sends to the browser.
 *
 * The sensor thread never waits for the device: it copies
 * the frames into one of two pinned slots, enqueues the H2D copy and the kernel
 * on the engine's stream and returns; the only wait is on a slot whose copy
 * from two estimates (>= 500 ms) ago has not finished, which does not happen.
 * The server thread's payload kernel is ordered behind it on the same stream
 * and is the one that synchronises.
 *
 * The sensor (rtl_sensor.h) and the signal source (signal_source.h) are NOT
 * part of this file: they are the reference's own units, or the synthetic
 * ones in synth_sensor.c / synth_signal_source.c.
 */
#include "cbb_main.h"
#include "rtlws_cbb.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "host_ctx.h"
#include "rtlws_hip.h"
#include "signal_source.h"

#define DEV_INDEX       0       /* reference src/cbb_main.c:15 */
#define SPECTRUM_EST_MS 250     /* :16 */
#define FFT_POINTS      1024    /* :17 */
#define FFT_AVERAGE     6       /* :18 */

static struct rtl_dev* g_dev = NULL;
static struct rf_decimator* g_decim = NULL;
static rtlws_engine* g_eng = NULL;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;   /* device work + published state */
static void* g_d_iq = NULL;             /* FFT_AVERAGE frames of cmplx_u8 */
static double* g_d_work = NULL;         /* sums being produced            */
static double* g_d_pub = NULL;          /* sums the server thread reads   */
static void* g_d_payload = NULL;
#define IQ_SLOTS 2
static cmplx_u8* g_h_iq[IQ_SLOTS];      /* pinned staging slots, used in turn      */
static void* g_iq_done[IQ_SLOTS];       /* event: that slot's H2D copy has finished */
static int g_iq_used[IQ_SLOTS];
static int g_iq_next = 0;
static unsigned char* g_h_payload = NULL;
static int g_pub_count = 0;             /* frames behind g_d_pub */
static uint64_t g_last_est_ms = 0;
/* written on the sensor thread, read on the server thread, without g_mu: the reference keeps these in a
 * `volatile int` (src/cbb_main.c:21,28), which is a data race in C11; here they are atomics */
static int g_new_spectrum = 0;
static uint64_t g_samples_seen = 0;
#define FLAG_SET(v)   __atomic_store_n(&g_new_spectrum, (v), __ATOMIC_RELEASE)
#define FLAG_GET()    __atomic_load_n(&g_new_spectrum, __ATOMIC_ACQUIRE)
static int g_max_blocks = FFT_AVERAGE;  /* RTLWS_CBB_ALL_FRAMES=1|2: every frame of the buffer */
#define MAX_BLOCKS_ALL 1024             /* device staging bound for those modes (2 MiB of IQ) */
/* RTLWS_CBB_ALL_FRAMES=2 ("Welch averaging over the whole 250 ms"):
 * EVERY sensor buffer is transformed, all of its frames, and folded into a running
 * row; the 250 ms gate only decides when that row is published.  The published
 * spectrum is what the reference's loop would leave after all those frames. */
static int g_welch = 0;
static double* g_d_acc = NULL;          /* running row of the current interval            */
static double* g_d_b = NULL;            /* scalar carried beside it (rtlws_welch_*_f64)   */
static long g_acc_count = 0;            /* frames folded into g_d_acc so far              */

static uint64_t now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);          /* as reference src/common.c:4-9 */
    return (uint64_t)ts.tv_sec * 1000u + (uint64_t)ts.tv_nsec / 1000000u;
}

/* callback #1 (reference src/cbb_main.c:30-33): the reference prints a rate
 * line every 60 s; the engine only keeps the count. */
static void count_samples(const cmplx_u8* signal, int len)
{
    (void)signal;
    __atomic_fetch_add(&g_samples_seen, (uint64_t)(len > 0 ? len : 0), __ATOMIC_RELAXED);
}

/* callback #2 (reference src/cbb_main.c:35-38) */
static void decimate(const cmplx_u8* signal, int len)
{
    rf_decimator_decimate_cmplx_u8(g_decim, signal, len);
}

/* callback #3 (reference src/cbb_main.c:40-70) */
static void estimate_spectrum(const cmplx_u8* signal, int len)
{
    int blocks = len / FFT_POINTS;
    rtlws_spectra_desc d;

    const int due = now_ms() >= g_last_est_ms + SPECTRUM_EST_MS;  /* :46-47 */
    if (!g_eng) return;                          /* cbb_init found no device: inert (rtlws_host.h) */
    if (!due && !g_welch) return;
    blocks = blocks <= g_max_blocks ? blocks : g_max_blocks;     /* :49 (6 unless ALL_FRAMES) */

    pthread_mutex_lock(&g_mu);
    if (blocks > 0) {
        const size_t bytes = (size_t)blocks * FFT_POINTS * sizeof(cmplx_u8);
        memset(&d, 0, sizeof d);
        d.n_fft = FFT_POINTS;
        d.k_avg = blocks;                       /* one output row = the whole average */
        d.input = RTLWS_IN_CU8;
        d.window = RTLWS_WIN_RECT;
        d.output = RTLWS_OUT_POWER_SUM;
        const int slot = g_iq_next;
        g_iq_next = (g_iq_next + 1) % IQ_SLOTS;
        if (g_iq_used[slot]) rtlws_event_sync(g_iq_done[slot]);   /* copy of 2 estimates ago */
        memcpy(g_h_iq[slot], signal, bytes);
        g_iq_used[slot] = 1;
        if (rtlws_copy_h2d(g_eng, g_d_iq, g_h_iq[slot], bytes, NULL) ||
            rtlws_event_record(g_iq_done[slot], g_eng, NULL) ||
            rtlws_spectra_batch_f64(g_eng, &d, g_d_iq, blocks, g_d_work, NULL) ||
            (g_welch && rtlws_welch_accumulate_f64(g_eng, g_d_acc, g_d_work, FFT_POINTS,
                                                   g_acc_count + blocks, g_d_b, NULL))) {
            fprintf(stderr, "rtlws: estimate_spectrum: device failure: %s\n", rtlws_last_error());
            pthread_mutex_unlock(&g_mu);
            return;                                               /* :54-58 */
        }
        g_acc_count += blocks;
    }
    if (g_welch) {
        if (!due) {                              /* folded in; the interval is still open */
            pthread_mutex_unlock(&g_mu);
            return;
        }
        /* close the interval: fix slot N/2 for the whole sequence, publish, start over */
        if (rtlws_welch_finish_f64(g_eng, g_d_acc, FFT_POINTS, g_acc_count, g_d_b, NULL)) {
            fprintf(stderr, "rtlws: estimate_spectrum: device failure: %s\n", rtlws_last_error());
            pthread_mutex_unlock(&g_mu);
            return;
        }
        {
            double* t = g_d_pub;
            g_d_pub = g_d_acc;
            g_d_acc = t;
            g_pub_count = (int)g_acc_count;
            g_acc_count = 0;
            rtlws_memset_dev(g_eng, g_d_acc, 0, FFT_POINTS * sizeof(double), NULL);
        }
        g_last_est_ms = now_ms();
        FLAG_SET(1);
        pthread_mutex_unlock(&g_mu);
        return;
    }
    g_last_est_ms = now_ms();                                     /* :61 */
    FLAG_SET(1);                                                  /* :62 */
    {   /* publish: swap instead of the reference's 8 KiB memcpy (:64-69) */
        double* t = g_d_pub;
        g_d_pub = g_d_work;
        g_d_work = t;
        g_pub_count = blocks;
    }
    pthread_mutex_unlock(&g_mu);
}

/* everything cbb_init created on the device side (also the way out of a failed cbb_init) */
static void free_engine_side(void)
{
    int k;
    if (!g_eng) return;
    rtlws_stream_sync(g_eng, NULL);
    rtlws_dev_free(g_eng, g_d_iq);
    rtlws_dev_free(g_eng, g_d_work);
    rtlws_dev_free(g_eng, g_d_pub);
    rtlws_dev_free(g_eng, g_d_payload);
    rtlws_dev_free(g_eng, g_d_acc);
    rtlws_dev_free(g_eng, g_d_b);
    g_d_iq = g_d_payload = NULL;
    g_d_work = g_d_pub = g_d_acc = g_d_b = NULL;
    for (k = 0; k < IQ_SLOTS; ++k) {
        rtlws_pinned_free(g_h_iq[k]);
        rtlws_event_destroy(g_iq_done[k]);
        g_h_iq[k] = NULL;
        g_iq_done[k] = NULL;
    }
    rtlws_pinned_free(g_h_payload);
    g_h_payload = NULL;
    rtlws_engine_destroy(g_eng);                                  /* replaces spectrum_free, :145 */
    g_eng = NULL;
}

/* the device side of cbb_init (replaces spectrum_alloc, reference src/cbb_main.c:83); 0 / -3 */
static int init_engine_side(void)
{
    int k;
    g_eng = rtlws_engine_create(rtlws_host_device());
    if (!g_eng) return -3;
    {   /* the reference transforms 6 of the ~128 frames a sensor
         * buffer carries (:46-49); with RTLWS_CBB_ALL_FRAMES=1 the same launch averages
         * all of them (K = len/1024) -- same payload format, smoother spectrum */
        const char* all = getenv("RTLWS_CBB_ALL_FRAMES");
        g_max_blocks = (all && atoi(all) > 0) ? MAX_BLOCKS_ALL : FFT_AVERAGE;
        g_welch = (all && atoi(all) == 2);
    }
    g_d_iq = rtlws_dev_alloc(g_eng, (size_t)g_max_blocks * FFT_POINTS * sizeof(cmplx_u8));
    g_d_work = (double*)rtlws_dev_alloc(g_eng, FFT_POINTS * sizeof(double));
    g_d_pub = (double*)rtlws_dev_alloc(g_eng, FFT_POINTS * sizeof(double));
    g_d_payload = rtlws_dev_alloc(g_eng, FFT_POINTS);
    g_d_acc = (double*)rtlws_dev_alloc(g_eng, FFT_POINTS * sizeof(double));
    g_d_b = (double*)rtlws_dev_alloc(g_eng, sizeof(double));
    if (g_d_acc) rtlws_memset_dev(g_eng, g_d_acc, 0, FFT_POINTS * sizeof(double), NULL);
    if (g_d_b) rtlws_memset_dev(g_eng, g_d_b, 0, sizeof(double), NULL);
    g_acc_count = 0;
    for (k = 0; k < IQ_SLOTS; ++k) {
        g_h_iq[k] = (cmplx_u8*)rtlws_pinned_alloc((size_t)g_max_blocks * FFT_POINTS * sizeof(cmplx_u8));
        g_iq_done[k] = rtlws_event_create();
        g_iq_used[k] = 0;
    }
    g_iq_next = 0;
    g_h_payload = (unsigned char*)rtlws_pinned_alloc(FFT_POINTS);
    if (!g_d_iq || !g_d_work || !g_d_pub || !g_d_payload || !g_d_acc || !g_d_b || !g_h_iq[0] || !g_h_iq[1] ||
        !g_iq_done[0] || !g_iq_done[1] || !g_h_payload)
        return -3;
    return 0;
}

void cbb_init(int decimated_bw_target_hz)
{
    rtl_init(&g_dev, DEV_INDEX);                                  /* :77 */

    g_decim = rf_decimator_alloc();                               /* :79-80 */
    rf_decimator_set_parameters(g_decim, rtl_sample_rate(g_dev),
                                (int)(rtl_sample_rate(g_dev) / (uint32_t)decimated_bw_target_hz));

    if (init_engine_side() != 0) {              /* no CPU path to fall back to, and no reason to take the */
        rtlws_host_fail("cbb_init", rtlws_last_error());      /* server down: inert spectrum side (rtlws_host.h) */
        free_engine_side();
    }
    g_pub_count = 0;
    g_last_est_ms = 0;
    FLAG_SET(0);
    __atomic_store_n(&g_samples_seen, (uint64_t)0, __ATOMIC_RELAXED);

    signal_source_start(g_dev);                                   /* :85 */
    signal_source_add_callback(count_samples);                    /* :86 */
    signal_source_add_callback(decimate);                         /* :87 */
    signal_source_add_callback(estimate_spectrum);                /* :88 */
}

struct rtl_dev* cbb_get_rtl_dev(void) { return g_dev; }          /* :91-94 */

struct rf_decimator* cbb_rf_decimator(void) { return g_decim; }   /* :96-99 */

int cbb_new_spectrum_available(void) { return FLAG_GET(); }       /* :101-104 */

int cbb_get_spectrum_payload(char* buf, int buf_len, int spectrum_gain_db)
{
    int len = 0;
    pthread_mutex_lock(&g_mu);
    if (g_pub_count > 0) {                                        /* :121 */
        if (rtlws_payload_from_sums_f64(g_eng, g_d_pub, FFT_POINTS, g_pub_count, spectrum_gain_db,
                                    g_d_payload, NULL) ||
            rtlws_copy_d2h(g_eng, g_h_payload, g_d_payload, FFT_POINTS, NULL) ||
            rtlws_stream_sync(g_eng, NULL)) {
            fprintf(stderr, "rtlws: cbb_get_spectrum_payload: device failure: %s\n", rtlws_last_error());
        } else {
            len = buf_len < FFT_POINTS ? (buf_len > 0 ? buf_len : 0) : FFT_POINTS;
            memcpy(buf, g_h_payload, (size_t)len);                /* :123-129 */
        }
    }
    pthread_mutex_unlock(&g_mu);
    FLAG_SET(0);                                                  /* :132 */
    return len;
}

void cbb_close(void)
{
    signal_source_remove_callbacks();                             /* :139 */
    signal_source_stop();                                         /* :141 */
    rf_decimator_free(g_decim);                                   /* :143 */
    g_decim = NULL;
    pthread_mutex_lock(&g_mu);
    free_engine_side();
    g_pub_count = 0;
    pthread_mutex_unlock(&g_mu);
    rtl_close(g_dev);                                             /* :149 */
    g_dev = NULL;
}

/* test/diagnostic hook: frames behind the spectrum the next payload call will convert */
int rtlws_cbb_published_frames(void)
{
    int n;
    pthread_mutex_lock(&g_mu);
    n = g_pub_count;
    pthread_mutex_unlock(&g_mu);
    return n;
}

/* test/diagnostic hook: complex samples the callbacks have seen so far */
uint64_t rtlws_cbb_samples_seen(void) { return __atomic_load_n(&g_samples_seen, __ATOMIC_RELAXED); }
