/* audio_gpu.c -- audio_main.h over the HIP shim (replaces reference
 * src/audio_main.c:1-161).  Per decimator block: H2D, fm_demod kernel,
 * half-band, half-band, D2H of len/4 floats; the phase carry and the two
 * 10-sample delay lines stay on the device between calls.
 */
#include "audio_main.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "host_ctx.h"
#include "resample.h"
#include "rtlws_hip.h"

#define AUDIO_BUFFER_POOL 50          /* reference src/audio_main.c:11 */
#define HIST (HALF_BAND_N - 1)

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static rtlws_engine* g_eng = NULL;
static int g_len = 0;                 /* block length the device buffers are sized for */
static void* g_d_iq = NULL;
static float* g_d_demod = NULL;       /* HIST history + len samples   */
static float* g_d_work = NULL;        /* HIST history + len/2 samples */
static float* g_d_audio = NULL;       /* len/4 samples                */
static float* g_d_phase = NULL;       /* two floats: carry in / out, swapped each call */
static int g_phase_idx = 0;
static cmplx_s32* g_h_iq = NULL;      /* pinned */
static float* g_h_audio = NULL;       /* pinned */

/* FIFO of finished audio buffers */
static float* g_pool[AUDIO_BUFFER_POOL];
static int g_q_head = 0, g_q_count = 0, g_audio_len = 0, g_read_pos = 0;

static void free_device(void)
{
    if (!g_eng) return;
    rtlws_dev_free(g_eng, g_d_iq); g_d_iq = NULL;
    rtlws_dev_free(g_eng, g_d_demod); g_d_demod = NULL;
    rtlws_dev_free(g_eng, g_d_work); g_d_work = NULL;
    rtlws_dev_free(g_eng, g_d_audio); g_d_audio = NULL;
    rtlws_pinned_free(g_h_iq); g_h_iq = NULL;
    rtlws_pinned_free(g_h_audio); g_h_audio = NULL;
}

void audio_init(void)
{
    pthread_mutex_lock(&g_mu);
    if (!g_eng) {
        g_eng = rtlws_engine_create(rtlws_host_device());
        if (!g_eng) {                 /* no CPU path; inert audio side, failure recorded (rtlws_host.h) */
            rtlws_host_fail("audio_init", rtlws_last_error());
        } else {
            g_d_phase = (float*)rtlws_dev_alloc(g_eng, 2 * sizeof(float));
            rtlws_memset_dev(g_eng, g_d_phase, 0, 2 * sizeof(float), NULL);
            rtlws_stream_sync(g_eng, NULL);
            g_len = 0;
            g_phase_idx = 0;
        }
    }
    /* a second audio_init without audio_close only restarts the queue: the phase
     * carry, the delay lines and the block length are function statics in the
     * reference (src/audio_main.c:76-79) and survive there, so they survive here */
    g_q_head = g_q_count = g_read_pos = 0;
    pthread_mutex_unlock(&g_mu);
}

int audio_new_audio_available(void)
{
    int r;
    pthread_mutex_lock(&g_mu);
    r = g_q_count > 0;
    pthread_mutex_unlock(&g_mu);
    return r;
}

int audio_get_audio_payload(char* buf, int buf_len)
{
    int want = buf_len / (int)sizeof(float), copied = 0;
    pthread_mutex_lock(&g_mu);
    while (want > 0 && g_q_count > 0) {
        const float* src = g_pool[g_q_head];
        int n = g_audio_len - g_read_pos;
        if (n > want) n = want;
        memcpy(buf + (size_t)copied * sizeof(float), src + g_read_pos, (size_t)n * sizeof(float));
        copied += n;
        want -= n;
        g_read_pos += n;
        if (g_read_pos >= g_audio_len) {          /* buffer drained: back to the pool */
            g_q_head = (g_q_head + 1) % AUDIO_BUFFER_POOL;
            g_q_count--;
            g_read_pos = 0;
        }
    }
    pthread_mutex_unlock(&g_mu);
    return copied * (int)sizeof(float);
}

static int resize_for(int len)
{
    int i;
    free_device();
    g_len = len;
    g_d_iq = rtlws_dev_alloc(g_eng, (size_t)len * sizeof(cmplx_s32));
    g_d_demod = (float*)rtlws_dev_alloc(g_eng, (size_t)(HIST + len) * sizeof(float));
    g_d_work = (float*)rtlws_dev_alloc(g_eng, (size_t)(HIST + len / 2) * sizeof(float));
    g_d_audio = (float*)rtlws_dev_alloc(g_eng, (size_t)(len / 4 + 1) * sizeof(float));
    g_h_iq = (cmplx_s32*)rtlws_pinned_alloc((size_t)len * sizeof(cmplx_s32));
    g_h_audio = (float*)rtlws_pinned_alloc((size_t)(len / 4 + 1) * sizeof(float));
    if (!g_d_iq || !g_d_demod || !g_d_work || !g_d_audio || !g_h_iq || !g_h_audio) return -3;
    /* reference src/audio_main.c:82-104: a new block length restarts the queue
     * (delay lines and phase carry are function statics there and survive) */
    rtlws_memset_dev(g_eng, g_d_demod, 0, HIST * sizeof(float), NULL);
    rtlws_memset_dev(g_eng, g_d_work, 0, HIST * sizeof(float), NULL);
    g_audio_len = (len / 2) / 2;
    for (i = 0; i < AUDIO_BUFFER_POOL; i++) {
        free(g_pool[i]);
        g_pool[i] = (float*)calloc((size_t)(g_audio_len > 0 ? g_audio_len : 1), sizeof(float));
    }
    g_q_head = g_q_count = g_read_pos = 0;
    return 0;
}

void audio_fm_demodulator(const cmplx_s32* signal, int len)
{
    const int half = len / 2, quarter = half / 2;
    int rc = 0, have_buf;
    if (len <= 0) return;
    pthread_mutex_lock(&g_mu);
    if (!g_eng) {                     /* before audio_init, or audio_init found no device: nothing is queued */
        pthread_mutex_unlock(&g_mu);
        rtlws_host_fail("audio_fm_demodulator", "no engine (audio_init not called, or no usable HIP device)");
        return;
    }
    if (g_len != len) {
        float hist1[HIST], hist2[HIST];
        int have = g_len > 0;
        /* carry the delay lines over a change of block length, as the
         * reference's statics do */
        if (have) {
            rtlws_copy_d2h(g_eng, hist1, g_d_demod, sizeof hist1, NULL);
            rtlws_copy_d2h(g_eng, hist2, g_d_work, sizeof hist2, NULL);
            rtlws_stream_sync(g_eng, NULL);
        }
        rc = resize_for(len);
        if (!rc && have) {
            rtlws_copy_h2d(g_eng, g_d_demod, hist1, sizeof hist1, NULL);
            rtlws_copy_h2d(g_eng, g_d_work, hist2, sizeof hist2, NULL);
            rtlws_stream_sync(g_eng, NULL);
        }
    }
    /* reference src/audio_main.c:137-142: the second half-band runs -- and its delay
     * line advances -- only when a pool buffer is free to take its output; a
     * block that meets an exhausted pool leaves delay line 2 untouched */
    have_buf = g_q_count < AUDIO_BUFFER_POOL;
    if (!rc) {
        float* prev_in = g_d_phase + g_phase_idx;
        float* prev_out = g_d_phase + (1 - g_phase_idx);
        memcpy(g_h_iq, signal, (size_t)len * sizeof(cmplx_s32));
        if (rtlws_copy_h2d(g_eng, g_d_iq, g_h_iq, (size_t)len * sizeof(cmplx_s32), NULL) ||
            rtlws_fm_demod(g_eng, g_d_iq, len, prev_in, prev_out, g_d_demod + HIST, NULL) ||
            rtlws_halfband(g_eng, g_d_demod, g_d_work + HIST, half, NULL) ||              /* :133 */
            /* delay line 1 <- last 10 inputs of the stage (src/resample.c:66) */
            rtlws_copy_d2d(g_eng, g_d_demod, g_d_demod + 2 * half, HIST * sizeof(float), NULL) ||
            (have_buf &&
             (rtlws_halfband(g_eng, g_d_work, g_d_audio, quarter, NULL) ||               /* :139 */
              rtlws_copy_d2d(g_eng, g_d_work, g_d_work + 2 * quarter, HIST * sizeof(float), NULL) ||
              rtlws_copy_d2h(g_eng, g_h_audio, g_d_audio, (size_t)quarter * sizeof(float), NULL))) ||
            rtlws_stream_sync(g_eng, NULL))
            rc = -3;
        else
            g_phase_idx = 1 - g_phase_idx;
    }
    if (rc) {                         /* void signature: the block yields no audio, the failure is recorded */
        pthread_mutex_unlock(&g_mu);
        rtlws_host_fail("audio_fm_demodulator", rtlws_last_error());
        return;
    }
    if (have_buf && quarter > 0) {
        float* dst = g_pool[(g_q_head + g_q_count) % AUDIO_BUFFER_POOL];
        memcpy(dst, g_h_audio, (size_t)quarter * sizeof(float));
        g_q_count++;
    }
    pthread_mutex_unlock(&g_mu);
}

void audio_close(void)
{
    int i;
    pthread_mutex_lock(&g_mu);
    free_device();
    if (g_eng) {
        rtlws_dev_free(g_eng, g_d_phase);
        g_d_phase = NULL;
        rtlws_engine_destroy(g_eng);
        g_eng = NULL;
    }
    for (i = 0; i < AUDIO_BUFFER_POOL; i++) { free(g_pool[i]); g_pool[i] = NULL; }
    g_len = 0;
    g_q_head = g_q_count = g_read_pos = 0;
    pthread_mutex_unlock(&g_mu);
}
