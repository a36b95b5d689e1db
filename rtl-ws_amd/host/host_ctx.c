/* host_ctx.c -- see host_ctx.h. */
#include "host_ctx.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct rtlws_host_ctx {
    rtlws_engine* eng;
    void* d_in;   size_t d_in_cap;
    void* d_out;  size_t d_out_cap;
    void* h_in;   size_t h_in_cap;     /* pinned */
    void* h_out;  size_t h_out_cap;    /* pinned */
};

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static struct rtlws_host_ctx g_ctx;
static int g_tried = 0;
static int g_zero_copy = 1;

/* The drop-in entry points hand over host buffers: by default the kernels read and write the
 * pinned (device-mapped) staging buffers themselves -- one launch and one synchronisation per
 * call; RTLWS_DROPIN_ZEROCOPY=0 puts an H2D and a D2H copy around the launch instead (A/B).
 * Read once, when the context is created; the device staging buffers exist only in that mode. */
static int zero_copy(void) { return g_zero_copy; }

/* ---- sticky failure record (include/rtlws_host.h) ---------------------- */
static pthread_mutex_t g_err_mu = PTHREAD_MUTEX_INITIALIZER;
static char g_err_first[320];
static long g_err_count = 0;

void rtlws_host_fail(const char* where, const char* what)
{
    long n;
    pthread_mutex_lock(&g_err_mu);
    n = ++g_err_count;
    if (n == 1) snprintf(g_err_first, sizeof g_err_first, "%s: %s", where, what ? what : "");
    pthread_mutex_unlock(&g_err_mu);
    if (n == 1 || n % 1024 == 0)
        fprintf(stderr, "rtlws: %s: %s (failure %ld; no CPU path: the output of this call is zeros / empty)\n",
                where, what ? what : "", n);
}

const char* rtlws_host_error(void) { return g_err_first; }

long rtlws_host_error_count(void)
{
    long n;
    pthread_mutex_lock(&g_err_mu);
    n = g_err_count;
    pthread_mutex_unlock(&g_err_mu);
    return n;
}

void rtlws_host_error_clear(void)
{
    pthread_mutex_lock(&g_err_mu);
    g_err_count = 0;
    g_err_first[0] = 0;
    pthread_mutex_unlock(&g_err_mu);
}

int rtlws_host_device(void)
{
    const char* s = getenv("RTLWS_DEVICE");
    return s ? atoi(s) : 0;
}

struct rtlws_host_ctx* rtlws_host_ctx_get(void)
{
    struct rtlws_host_ctx* r = NULL;
    pthread_mutex_lock(&g_mu);
    if (!g_ctx.eng && !g_tried) {
        const char* z = getenv("RTLWS_DROPIN_ZEROCOPY");
        g_zero_copy = !(z && z[0] == '0');
        g_tried = 1;
        g_ctx.eng = rtlws_engine_create(rtlws_host_device());
        if (!g_ctx.eng)
            fprintf(stderr, "rtlws: no usable HIP device (%s); this library has no CPU path\n",
                    rtlws_last_error());
    }
    if (g_ctx.eng) r = &g_ctx;
    pthread_mutex_unlock(&g_mu);
    return r;
}

static int grow_dev(rtlws_engine* e, void** p, size_t* cap, size_t need)
{
    if (*cap >= need) return 0;
    rtlws_dev_free(e, *p);
    *p = rtlws_dev_alloc(e, need);
    *cap = *p ? need : 0;
    return *p ? 0 : -3;
}

static int grow_pinned(void** p, size_t* cap, size_t need)
{
    if (*cap >= need) return 0;
    rtlws_pinned_free(*p);
    *p = rtlws_pinned_alloc(need);
    *cap = *p ? need : 0;
    return *p ? 0 : -3;
}

/* int32 arithmetic that wraps like the reference's plain int32 adds do */
static int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

int rtlws_host_cic(int R, const cmplx_u8* src, int src_len, cmplx_s32* dst, int dst_len,
                   struct cic_delay_line* delay)
{
    struct rtlws_host_ctx* c;
    size_t in_bytes, out_bytes;
    int rc = 0, m;
    int32_t sum_re = 0, sum_im = 0;

    if (dst_len * R != src_len) return -1;        /* reference src/resample.c:18-19 */
    if (src_len <= 0) return 0;                   /* loop body never runs; state unchanged */

    c = rtlws_host_ctx_get();
    if (!c) return -3;
    in_bytes = (size_t)src_len * sizeof(cmplx_u8);
    out_bytes = (size_t)dst_len * sizeof(cmplx_s32);

    pthread_mutex_lock(&g_mu);
    if ((!zero_copy() && (grow_dev(c->eng, &c->d_in, &c->d_in_cap, in_bytes) ||
                          grow_dev(c->eng, &c->d_out, &c->d_out_cap, out_bytes))) ||
        grow_pinned(&c->h_in, &c->h_in_cap, in_bytes) ||
        grow_pinned(&c->h_out, &c->h_out_cap, out_bytes)) {
        rc = -3;
    } else {
        memcpy(c->h_in, src, in_bytes);
        if (zero_copy()
                ? (rtlws_cic_block_sums(c->eng, R, c->h_in, dst_len, c->h_out, NULL) || rtlws_stream_sync(c->eng, NULL))
                : (rtlws_copy_h2d(c->eng, c->d_in, c->h_in, in_bytes, NULL) ||
                   rtlws_cic_block_sums(c->eng, R, c->d_in, dst_len, c->d_out, NULL) ||
                   rtlws_copy_d2h(c->eng, c->h_out, c->d_out, out_bytes, NULL) ||
                   rtlws_stream_sync(c->eng, NULL)))
            rc = -3;
        else
            memcpy(dst, c->h_out, out_bytes);
    }
    pthread_mutex_unlock(&g_mu);
    if (rc) {
        fprintf(stderr, "rtlws: cic_decimate failed on the device: %s\n", rtlws_last_error());
        return rc;
    }

    /* Delay-line bookkeeping.  The kernel returns pure block sums; the
     * reference's first output also carries (integrator - comb) of the
     * incoming state (zero whenever the state came from a previous call), and
     * the state it leaves is the running int32 sum in both fields. */
    for (m = 0; m < dst_len; m++) {
        sum_re = wadd(sum_re, dst[m].p.re);
        sum_im = wadd(sum_im, dst[m].p.im);
    }
    dst[0].p.re = wadd(dst[0].p.re, wsub(delay->integrator_prev_out.p.re, delay->comb_prev_in.p.re));
    dst[0].p.im = wadd(dst[0].p.im, wsub(delay->integrator_prev_out.p.im, delay->comb_prev_in.p.im));
    delay->integrator_prev_out.p.re = wadd(delay->integrator_prev_out.p.re, sum_re);
    delay->integrator_prev_out.p.im = wadd(delay->integrator_prev_out.p.im, sum_im);
    delay->comb_prev_in = delay->integrator_prev_out;
    return 0;
}

/* ---- resample.h entry points ---------------------------------------- */

int cic_decimate(int R, const cmplx_u8* src, int src_len, cmplx_s32* dst, int dst_len,
                 struct cic_delay_line* delay)
{
    return rtlws_host_cic(R, src, src_len, dst, dst_len, delay);
}

/* the delay line becomes the last 10 samples of history + input (src/resample.c:66) */
static void advance_delay(float* delay, const float* input, size_t n_in)
{
    const size_t keep = HALF_BAND_N - 1;
    if (n_in >= keep) {
        memcpy(delay, input + n_in - keep, keep * sizeof(float));
    } else {
        memmove(delay, delay + n_in, (keep - n_in) * sizeof(float));
        memcpy(delay + keep - n_in, input, n_in * sizeof(float));
    }
}

void halfband_decimate(const float* input, float* output, int output_len, float* delay)
{
    struct rtlws_host_ctx* c = rtlws_host_ctx_get();
    const size_t n_in = (size_t)(output_len > 0 ? output_len : 0) * 2;
    const size_t in_bytes = (n_in + (HALF_BAND_N - 1)) * sizeof(float);
    const size_t out_bytes = (size_t)(output_len > 0 ? output_len : 0) * sizeof(float);
    int rc = 0;
    if (output_len <= 0) return;
    if (!c) {
        /* the reference signature is void and there is no CPU path: silence out, the delay
         * line advanced as src/resample.c:66 would, the failure recorded (rtlws_host.h) */
        advance_delay(delay, input, n_in);
        memset(output, 0, out_bytes);
        rtlws_host_fail("halfband_decimate", "no usable HIP device");
        return;
    }

    pthread_mutex_lock(&g_mu);
    if ((!zero_copy() && (grow_dev(c->eng, &c->d_in, &c->d_in_cap, in_bytes) ||
                          grow_dev(c->eng, &c->d_out, &c->d_out_cap, out_bytes))) ||
        grow_pinned(&c->h_in, &c->h_in_cap, in_bytes) ||
        grow_pinned(&c->h_out, &c->h_out_cap, out_bytes)) {
        rc = -3;
    } else {
        float* stage = (float*)c->h_in;
        /* [10 history samples | 2*output_len new samples] */
        memcpy(stage, delay, (HALF_BAND_N - 1) * sizeof(float));
        memcpy(stage + (HALF_BAND_N - 1), input, n_in * sizeof(float));
        if (zero_copy()
                ? (rtlws_halfband(c->eng, stage, (float*)c->h_out, output_len, NULL) || rtlws_stream_sync(c->eng, NULL))
                : (rtlws_copy_h2d(c->eng, c->d_in, stage, in_bytes, NULL) ||
                   rtlws_halfband(c->eng, (const float*)c->d_in, (float*)c->d_out, output_len, NULL) ||
                   rtlws_copy_d2h(c->eng, c->h_out, c->d_out, out_bytes, NULL) ||
                   rtlws_stream_sync(c->eng, NULL)))
            rc = -3;
        else {
            memcpy(output, c->h_out, out_bytes);
            /* reference src/resample.c:66: the delay line becomes the last
             * 10 inputs (taken from history+input so short calls are safe) */
            memcpy(delay, stage + n_in, (HALF_BAND_N - 1) * sizeof(float));
        }
    }
    pthread_mutex_unlock(&g_mu);
    if (rc) {
        /* the reference signature is void: a defined result (silence, delay line advanced
         * from the inputs) and a recorded failure rather than garbage or a dead server */
        advance_delay(delay, input, n_in);
        memset(output, 0, out_bytes);
        rtlws_host_fail("halfband_decimate", rtlws_last_error());
    }
}
