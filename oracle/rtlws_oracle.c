/*
 * rtlws_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see rtlws_oracle.h).
 *
 * f64 restatement of the reference hot path with its own FFT.  Every function
 * names the reference lines it follows (paths under /root/reference).
 * Spectrum stage / dB payload: parity unpinned by reference execution (FFTW3
 * absent here); CIC / half-band / re-blocker: pinned against oracle/_ref.
 */
#include "rtlws_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* DFT                                                                 */
/* ------------------------------------------------------------------ */

struct dft_plan {
    int N;
    int pow2;
    double* tw;       /* non-pow2: N twiddles exp(-2 pi i k / N), interleaved */
    double* st_re;    /* pow2: per-stage twiddles, stage with half-length h at offset h-1 */
    double* st_im;
    int* bitrev;      /* pow2 only */
};

#define MAX_PLANS 32
static struct dft_plan* g_plans[MAX_PLANS];
static int g_nplans = 0;
static pthread_mutex_t g_plan_mutex = PTHREAD_MUTEX_INITIALIZER;

static struct dft_plan* plan_build(int N)
{
    struct dft_plan* p = (struct dft_plan*)calloc(1, sizeof(*p));
    const long double two_pi = 6.283185307179586476925286766559005768L;
    int k;
    p->N = N;
    p->pow2 = (N >= 2) && ((N & (N - 1)) == 0);
    if (p->pow2) {
        int bits = 0, i, half;
        while ((1 << bits) < N) bits++;
        /* stage of butterfly span len = 2*half uses W_len^j, j < half; stored
         * contiguously per stage so the inner loops are unit-stride */
        p->st_re = (double*)malloc(sizeof(double) * N);
        p->st_im = (double*)malloc(sizeof(double) * N);
        for (half = 1; half < N; half <<= 1)
            for (k = 0; k < half; k++) {
                long double a = -two_pi * (long double)k / (long double)(2 * half);
                p->st_re[half - 1 + k] = (double)cosl(a);
                p->st_im[half - 1 + k] = (double)sinl(a);
            }
        p->bitrev = (int*)malloc(sizeof(int) * N);
        for (i = 0; i < N; i++) {
            int r = 0, b;
            for (b = 0; b < bits; b++)
                if (i & (1 << b)) r |= 1 << (bits - 1 - b);
            p->bitrev[i] = r;
        }
    } else {
        p->tw = (double*)malloc(sizeof(double) * 2 * (N > 0 ? N : 1));
        for (k = 0; k < N; k++) {
            long double a = -two_pi * (long double)k / (long double)N;
            p->tw[2 * k] = (double)cosl(a);
            p->tw[2 * k + 1] = (double)sinl(a);
        }
    }
    return p;
}

static const struct dft_plan* plan_get(int N)
{
    int i;
    struct dft_plan* p = NULL;
    pthread_mutex_lock(&g_plan_mutex);
    for (i = 0; i < g_nplans; i++)
        if (g_plans[i]->N == N) { p = g_plans[i]; break; }
    if (!p) {
        p = plan_build(N);
        if (g_nplans < MAX_PLANS) g_plans[g_nplans++] = p;   /* else leaked: test code */
    }
    pthread_mutex_unlock(&g_plan_mutex);
    return p;
}

/* Radix-2 decimation-in-time on split re/im arrays that already hold the
 * input in bit-reversed order; in place; f64 throughout. */
static void dft_pow2_split(const struct dft_plan* p, double* restrict re, double* restrict im)
{
    const int N = p->N;
    int half, base, j;
    for (half = 1; half < N; half <<= 1) {
        const double* restrict wr = p->st_re + (half - 1);
        const double* restrict wi = p->st_im + (half - 1);
        const int len = 2 * half;
        for (base = 0; base < N; base += len) {
            double* restrict ar = re + base;
            double* restrict ai = im + base;
            double* restrict br = re + base + half;
            double* restrict bi = im + base + half;
            for (j = 0; j < half; j++) {
                const double tr = br[j] * wr[j] - bi[j] * wi[j];
                const double ti = br[j] * wi[j] + bi[j] * wr[j];
                br[j] = ar[j] - tr;
                bi[j] = ai[j] - ti;
                ar[j] += tr;
                ai[j] += ti;
            }
        }
    }
}

static void dft_pow2(const struct dft_plan* p, const double* in, double* out)
{
    const int N = p->N;
    double* re = (double*)malloc(sizeof(double) * 2 * N);
    double* im = re + N;
    int i;
    for (i = 0; i < N; i++) {
        const int r = p->bitrev[i];
        re[r] = in[2 * i];
        im[r] = in[2 * i + 1];
    }
    dft_pow2_split(p, re, im);
    for (i = 0; i < N; i++) { out[2 * i] = re[i]; out[2 * i + 1] = im[i]; }
    free(re);
}

void orc_dft_direct(int N, const double* in, double* out)
{
    const long double two_pi = 6.283185307179586476925286766559005768L;
    int k, n;
    for (k = 0; k < N; k++) {
        long double sr = 0.0L, si = 0.0L;
        for (n = 0; n < N; n++) {
            /* reduce n*k mod N before the angle so the argument stays small */
            long long m = ((long long)n * (long long)k) % (long long)N;
            long double a = -two_pi * (long double)m / (long double)N;
            long double c = cosl(a), s = sinl(a);
            sr += (long double)in[2 * n] * c - (long double)in[2 * n + 1] * s;
            si += (long double)in[2 * n] * s + (long double)in[2 * n + 1] * c;
        }
        out[2 * k] = (double)sr;
        out[2 * k + 1] = (double)si;
    }
}

void orc_dft_forward(int N, const double* in, double* out)
{
    if (N <= 0) return;
    if (N == 1) { out[0] = in[0]; out[1] = in[1]; return; }
    if ((N & (N - 1)) == 0)
        dft_pow2(plan_get(N), in, out);
    else
        orc_dft_direct(N, in, out);
}

/* ------------------------------------------------------------------ */
/* spectrum.c                                                          */
/* ------------------------------------------------------------------ */

/* src/spectrum.c:15-35: after the forward DFT of `in`, walk the output slots
 * in increasing i; bin idx=(N/2+i)%len; idx>0 adds |X[idx]|^2, idx==0 adds the
 * value slot i-1 holds at that moment. */
static void accumulate_shifted_power(int N, int len, const double* X, double* ps)
{
    const int offset = N / 2;
    int i;
    for (i = 0; i < len; i++) {
        const int idx = (offset + i) % len;
        if (idx > 0)
            ps[i] += X[2 * idx] * X[2 * idx] + X[2 * idx + 1] * X[2 * idx + 1];
        else
            ps[i] += ps[i - 1];   /* i == N/2 here; N==1 would index -1 like the reference */
    }
}

static int spectrum_frame(int N, double* in, const double* window, double* ps, int len)
{
    double* X;
    int i;
    if (window)
        for (i = 0; i < N; i++) { in[2 * i] *= window[i]; in[2 * i + 1] *= window[i]; }
    X = (double*)malloc(sizeof(double) * 2 * N);
    orc_dft_forward(N, in, X);
    accumulate_shifted_power(N, len, X, ps);
    free(X);
    return 0;
}

int orc_spectrum_add_cmplx_u8(int N, const uint8_t* src, const double* window,
                              double* ps, int len)
{
    double* in;
    int i;
    if (len != N) return -1;                       /* src/spectrum.c:51-52 */
    in = (double*)malloc(sizeof(double) * 2 * N);
    for (i = 0; i < N; i++) {                      /* src/spectrum.c:54-58 */
        in[2 * i] = (((double)src[2 * i]) - 128) / 128;
        in[2 * i + 1] = (((double)src[2 * i + 1]) - 128) / 128;
    }
    spectrum_frame(N, in, window, ps, len);
    free(in);
    return 0;
}

int orc_spectrum_add_cmplx_s32(int N, const int32_t* src, const double* window,
                               double* ps, int len)
{
    double* in;
    int i;
    if (len != N) return -1;                       /* src/spectrum.c:69-70 */
    in = (double*)malloc(sizeof(double) * 2 * N);
    for (i = 0; i < N; i++) {                      /* src/spectrum.c:72-76 */
        in[2 * i] = ((double)src[2 * i]) / 128;
        in[2 * i + 1] = ((double)src[2 * i + 1]) / 128;
    }
    spectrum_frame(N, in, window, ps, len);
    free(in);
    return 0;
}

int orc_spectrum_add_real_f32(int N, const float* src, const double* window,
                              double* ps, int len)
{
    double* in;
    int i;
    if (len != N) return -1;                       /* src/spectrum.c:87-88 */
    in = (double*)malloc(sizeof(double) * 2 * N);
    for (i = 0; i < N; i++) {                      /* src/spectrum.c:90-94 */
        in[2 * i] = src[i];
        in[2 * i + 1] = 0;
    }
    spectrum_frame(N, in, window, ps, len);
    free(in);
    return 0;
}

/* ---- batch helpers (harness around the per-frame calls) ------------- */

struct batch_job {
    int N, K, R;
    long row0, row1;
    const uint8_t* src;
    const double* window;
    double* out;
};

/* One power-of-two frame with thread-owned scratch: same arithmetic as
 * orc_spectrum_add_cmplx_u8 / _s32 (conversion src/spectrum.c:54-58 / :72-76,
 * window, forward DFT, src/spectrum.c:23-34), minus the per-call allocations. */
static void fast_frame(const struct dft_plan* p, const uint8_t* u8, const int32_t* s32,
                       const double* window, double* re, double* im, double* ps)
{
    const int N = p->N, offset = N / 2;
    int i;
    for (i = 0; i < N; i++) {
        const int r = p->bitrev[i];
        double a, b;
        /* x / 128 written as x * 2^-7: the same f64 value, no divider */
        if (u8) {
            a = (((double)u8[2 * i]) - 128) * 0.0078125;
            b = (((double)u8[2 * i + 1]) - 128) * 0.0078125;
        } else {
            a = ((double)s32[2 * i]) * 0.0078125;
            b = ((double)s32[2 * i + 1]) * 0.0078125;
        }
        if (window) { a *= window[i]; b *= window[i]; }
        re[r] = a;
        im[r] = b;
    }
    dft_pow2_split(p, re, im);
    /* slot i shows bin (offset + i) % N: first the upper half of the bins, then
     * the slot of bin 0 (running left neighbour), then bins 1 .. offset-1 */
    for (i = 0; i < N - offset; i++)
        ps[i] += re[offset + i] * re[offset + i] + im[offset + i] * im[offset + i];
    ps[N - offset] += ps[N - offset - 1];
    for (i = N - offset + 1; i < N; i++) {
        const int idx = i - (N - offset);
        ps[i] += re[idx] * re[idx] + im[idx] * im[idx];
    }
}

static void* batch_worker(void* arg)
{
    struct batch_job* j = (struct batch_job*)arg;
    const int N = j->N, K = j->K, R = j->R;
    const int pow2 = (N >= 2) && ((N & (N - 1)) == 0);
    const struct dft_plan* plan = pow2 ? plan_get(N) : NULL;
    int32_t* dec = NULL;
    double* scratch = pow2 ? (double*)malloc(sizeof(double) * 2 * N) : NULL;
    long row;
    if (R > 1) dec = (int32_t*)malloc(sizeof(int32_t) * 2 * N);
    for (row = j->row0; row < j->row1; row++) {
        double* ps = j->out + (size_t)row * N;
        int k;
        memset(ps, 0, sizeof(double) * N);          /* src/cbb_main.c:50 */
        for (k = 0; k < K; k++) {
            const size_t frame = (size_t)row * K + k;
            if (R > 1) {
                int32_t st[4] = {0, 0, 0, 0};
                orc_cic_decimate(R, j->src + frame * (size_t)N * R * 2, N * R, dec, N, st);
                if (pow2) fast_frame(plan, NULL, dec, j->window, scratch, scratch + N, ps);
                else orc_spectrum_add_cmplx_s32(N, dec, j->window, ps, N);
            } else {
                const uint8_t* src = j->src + frame * (size_t)N * 2;
                if (pow2) fast_frame(plan, src, NULL, j->window, scratch, scratch + N, ps);
                else orc_spectrum_add_cmplx_u8(N, src, j->window, ps, N);
            }
        }
    }
    free(dec);
    free(scratch);
    return NULL;
}

static int run_batch(int N, int K, int R, long nframes, const uint8_t* src,
                     const double* window, double* out, int nthreads)
{
    long rows, per, r0 = 0;
    int t, used = 0;
    pthread_t* th;
    struct batch_job* jobs;
    if (N <= 0 || K <= 0 || nframes < 0 || nframes % K) return -1;
    rows = nframes / K;
    if (nthreads < 1) nthreads = 1;
    if ((long)nthreads > rows) nthreads = rows > 0 ? (int)rows : 1;
    (void)plan_get(N);   /* build the plan once, outside the threads */
    th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    jobs = (struct batch_job*)malloc(sizeof(struct batch_job) * nthreads);
    per = (rows + nthreads - 1) / nthreads;
    for (t = 0; t < nthreads && r0 < rows; t++) {
        jobs[t].N = N; jobs[t].K = K; jobs[t].R = R;
        jobs[t].row0 = r0;
        jobs[t].row1 = (r0 + per < rows) ? r0 + per : rows;
        jobs[t].src = src; jobs[t].window = window; jobs[t].out = out;
        r0 = jobs[t].row1;
        used++;
    }
    if (used == 1) {
        batch_worker(&jobs[0]);
    } else {
        for (t = 0; t < used; t++) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
        for (t = 0; t < used; t++) pthread_join(th[t], NULL);
    }
    free(th);
    free(jobs);
    return 0;
}

int orc_batch_spectra_u8(int N, int K, long nframes, const uint8_t* src,
                         const double* window, double* out, int nthreads)
{
    return run_batch(N, K, 1, nframes, src, window, out, nthreads);
}

int orc_batch_spectra_cic_u8(int N, int K, int R, long nframes, const uint8_t* src,
                             const double* window, double* out, int nthreads)
{
    if (R < 1) return -1;
    return run_batch(N, K, R, nframes, src, window, out, nthreads);
}

/* ------------------------------------------------------------------ */
/* resample.c                                                          */
/* ------------------------------------------------------------------ */

/* int32 add/sub with two's-complement wrap (the reference relies on it
 * implicitly through plain int32 arithmetic). */
static int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wrap_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

int orc_cic_decimate(int R, const uint8_t* src, int src_len, int32_t* dst, int dst_len,
                     int32_t state[4])
{
    int32_t int_re = state[0], int_im = state[1];       /* src/resample.c:15 */
    int32_t comb_re = state[2], comb_im = state[3];     /* src/resample.c:16 */
    int src_idx, dst_idx = 0;

    if (dst_len * R != src_len) return -1;              /* src/resample.c:18-19 */

    for (src_idx = 0; src_idx < src_len; src_idx++) {
        /* integrator y(n) = y(n-1) + (x(n) - 128)         src/resample.c:24-25 */
        int_re = wrap_add(int_re, (int32_t)src[2 * src_idx] - 128);
        int_im = wrap_add(int_im, (int32_t)src[2 * src_idx + 1] - 128);
        if (((src_idx + 1) % R) == 0) {                 /* src/resample.c:28 */
            if (dst_idx >= dst_len) return -2;          /* src/resample.c:31-34 */
            /* comb y(n) = x(n) - x(n-1)                   src/resample.c:35-36 */
            dst[2 * dst_idx] = wrap_sub(int_re, comb_re);
            dst[2 * dst_idx + 1] = wrap_sub(int_im, comb_im);
            comb_re = int_re;
            comb_im = int_im;
            dst_idx++;
        }
    }
    state[0] = int_re; state[1] = int_im;               /* src/resample.c:42 */
    state[2] = comb_re; state[3] = comb_im;             /* src/resample.c:43 */
    return 0;
}

static const float k_halfband[11] = {                   /* src/resample.c:4 */
    0.01824f, 0.0f, -0.11614f, 0.0f, 0.34790f, 0.5f,
    0.34790f, 0.0f, -0.11614f, 0.0f, 0.01824f};

void orc_halfband_decimate(const float* input, float* output, int output_len, float* delay)
{
    int n, k;
    for (n = 0; n < output_len; n++) {
        int idx = 2 * n - 11 / 2;                       /* src/resample.c:56 */
        float acc = k_halfband[5] * (idx >= 0 ? input[idx] : delay[10 + idx]);
        for (k = 0; k < 11; k += 2) {                   /* src/resample.c:60-64 */
            idx = 2 * n - k;
            acc = acc + k_halfband[k] * (idx >= 0 ? input[idx] : delay[10 + idx]);
        }
        output[n] = acc;
    }
    /* src/resample.c:66: keep the last 10 inputs */
    memmove(delay, &input[2 * output_len - 10], 10 * sizeof(float));
}

/* ------------------------------------------------------------------ */
/* audio front end: common_sp.h atan2_approx + audio_main.c demodulator */
/* ------------------------------------------------------------------ */

float orc_atan2_approx(float y, float x)
{
    const float pi_by_2 = (float)(M_PI / 2);                 /* src/common_sp.h:43 */
    float at, z;
    if (x == 0) {                                            /* :47-56 */
        if (y > 0.0f) return pi_by_2;
        if (y == 0) return 0;
        return -pi_by_2;
    }
    z = y / x;                                               /* :57 */
    if (fabs(z) < 1.0f) {                                    /* :58 */
        at = z / (1.0f + 0.28f * z * z);                     /* :60 */
        if (x < 0) {                                         /* :61-67: the +-M_PI is a double add */
            if (y < 0.0f) return (float)((double)at - M_PI);
            return (float)((double)at + M_PI);
        }
    } else {
        at = pi_by_2 - z / (z * z + 0.28f);                  /* :71 */
        if (y < 0.0f) return (float)((double)at - M_PI);     /* :72-73 */
    }
    return at;
}

void orc_fm_demod(const int32_t* iq, int len, float* prev_phase, float* out)
{
    float prev = *prev_phase;
    int i;
    for (i = 0; i < len; i++) {                              /* src/audio_main.c:110-131 */
        const float ph = orc_atan2_approx((float)iq[2 * i + 1], (float)iq[2 * i]);
        float d = ph - prev;
        prev = ph;
        if (d > 1.0f) d = 1;                                 /* hard limit, scale == 1 */
        else if (d < -1.0f) d = -1;
        else d = d / 1.0f;
        out[i] = d;
    }
    *prev_phase = prev;
}

void orc_audio_block(const int32_t* iq, int len, float state[21], float* audio_out)
{
    float* demod = (float*)malloc(sizeof(float) * (size_t)(len > 0 ? len : 1));
    float* work = (float*)malloc(sizeof(float) * (size_t)(len / 2 > 0 ? len / 2 : 1));
    orc_fm_demod(iq, len, &state[0], demod);
    orc_halfband_decimate(demod, work, len / 2, &state[1]);              /* src/audio_main.c:133 */
    orc_halfband_decimate(work, audio_out, (len / 2) / 2, &state[11]);   /* :139 */
    free(demod);
    free(work);
}

/* ------------------------------------------------------------------ */
/* rf_decimator.c                                                      */
/* ------------------------------------------------------------------ */

struct orc_rfdec {
    double sample_rate;
    int down_factor;
    uint8_t* input;          /* input_len cmplx_u8 */
    int input_len;
    int surplus;
    int32_t* resampled;      /* resampled_len cmplx_s32 */
    int resampled_len;
    int32_t delay[4];
};

struct orc_rfdec* orc_rfdec_new(void)
{
    return (struct orc_rfdec*)calloc(1, sizeof(struct orc_rfdec));
}

int orc_rfdec_set_parameters(struct orc_rfdec* d, double sample_rate, int down_factor)
{
    if (!(sample_rate > 0 && down_factor > 0)) return -1;     /* src/rf_decimator.c:58,73 */
    if (fabs(d->sample_rate - sample_rate) > 0.0001 || d->down_factor != down_factor) {
        d->sample_rate = sample_rate;                          /* :63-66 */
        d->down_factor = down_factor;
        d->resampled_len = (int)((d->sample_rate / d->down_factor) * 100 / 1000);
        d->input_len = d->resampled_len * d->down_factor;
        d->resampled = (int32_t*)realloc(d->resampled, sizeof(int32_t) * 2 * (size_t)d->resampled_len + 8);
        d->input = (uint8_t*)realloc(d->input, 2 * (size_t)d->input_len + 8);
        d->surplus = 0;                                        /* :71 */
    }
    return 0;
}

int orc_rfdec_decimate(struct orc_rfdec* d, const uint8_t* iq, int len,
                       orc_rfdec_cb cb, void* user)
{
    int current = 0, remaining = len;
    int block;
    if (d->resampled == NULL || d->input == NULL) return -1;   /* :90-91 */
    block = d->input_len - d->surplus;                         /* :88 */
    while (remaining >= block) {                               /* :93 */
        memcpy(d->input + 2 * (size_t)d->surplus, iq + 2 * (size_t)current, 2 * (size_t)block);
        remaining -= block;
        current += block;
        if (orc_cic_decimate(d->down_factor, d->input, d->input_len, d->resampled,
                             d->resampled_len, d->delay))
            return -2;                                         /* :99-103 */
        if (cb) cb(d->resampled, d->resampled_len, user);      /* :105 */
        d->surplus = 0;
        block = d->input_len;
    }
    if (remaining > 0) {                                       /* :111-115 */
        memcpy(d->input + 2 * (size_t)d->surplus, iq + 2 * (size_t)(len - remaining),
               2 * (size_t)remaining);
        d->surplus += remaining;
    }
    return 0;
}

int orc_rfdec_input_len(const struct orc_rfdec* d) { return d->input_len; }
int orc_rfdec_resampled_len(const struct orc_rfdec* d) { return d->resampled_len; }

void orc_rfdec_free(struct orc_rfdec* d)
{
    if (!d) return;
    free(d->input);
    free(d->resampled);
    free(d);
}

/* ------------------------------------------------------------------ */
/* cbb_main.c                                                          */
/* ------------------------------------------------------------------ */

int orc_spectrum_payload(int N, const double* ps, int count, int spectrum_gain_db, uint8_t* buf)
{
    /* src/cbb_main.c:112: integer division inside pow() -> 10 dB steps */
    const double gain = pow(10, spectrum_gain_db / 10);
    int idx, len = 0;
    if (count > 0) {                                           /* :121 */
        for (idx = 0; idx < N; idx++) {
            const double d = 10 * log10(fabs(gain * ps[idx] / count));   /* :125 */
            int m;
            /* (int)d truncates toward zero; -inf / NaN convert to INT_MIN on
             * x86 and are clamped to 0 by :126, so fold that in explicitly. */
            if (d >= 0) m = (d <= 255) ? (int)d : 255; else m = 0;      /* :125-127 */
            buf[len++] = (uint8_t)m;                                   /* :128 */
        }
    }
    return len;
}

int orc_estimate_spectrum(const uint8_t* iq, int len, double* ps)
{
    int blocks = len / 1024, i;                                /* src/cbb_main.c:44 */
    blocks = blocks <= 6 ? blocks : 6;                         /* :49 */
    memset(ps, 0, 1024 * sizeof(double));                      /* :50 */
    for (i = 0; i < blocks; i++)                               /* :52-59 */
        orc_spectrum_add_cmplx_u8(1024, iq + 2 * (size_t)i * 1024, NULL, ps, 1024);
    return blocks;
}

void orc_mean_db(int N, const double* ps, int count, double* db)
{
    int i;
    for (i = 0; i < N; i++) db[i] = 10.0 * log10(ps[i] / (double)count);
}
