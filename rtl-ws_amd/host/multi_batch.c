/* multi_batch.c -- rtlws_multi.h: a device-resident batch sharded over the GPUs of a node,
 * one host pthread + one engine per shard, no collective.  Plain C over
 * include/rtlws_hip.h. */
#include "rtlws_multi.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct shard {
    int device;
    long first_frame, frames;
    rtlws_engine* eng;
    void* d_in;
    void* d_out;
    void* ev0;
    void* ev1;
};

struct rtlws_multi {
    int n;
    int f64;
    rtlws_spectra_desc desc;
    long nframes;
    size_t frame_bytes, row_bytes;
    struct shard* sh;
};

int rtlws_multi_partition(long nframes, int k_avg, int shards, int g, long* first_frame, long* frame_count)
{
    long rows, lo, hi;
    if (nframes < 0 || k_avg < 1 || shards < 1 || g < 0 || g >= shards) return -1;
    rows = nframes / k_avg;
    lo = (long)(((__int128)rows * g) / shards);
    hi = (long)(((__int128)rows * (g + 1)) / shards);
    if (first_frame) *first_frame = lo * k_avg;
    if (frame_count) *frame_count = (hi - lo) * k_avg;
    return 0;
}

static size_t frame_bytes_of(const rtlws_spectra_desc* d)
{
    const size_t r = d->cic_r > 1 ? (size_t)d->cic_r : 1u;
    const size_t per = d->input == RTLWS_IN_CS32 ? 8u : d->input == RTLWS_IN_RF32 ? 4u : 2u * r;
    return per * (size_t)d->n_fft;
}

static size_t row_bytes_of(const rtlws_spectra_desc* d, int f64)
{
    const size_t e = d->output == RTLWS_OUT_PAYLOAD_U8 ? 1u
                     : (f64 && !(d->flags & RTLWS_FLAG_ROWS_F32)) ? 8u : 4u;
    return e * (size_t)d->n_fft;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

rtlws_multi* rtlws_multi_open(int n_shards, const int* device_ids, const rtlws_spectra_desc* desc,
                              long nframes, int f64)
{
    rtlws_multi* m;
    int g;
    if (!desc || rtlws_spectra_kernel_kind(desc) == 0 || nframes < 0 || n_shards < 0 ||
        (device_ids && n_shards < 1))
        return NULL;
    if (n_shards == 0) n_shards = rtlws_device_count();
    if (n_shards < 1) return NULL;                      /* no device: there is no CPU path */
    m = (rtlws_multi*)calloc(1, sizeof(*m));
    if (!m) return NULL;
    m->sh = (struct shard*)calloc((size_t)n_shards, sizeof(struct shard));
    if (!m->sh) { free(m); return NULL; }
    m->n = n_shards;
    m->f64 = f64 != 0;
    m->desc = *desc;
    m->nframes = nframes - nframes % desc->k_avg;        /* whole K-groups */
    m->frame_bytes = frame_bytes_of(desc);
    m->row_bytes = row_bytes_of(desc, m->f64);
    for (g = 0; g < n_shards; g++) {
        struct shard* s = &m->sh[g];
        s->device = device_ids ? device_ids[g] : g;
        rtlws_multi_partition(nframes, desc->k_avg, n_shards, g, &s->first_frame, &s->frames);
        s->eng = rtlws_engine_create(s->device);
        if (!s->eng) { rtlws_multi_close(m); return NULL; }
        /* an empty shard still owns (1-byte) buffers: its launches are no-ops */
        s->d_in = rtlws_dev_alloc(s->eng, (size_t)s->frames * m->frame_bytes);
        s->d_out = rtlws_dev_alloc(s->eng, (size_t)(s->frames / desc->k_avg) * m->row_bytes);
        s->ev0 = rtlws_event_create();
        s->ev1 = rtlws_event_create();
        if (!s->d_in || !s->d_out || !s->ev0 || !s->ev1 ||
            (m->f64 ? rtlws_engine_prepare_f64(s->eng, desc->n_fft) : rtlws_engine_prepare(s->eng, desc->n_fft))) {
            rtlws_multi_close(m);
            return NULL;
        }
    }
    return m;
}

int rtlws_multi_shards(const rtlws_multi* m) { return m ? m->n : 0; }
long rtlws_multi_frames(const rtlws_multi* m) { return m ? m->nframes : 0; }
size_t rtlws_multi_frame_bytes(const rtlws_multi* m) { return m ? m->frame_bytes : 0; }
size_t rtlws_multi_row_bytes(const rtlws_multi* m) { return m ? m->row_bytes : 0; }

/* ---- one thread per shard ------------------------------------------------ */

enum { JOB_UPLOAD, JOB_RUN, JOB_DOWNLOAD };

/* every shard's thread waits here until all of them exist: they start their work together
 * (or not at all, if a thread could not be created) */
struct gate {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int state;                 /* 0 wait, 1 go, -1 give up */
};

struct job {
    rtlws_multi* m;
    int g, kind, launches, rc;
    const unsigned char* host_in;
    unsigned char* host_out;
    struct gate* start;
    double event_ms, wall_ms;
};

static int launch_once(rtlws_multi* m, struct shard* s)
{
    return m->f64 ? rtlws_spectra_batch_f64(s->eng, &m->desc, s->d_in, s->frames, s->d_out, NULL)
                  : rtlws_spectra_batch(s->eng, &m->desc, s->d_in, s->frames, s->d_out, NULL);
}

static void* job_main(void* arg)
{
    struct job* j = (struct job*)arg;
    rtlws_multi* m = j->m;
    struct shard* s = &m->sh[j->g];
    const size_t in_off = (size_t)s->first_frame * m->frame_bytes;
    const size_t out_off = (size_t)(s->first_frame / m->desc.k_avg) * m->row_bytes;
    const size_t in_bytes = (size_t)s->frames * m->frame_bytes;
    const size_t out_bytes = (size_t)(s->frames / m->desc.k_avg) * m->row_bytes;
    int i;
    double t0;
    j->rc = 0;
    pthread_mutex_lock(&j->start->mu);
    while (j->start->state == 0) pthread_cond_wait(&j->start->cv, &j->start->mu);
    i = j->start->state;
    pthread_mutex_unlock(&j->start->mu);
    if (i < 0) { j->rc = -3; return NULL; }
    t0 = now_ms();
    switch (j->kind) {
    case JOB_UPLOAD:
        if (in_bytes && (rtlws_copy_h2d(s->eng, s->d_in, j->host_in + in_off, in_bytes, NULL) ||
                         rtlws_stream_sync(s->eng, NULL)))
            j->rc = -3;
        break;
    case JOB_DOWNLOAD:
        if (out_bytes && (rtlws_copy_d2h(s->eng, j->host_out + out_off, s->d_out, out_bytes, NULL) ||
                          rtlws_stream_sync(s->eng, NULL)))
            j->rc = -3;
        break;
    default:
        if ((j->rc = rtlws_event_record(s->ev0, s->eng, NULL)) != 0) break;
        for (i = 0; i < j->launches && j->rc == 0; i++) j->rc = launch_once(m, s);
        if (j->rc == 0) j->rc = rtlws_event_record(s->ev1, s->eng, NULL);
        if (j->rc == 0) j->rc = rtlws_stream_sync(s->eng, NULL);
        if (j->rc == 0) j->event_ms = (double)rtlws_event_elapsed_ms(s->ev0, s->ev1);
        break;
    }
    j->wall_ms = now_ms() - t0;
    return NULL;
}

static int run_jobs(rtlws_multi* m, int kind, int launches, const void* host_in, void* host_out,
                    rtlws_multi_shard_stats* stats, double* wall_ms_max)
{
    struct job* jobs;
    pthread_t* th;
    struct gate start;
    int g, rc = 0, started = 0;
    double worst = 0.0;
    if (!m) return -1;
    jobs = (struct job*)calloc((size_t)m->n, sizeof(*jobs));
    th = (pthread_t*)calloc((size_t)m->n, sizeof(*th));
    if (!jobs || !th) { free(jobs); free(th); return -3; }
    pthread_mutex_init(&start.mu, NULL);
    pthread_cond_init(&start.cv, NULL);
    start.state = 0;
    for (g = 0; g < m->n; g++) {
        jobs[g].m = m;
        jobs[g].g = g;
        jobs[g].kind = kind;
        jobs[g].launches = launches;
        jobs[g].host_in = (const unsigned char*)host_in;
        jobs[g].host_out = (unsigned char*)host_out;
        jobs[g].start = &start;
        if (pthread_create(&th[g], NULL, job_main, &jobs[g]) != 0) break;
        started++;
    }
    pthread_mutex_lock(&start.mu);
    start.state = (started == m->n) ? 1 : -1;     /* all shards, or none */
    pthread_cond_broadcast(&start.cv);
    pthread_mutex_unlock(&start.mu);
    for (g = 0; g < started; g++) pthread_join(th[g], NULL);
    pthread_mutex_destroy(&start.mu);
    pthread_cond_destroy(&start.cv);
    if (started < m->n) rc = -3;
    for (g = 0; g < m->n; g++) {
        if (jobs[g].rc && !rc) rc = jobs[g].rc;
        if (jobs[g].wall_ms > worst) worst = jobs[g].wall_ms;
        if (stats) {
            stats[g].device = m->sh[g].device;
            stats[g].first_frame = m->sh[g].first_frame;
            stats[g].frames = m->sh[g].frames;
            stats[g].launches = launches;
            stats[g].event_ms = jobs[g].event_ms;
            stats[g].wall_ms = jobs[g].wall_ms;
            stats[g].rc = jobs[g].rc;
        }
    }
    if (wall_ms_max) *wall_ms_max = worst;
    free(jobs);
    free(th);
    return rc;
}

int rtlws_multi_upload(rtlws_multi* m, const void* host_frames)
{
    if (!m || (!host_frames && m->nframes)) return -1;
    return run_jobs(m, JOB_UPLOAD, 0, host_frames, NULL, NULL, NULL);
}

int rtlws_multi_run(rtlws_multi* m, int launches, rtlws_multi_shard_stats* stats, double* wall_ms_max)
{
    if (!m || launches < 0) return -1;
    return run_jobs(m, JOB_RUN, launches, NULL, NULL, stats, wall_ms_max);
}

int rtlws_multi_download(rtlws_multi* m, void* host_rows)
{
    if (!m || (!host_rows && m->nframes)) return -1;
    return run_jobs(m, JOB_DOWNLOAD, 0, NULL, host_rows, NULL, NULL);
}

void rtlws_multi_close(rtlws_multi* m)
{
    int g;
    if (!m) return;
    for (g = 0; g < m->n; g++) {
        struct shard* s = &m->sh[g];
        if (s->eng) {
            rtlws_stream_sync(s->eng, NULL);
            rtlws_dev_free(s->eng, s->d_in);
            rtlws_dev_free(s->eng, s->d_out);
        }
        rtlws_event_destroy(s->ev0);
        rtlws_event_destroy(s->ev1);
        rtlws_engine_destroy(s->eng);
    }
    free(m->sh);
    free(m);
}
