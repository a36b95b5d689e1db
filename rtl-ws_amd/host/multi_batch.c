/* multi_batch.c -- rtlws_multi.h: a device-resident batch sharded over the GPUs of a node,
 * one host pthread + one engine per shard, no collective.  Plain C over
 * include/rtlws_hip.h.
 *
 * A shard's thread lives as long as the handle.  It pins itself to the CPUs of its device's NUMA node
 * (rtlws_topo.h) BEFORE it creates its engine and allocates -- device buffers, events, and the two pinned
 * staging buffers its uploads and downloads go through -- so that on a two-socket node neither the thread
 * nor the first touch of its pinned memory lands on the far socket.  Commands (upload / run / download)
 * are posted to all shards at once under one mutex and one broadcast: that IS the start gate, and a
 * thread that times a run has long made its first HIP calls (a thread's first call costs milliseconds:
 * a freshly created thread per command, as in round 4, put that inside the timed region). */
#define _GNU_SOURCE
#include "rtlws_multi.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rtlws_topo.h"

enum { CMD_NONE = 0, CMD_INIT, CMD_UPLOAD, CMD_RUN, CMD_DOWNLOAD, CMD_EXIT };
#define STAGE_BYTES ((size_t)8 << 20)      /* pinned staging buffer: 2 per shard */

struct shard {
    rtlws_multi* m;
    int g, device;
    long first_frame, frames;
    rtlws_engine* eng;
    void* d_in;
    void* d_out;
    void* ev0;
    void* ev1;
    void* h_stage[2];          /* pinned, allocated by the shard's own (pinned) thread */
    void* stage_ev[2];
    rtlws_topo_info topo;
    int cpus_pinned;
    pthread_t th;
    int started;
    long seen;                 /* generation of the last command this thread took */
    int rc;
    double event_ms, wall_ms;
    char err[256];
};

struct rtlws_multi {
    int n;
    int f64;
    rtlws_spectra_desc desc;
    long nframes;
    size_t frame_bytes, row_bytes;
    struct shard* sh;
    /* command mailbox */
    pthread_mutex_t mu;
    pthread_cond_t cv_cmd, cv_done;
    long gen;
    int cmd, pending, launches;
    const unsigned char* host_in;
    unsigned char* host_out;
    char err[320];             /* "shard g (device d): ..." of the first failing shard of the last command */
};

/* the text of the last failed rtlws_multi_open (there is no handle to keep it in) */
/* why the calling thread's last rtlws_multi_open failed: per thread, like rtlws_last_error() */
static __thread char g_open_err[352];

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

/* a failing call happened on THIS (worker) thread: its text is in this thread's rtlws_last_error() and
 * nowhere the caller can see -- keep it in the shard */
static int fail(struct shard* s, const char* what, int rc)
{
    snprintf(s->err, sizeof s->err, "%s: %s", what, rtlws_last_error());
    return rc ? rc : -3;
}

static int launch_once(rtlws_multi* m, struct shard* s)
{
    return m->f64 ? rtlws_spectra_batch_f64(s->eng, &m->desc, s->d_in, s->frames, s->d_out, NULL)
                  : rtlws_spectra_batch(s->eng, &m->desc, s->d_in, s->frames, s->d_out, NULL);
}

/* pin, then create everything this shard owns -- in that order */
static int shard_init(struct shard* s)
{
    rtlws_multi* m = s->m;
    int k;
    rtlws_topo_describe(s->device, NULL, NULL, &s->topo);
    s->cpus_pinned = rtlws_topo_pin_thread(&s->topo);
    if (s->cpus_pinned < 0) s->cpus_pinned = 0;            /* could not pin: run where we are */
    s->eng = rtlws_engine_create(s->device);
    if (!s->eng) return fail(s, "rtlws_engine_create", -3);
    /* an empty shard still owns (1-byte) buffers: its launches are no-ops */
    s->d_in = rtlws_dev_alloc(s->eng, (size_t)s->frames * m->frame_bytes);
    s->d_out = rtlws_dev_alloc(s->eng, (size_t)(s->frames / m->desc.k_avg) * m->row_bytes);
    s->ev0 = rtlws_event_create();
    s->ev1 = rtlws_event_create();
    if (!s->d_in || !s->d_out || !s->ev0 || !s->ev1) return fail(s, "device buffers / events", -3);
    for (k = 0; k < 2; k++) {
        s->h_stage[k] = rtlws_pinned_alloc(STAGE_BYTES);
        s->stage_ev[k] = rtlws_event_create();
        if (!s->h_stage[k] || !s->stage_ev[k]) return fail(s, "pinned staging", -3);
        memset(s->h_stage[k], 0, STAGE_BYTES);             /* first touch here, on the pinned thread */
    }
    if (m->f64 ? rtlws_engine_prepare_f64(s->eng, m->desc.n_fft) : rtlws_engine_prepare(s->eng, m->desc.n_fft))
        return fail(s, "rtlws_engine_prepare", -3);
    return 0;
}

static void shard_release(struct shard* s)
{
    int k;
    if (s->eng) {
        rtlws_stream_sync(s->eng, NULL);
        rtlws_dev_free(s->eng, s->d_in);
        rtlws_dev_free(s->eng, s->d_out);
    }
    for (k = 0; k < 2; k++) {
        rtlws_pinned_free(s->h_stage[k]);
        rtlws_event_destroy(s->stage_ev[k]);
        s->h_stage[k] = s->stage_ev[k] = NULL;
    }
    rtlws_event_destroy(s->ev0);
    rtlws_event_destroy(s->ev1);
    rtlws_engine_destroy(s->eng);
    s->eng = NULL;
    s->d_in = s->d_out = s->ev0 = s->ev1 = NULL;
}

/* the caller's (pageable) frames -> pinned staging -> device, two buffers in turn */
static int shard_upload(struct shard* s, const unsigned char* host)
{
    rtlws_multi* m = s->m;
    const size_t total = (size_t)s->frames * m->frame_bytes;
    const unsigned char* src = host + (size_t)s->first_frame * m->frame_bytes;
    size_t off = 0;
    int k = 0, used[2] = {0, 0};
    while (off < total) {
        const size_t n = total - off < STAGE_BYTES ? total - off : STAGE_BYTES;
        if (used[k] && rtlws_event_sync(s->stage_ev[k])) return fail(s, "upload: event", -3);
        memcpy(s->h_stage[k], src + off, n);
        if (rtlws_copy_h2d(s->eng, (unsigned char*)s->d_in + off, s->h_stage[k], n, NULL) ||
            rtlws_event_record(s->stage_ev[k], s->eng, NULL))
            return fail(s, "upload: copy", -3);
        used[k] = 1;
        off += n;
        k ^= 1;
    }
    if (rtlws_stream_sync(s->eng, NULL)) return fail(s, "upload: sync", -3);
    return 0;
}

/* device -> pinned staging -> the caller's rows; the copy of piece i+1 runs under the memcpy of piece i */
static int shard_download(struct shard* s, unsigned char* host)
{
    rtlws_multi* m = s->m;
    const size_t total = (size_t)(s->frames / m->desc.k_avg) * m->row_bytes;
    unsigned char* dst = host + (size_t)(s->first_frame / m->desc.k_avg) * m->row_bytes;
    size_t issued = 0, done = 0;
    size_t len[2] = {0, 0};
    int ki = 0, kd = 0, inflight = 0;
    while (done < total) {
        while (inflight < 2 && issued < total) {
            const size_t n = total - issued < STAGE_BYTES ? total - issued : STAGE_BYTES;
            if (rtlws_copy_d2h(s->eng, s->h_stage[ki], (const unsigned char*)s->d_out + issued, n, NULL) ||
                rtlws_event_record(s->stage_ev[ki], s->eng, NULL))
                return fail(s, "download: copy", -3);
            len[ki] = n;
            issued += n;
            ki ^= 1;
            ++inflight;
        }
        if (rtlws_event_sync(s->stage_ev[kd])) return fail(s, "download: event", -3);
        memcpy(dst + done, s->h_stage[kd], len[kd]);
        done += len[kd];
        kd ^= 1;
        --inflight;
    }
    return 0;
}

static int shard_run(struct shard* s, int launches)
{
    rtlws_multi* m = s->m;
    int i, rc;
    const double t0 = now_ms();
    s->event_ms = 0.0;
    rc = rtlws_event_record(s->ev0, s->eng, NULL);
    for (i = 0; i < launches && rc == 0; i++) rc = launch_once(m, s);
    if (rc == 0) rc = rtlws_event_record(s->ev1, s->eng, NULL);
    if (rc == 0) rc = rtlws_stream_sync(s->eng, NULL);
    if (rc == 0) s->event_ms = (double)rtlws_event_elapsed_ms(s->ev0, s->ev1);
    s->wall_ms = now_ms() - t0;
    return rc ? fail(s, "run", rc) : 0;
}

static void* shard_main(void* arg)
{
    struct shard* s = (struct shard*)arg;
    rtlws_multi* m = s->m;
    for (;;) {
        int cmd, launches, rc = 0;
        const unsigned char* in;
        unsigned char* out;
        pthread_mutex_lock(&m->mu);
        while (m->gen == s->seen) pthread_cond_wait(&m->cv_cmd, &m->mu);
        s->seen = m->gen;
        cmd = m->cmd;
        launches = m->launches;
        in = m->host_in;
        out = m->host_out;
        pthread_mutex_unlock(&m->mu);

        s->err[0] = 0;
        s->wall_ms = 0.0;
        switch (cmd) {
        case CMD_INIT: rc = shard_init(s); break;
        case CMD_UPLOAD: { const double t0 = now_ms(); rc = s->frames ? shard_upload(s, in) : 0; s->wall_ms = now_ms() - t0; } break;
        case CMD_DOWNLOAD: { const double t0 = now_ms(); rc = s->frames ? shard_download(s, out) : 0; s->wall_ms = now_ms() - t0; } break;
        case CMD_RUN: rc = shard_run(s, launches); break;
        default: shard_release(s); break;          /* CMD_EXIT: what this thread created, it frees */
        }
        s->rc = rc;

        pthread_mutex_lock(&m->mu);
        if (rc && !m->err[0]) snprintf(m->err, sizeof m->err, "shard %d (device %d): %s", s->g, s->device, s->err);
        if (--m->pending == 0) pthread_cond_signal(&m->cv_done);
        pthread_mutex_unlock(&m->mu);
        if (cmd == CMD_EXIT) return NULL;
    }
}

/* post one command to the first `count` shards' threads and wait for all of them; the first failing shard's code */
static int post(rtlws_multi* m, int count, int cmd, int launches, const void* host_in, void* host_out)
{
    int g, rc = 0;
    if (count < 1) return 0;
    pthread_mutex_lock(&m->mu);
    m->err[0] = 0;
    m->cmd = cmd;
    m->launches = launches;
    m->host_in = (const unsigned char*)host_in;
    m->host_out = (unsigned char*)host_out;
    m->pending = count;
    m->gen++;
    pthread_cond_broadcast(&m->cv_cmd);           /* every shard starts here, together */
    while (m->pending) pthread_cond_wait(&m->cv_done, &m->mu);
    for (g = 0; g < count; g++)
        if (m->sh[g].rc && !rc) rc = m->sh[g].rc;
    pthread_mutex_unlock(&m->mu);
    return rc;
}

rtlws_multi* rtlws_multi_open(int n_shards, const int* device_ids, const rtlws_spectra_desc* desc,
                              long nframes, int f64)
{
    rtlws_multi* m;
    int g, started = 0;
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
    pthread_mutex_init(&m->mu, NULL);
    pthread_cond_init(&m->cv_cmd, NULL);
    pthread_cond_init(&m->cv_done, NULL);
    for (g = 0; g < n_shards; g++) {
        struct shard* s = &m->sh[g];
        s->m = m;
        s->g = g;
        s->device = device_ids ? device_ids[g] : g;
        s->topo.numa_node = -1;
        rtlws_multi_partition(nframes, desc->k_avg, n_shards, g, &s->first_frame, &s->frames);
    }
    for (g = 0; g < n_shards; g++) {
        if (pthread_create(&m->sh[g].th, NULL, shard_main, &m->sh[g]) != 0) break;
        m->sh[g].started = 1;
        started++;
    }
    /* every shard pins itself and builds its own engine, buffers and tables, concurrently */
    if (started < n_shards || post(m, started, CMD_INIT, 0, NULL, NULL) != 0) {
        if (started < n_shards) snprintf(g_open_err, sizeof g_open_err, "rtlws_multi_open: could not create %d threads", n_shards);
        else snprintf(g_open_err, sizeof g_open_err, "rtlws_multi_open: %s", m->err);
        m->n = started;                                  /* only these have a thread to tell */
        rtlws_multi_close(m);
        return NULL;
    }
    return m;
}

int rtlws_multi_shards(const rtlws_multi* m) { return m ? m->n : 0; }
long rtlws_multi_frames(const rtlws_multi* m) { return m ? m->nframes : 0; }
size_t rtlws_multi_frame_bytes(const rtlws_multi* m) { return m ? m->frame_bytes : 0; }
size_t rtlws_multi_row_bytes(const rtlws_multi* m) { return m ? m->row_bytes : 0; }

const char* rtlws_multi_error(const rtlws_multi* m) { return m ? m->err : g_open_err; }

int rtlws_multi_shard_topology(const rtlws_multi* m, int g, rtlws_topo_info* out, int* cpus_pinned)
{
    if (!m || g < 0 || g >= m->n) return -1;
    if (out) *out = m->sh[g].topo;
    if (cpus_pinned) *cpus_pinned = m->sh[g].cpus_pinned;
    return 0;
}

int rtlws_multi_upload(rtlws_multi* m, const void* host_frames)
{
    if (!m || (!host_frames && m->nframes)) return -1;
    return post(m, m->n, CMD_UPLOAD, 0, host_frames, NULL);
}

int rtlws_multi_run(rtlws_multi* m, int launches, rtlws_multi_shard_stats* stats, double* wall_ms_max)
{
    int g, rc;
    double worst = 0.0;
    if (!m || launches < 0) return -1;
    rc = post(m, m->n, CMD_RUN, launches, NULL, NULL);
    for (g = 0; g < m->n; g++) {
        const struct shard* s = &m->sh[g];
        if (s->wall_ms > worst) worst = s->wall_ms;
        if (stats) {
            stats[g].device = s->device;
            stats[g].first_frame = s->first_frame;
            stats[g].frames = s->frames;
            stats[g].launches = launches;
            stats[g].event_ms = s->event_ms;
            stats[g].wall_ms = s->wall_ms;
            stats[g].rc = s->rc;
        }
    }
    if (wall_ms_max) *wall_ms_max = worst;
    return rc;
}

int rtlws_multi_download(rtlws_multi* m, void* host_rows)
{
    if (!m || (!host_rows && m->nframes)) return -1;
    return post(m, m->n, CMD_DOWNLOAD, 0, NULL, host_rows);
}

void rtlws_multi_close(rtlws_multi* m)
{
    int g;
    if (!m) return;
    post(m, m->n, CMD_EXIT, 0, NULL, NULL);            /* each thread frees what it created, then leaves */
    for (g = 0; g < m->n; g++)
        if (m->sh[g].started) pthread_join(m->sh[g].th, NULL);
    pthread_mutex_destroy(&m->mu);
    pthread_cond_destroy(&m->cv_cmd);
    pthread_cond_destroy(&m->cv_done);
    free(m->sh);
    free(m);
}
