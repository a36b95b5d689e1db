/* stream_gpu.c -- rtlws_stream.h: pinned ring -> async H2D -> fused kernel ->
 * async D2H, several chunks in flight, in-order completion on a worker thread.
 * Plain C over include/rtlws_hip.h.
 *
 * A chunk's copy-in, transform and copy-out stay in program order on ONE in-order queue;
 * a stream opened with Q > 1 queues sends consecutive chunks to different queues, so they
 * overlap: chunk n+1's H2D copy runs under chunk n's kernel and chunk n-1's D2H copy (the
 * sensor buffers of reference src/signal_source.c:29-35 keep arriving while earlier ones are
 * still being transformed and returned).  No event chains between queues: the HIP calls per
 * chunk stay at four (copy, launch, copy, record).  Results are delivered in push order
 * whatever order the queues finish in (the worker walks the ring).
 *
 * How many queues (measured with rows copied back by the copy engine, one MI355X, 128-frame
 * chunks, profiles/r03_multi_stream_experiments.txt; the kernel now stores rows into the pinned
 * slot itself, see rtlws_stream_open_q):
 *   one sensor on the device      1 queue 2.56e6 spectra/s, 4 queues 4.81e6 (+88 %)
 *   four / eight sensors          1 queue EACH 6.9e6 / 5.9e6; 4 each 5.5e6 / 3.8e6 (32 HIP
 *                                 streams oversubscribe the hardware queues); a shared pool of
 *                                 4 or 8 queues 4.3-5.3e6; copy-in / transform / copy-out
 *                                 queues chained by events (8 HIP calls per chunk) 3.7e6 / 2.3e6
 * so the count is the caller's: rtlws_stream_open() takes RTLWS_STREAM_QUEUES (default 1),
 * rtlws_stream_open_q() an argument -- the configs[4] driver gives a device's only sensor
 * four queues and sensors that share a device one each. */
#define _GNU_SOURCE
#include "rtlws_stream.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "topology.h"

enum { SLOT_FREE = 0, SLOT_IN_FLIGHT = 1 };

struct slot {
    void* h_in;      /* pinned */
    void* h_out;     /* pinned */
    void* d_in;      /* device staging of the input (NULL when the kernel reads the pinned slot) */
    void* d_out;     /* device staging of the rows (NULL when the kernel stores into the pinned slot) */
    void* q;         /* this slot's in-order queue (NULL: the engine's own) */
    void* done;      /* recorded after the D2H copy; the worker sleeps on it */
    int state;
    long first_frame;
    double t_push_ms;
};

struct rtlws_stream {
    rtlws_engine* eng;
    int zero_copy_out;               /* the kernel stores its rows straight into the pinned host slot */
    int zero_copy_in;                /* experiment: the kernel reads the pinned host slot itself */
    int nq;                          /* dedicated queues of this stream (0: the engine's own) */
    void* q[8];
    rtlws_spectra_desc desc;
    long frames_per_chunk, rows_per_chunk;
    size_t in_bytes, out_bytes;
    int nslots;
    struct slot* slots;
    int head, tail;                  /* next slot to fill / next to complete */
    long next_frame;
    rtlws_stream_callback cb;
    void* user;
    pthread_t worker;
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_free;
    int closing;
    int worker_ready;                /* the worker thread has made its first HIP call */
    rtlws_stream_stats st;
    double lat_sum;
    rtlws_topo_info topo;            /* the device's NUMA node and its CPUs */
    int cpus_pinned;                 /* CPUs the worker thread (and the slots' first touch) is pinned to */
};

int rtlws_stream_device_for(int stream_index, int device_count)
{
    if (stream_index < 0 || device_count < 1) return -1;
    return stream_index % device_count;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static size_t sample_bytes(const rtlws_spectra_desc* d)
{
    const size_t r = d->cic_r > 1 ? (size_t)d->cic_r : 1u;
    switch (d->input) {
    case RTLWS_IN_CS32: return 8;
    case RTLWS_IN_RF32: return 4;
    default: return 2 * r;
    }
}

/* bytes of one output value: payload bytes, f64 rows (RTLWS_FLAG_F64 without
 * RTLWS_FLAG_ROWS_F32), f32 rows otherwise */
static size_t out_elem_bytes(const rtlws_spectra_desc* d)
{
    if (d->output == RTLWS_OUT_PAYLOAD_U8) return 1;
    return ((d->flags & RTLWS_FLAG_F64) && !(d->flags & RTLWS_FLAG_ROWS_F32)) ? 8 : 4;
}

/* copy in, transform, copy out, mark: in program order on this slot's queue */
static int enqueue_chunk(rtlws_stream* s, struct slot* sl)
{
    const void* in = s->zero_copy_in ? sl->h_in : sl->d_in;
    void* out = s->zero_copy_out ? sl->h_out : sl->d_out;
    return (!s->zero_copy_in && rtlws_copy_h2d(s->eng, sl->d_in, sl->h_in, s->in_bytes, sl->q)) ||
           ((s->desc.flags & RTLWS_FLAG_F64)      /* the reference's arithmetic (src/spectrum.c:54-60,21,28) */
                ? rtlws_spectra_batch_f64(s->eng, &s->desc, in, s->frames_per_chunk, out, sl->q)
                : rtlws_spectra_batch(s->eng, &s->desc, in, s->frames_per_chunk, out, sl->q)) ||
           (!s->zero_copy_out && rtlws_copy_d2h(s->eng, sl->h_out, sl->d_out, s->out_bytes, sl->q)) ||
           rtlws_event_record(sl->done, s->eng, sl->q);
}

static void* worker_main(void* arg)
{
    rtlws_stream* s = (rtlws_stream*)arg;
    /* A thread's first HIP call sets up its runtime state (milliseconds): make it now, on slot 0's
     * already completed warm-start event, not on the first real chunk's. */
    (void)rtlws_event_sync(s->slots[0].done);
    pthread_mutex_lock(&s->mu);
    s->worker_ready = 1;
    pthread_cond_broadcast(&s->cv_free);
    pthread_mutex_unlock(&s->mu);
    for (;;) {
        struct slot* sl;
        double lat;
        int failed;
        pthread_mutex_lock(&s->mu);
        while (!s->closing && s->slots[s->tail].state != SLOT_IN_FLIGHT)
            pthread_cond_wait(&s->cv_work, &s->mu);
        if (s->slots[s->tail].state != SLOT_IN_FLIGHT) {      /* closing and drained */
            pthread_mutex_unlock(&s->mu);
            return NULL;
        }
        sl = &s->slots[s->tail];
        pthread_mutex_unlock(&s->mu);

        /* a chunk whose device work failed has no valid rows: it is counted, never
         * delivered (h_out holds stale or partial data) */
        failed = rtlws_event_sync(sl->done) != 0;
        if (failed) fprintf(stderr, "rtlws_stream: device failure: %s\n", rtlws_last_error());
        lat = now_ms() - sl->t_push_ms;
        if (s->cb && !failed) s->cb(sl->h_out, s->rows_per_chunk, sl->first_frame, lat, s->user);

        pthread_mutex_lock(&s->mu);
        sl->state = SLOT_FREE;
        s->tail = (s->tail + 1) % s->nslots;
        s->st.chunks_done++;                      /* retired: delivered or failed */
        if (failed) {
            s->st.chunks_failed++;
        } else {
            s->st.frames_done += s->frames_per_chunk;
            s->lat_sum += lat;
            if (lat > s->st.latency_ms_max) s->st.latency_ms_max = lat;
        }
        pthread_cond_broadcast(&s->cv_free);
        pthread_mutex_unlock(&s->mu);
    }
}

rtlws_stream* rtlws_stream_open(int device, const rtlws_spectra_desc* desc, long frames_per_chunk,
                                int ring_slots, rtlws_stream_callback cb, void* user)
{
    const char* q = getenv("RTLWS_STREAM_QUEUES");
    int queues = q ? atoi(q) : 1;
    if (queues < 1) queues = 1;
    if (queues > ring_slots) queues = ring_slots;
    if (queues > 8) queues = 8;
    return rtlws_stream_open_q(device, desc, frames_per_chunk, ring_slots, queues, cb, user);
}

static rtlws_stream* open_pinned(int device, const rtlws_spectra_desc* desc, long frames_per_chunk,
                                 int ring_slots, int queues, rtlws_stream_callback cb, void* user,
                                 const rtlws_topo_info* topo, int cpus_pinned);

rtlws_stream* rtlws_stream_open_q(int device, const rtlws_spectra_desc* desc, long frames_per_chunk,
                                  int ring_slots, int queues, rtlws_stream_callback cb, void* user)
{
    rtlws_stream* s;
    rtlws_topo_info topo;
    cpu_set_t saved;
    int have_saved = 0, pinned;
    if (!desc || rtlws_spectra_kernel_kind(desc) == 0 || frames_per_chunk <= 0 ||
        frames_per_chunk % desc->k_avg || ring_slots < 2 || queues < 1 || queues > 8 || queues > ring_slots)
        return NULL;
    /* everything below -- the pinned slots' allocation and first touch, the warm start, the worker thread's
     * creation (a new thread inherits its creator's mask) -- happens next to the device; the caller gets its
     * own mask back on every way out */
    rtlws_topo_describe(device, NULL, NULL, &topo);
    pinned = rtlws_topo_pin_save(&topo, &saved, &have_saved);
    s = open_pinned(device, desc, frames_per_chunk, ring_slots, queues, cb, user, &topo, pinned > 0 ? pinned : 0);
    rtlws_topo_restore(&saved, have_saved);
    return s;
}

static rtlws_stream* open_pinned(int device, const rtlws_spectra_desc* desc, long frames_per_chunk,
                                 int ring_slots, int queues, rtlws_stream_callback cb, void* user,
                                 const rtlws_topo_info* topo, int cpus_pinned)
{
    rtlws_stream* s;
    int i;
    s = (rtlws_stream*)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->topo = *topo;
    s->cpus_pinned = cpus_pinned;
    s->eng = rtlws_engine_create(device);
    if (!s->eng) { free(s); return NULL; }
    s->desc = *desc;
    s->frames_per_chunk = frames_per_chunk;
    s->rows_per_chunk = frames_per_chunk / desc->k_avg;
    s->in_bytes = (size_t)frames_per_chunk * (size_t)desc->n_fft * sample_bytes(desc);
    s->out_bytes = (size_t)s->rows_per_chunk * (size_t)desc->n_fft * out_elem_bytes(desc);
    s->nslots = ring_slots;
    s->cb = cb;
    s->user = user;
    s->slots = (struct slot*)calloc((size_t)ring_slots, sizeof(struct slot));
    if (!s->slots) {
        rtlws_engine_destroy(s->eng);
        free(s);
        return NULL;
    }
    pthread_mutex_init(&s->mu, NULL);
    pthread_cond_init(&s->cv_work, NULL);
    pthread_cond_init(&s->cv_free, NULL);
    {
        /* Rows written by the kernel itself into pinned (device-mapped) host memory -- no D2H copy,
         * one HIP call fewer per chunk, and the copy engine out of the way of the H2D copies:
         * one sensor 4.8e6 -> 6.9e6 spectra/s, eight on one device 5.8e6 -> 9.1e6, byte-payload rows
         * 1.85e7 -> 2.2e7 (profiles/r03_multi_stream.jsonl).  RTLWS_STREAM_ZEROCOPY_OUT=0 stages the
         * rows in device memory and copies them (A/B).  Reading the INPUT from the pinned slot the
         * same way (RTLWS_STREAM_ZEROCOPY_IN=1) helps one sensor (+22 %) and costs 27 % where the
         * H2D link is the limit (the copy engine moves 45 GB/s, the kernel's 2-byte loads 32): off. */
        const char* z = getenv("RTLWS_STREAM_ZEROCOPY_OUT");
        s->zero_copy_out = !(z && z[0] == '0');
        z = getenv("RTLWS_STREAM_ZEROCOPY_IN");
        s->zero_copy_in = (z && z[0] == '1');
    }
    if (queues > 1) {                 /* 1: the engine's own queue, as before */
        for (i = 0; i < queues; i++) {
            s->q[i] = rtlws_queue_create(s->eng);
            if (!s->q[i]) { rtlws_stream_close(s); return NULL; }
            s->nq = i + 1;
        }
    }
    {
        /* the worker sleeps on the event, it does not spin (RTLWS_STREAM_SPIN=1: it spins; A/B) */
        const char* sp = getenv("RTLWS_STREAM_SPIN");
        const int spin = sp && sp[0] == '1';
        for (i = 0; i < ring_slots; i++) {
            struct slot* sl = &s->slots[i];
            sl->q = s->nq ? s->q[i % s->nq] : NULL;
            sl->h_in = rtlws_pinned_alloc(s->in_bytes);
            sl->h_out = rtlws_pinned_alloc(s->out_bytes);
            /* device staging only for the side that is not zero-copy */
            if (!s->zero_copy_in) sl->d_in = rtlws_dev_alloc(s->eng, s->in_bytes);
            if (!s->zero_copy_out) sl->d_out = rtlws_dev_alloc(s->eng, s->out_bytes);
            sl->done = spin ? rtlws_event_create() : rtlws_event_create_blocking();
            if (!sl->h_in || !sl->h_out || (!s->zero_copy_in && !sl->d_in) || (!s->zero_copy_out && !sl->d_out) ||
                !sl->done) {
                rtlws_stream_close(s);
                return NULL;
            }
        }
    }
    /* Warm start: one chunk of mid-scale samples through every slot's own chain, now.  The first
     * launch of a process pays the code-object load and the twiddle-table upload, the first touch
     * of a pinned slot its mapping -- 19-25 ms on the first buffers of three of eight sensors in
     * round 3 (profiles/r03_multi_stream.jsonl, latency_ms_max), against 0.05-0.5 ms afterwards.
     * Paid here, before the sensor starts, a latency budget never sees it. */
    for (i = 0; i < ring_slots; i++) {
        struct slot* sl = &s->slots[i];
        memset(sl->h_in, desc->input == RTLWS_IN_CU8 ? 128 : 0, s->in_bytes);
        if (enqueue_chunk(s, sl) || rtlws_event_sync(sl->done)) {
            rtlws_stream_close(s);
            return NULL;
        }
    }
    if (pthread_create(&s->worker, NULL, worker_main, s) != 0) {
        rtlws_stream_close(s);
        return NULL;
    }
    pthread_mutex_lock(&s->mu);
    while (!s->worker_ready) pthread_cond_wait(&s->cv_free, &s->mu);
    pthread_mutex_unlock(&s->mu);
    return s;
}

int rtlws_stream_push(rtlws_stream* s, const void* iq_host, int block)
{
    struct slot* sl;
    int rc = 0;
    if (!s || !iq_host) return -1;
    pthread_mutex_lock(&s->mu);
    while (s->slots[s->head].state != SLOT_FREE) {
        if (!block) {
            s->st.chunks_dropped++;
            s->next_frame += s->frames_per_chunk;      /* the dropped frames keep their numbers */
            pthread_mutex_unlock(&s->mu);
            return 1;
        }
        pthread_cond_wait(&s->cv_free, &s->mu);
    }
    sl = &s->slots[s->head];
    sl->first_frame = s->next_frame;
    sl->t_push_ms = now_ms();
    memcpy(sl->h_in, iq_host, s->in_bytes);
    if (enqueue_chunk(s, sl)) {
        /* part of the chain may already be queued against this slot's buffers, and
         * the slot stays FREE: drain the queue so that the next push cannot
         * overwrite h_in / d_in under a copy or a kernel still in flight */
        rtlws_stream_sync(s->eng, sl->q);
        s->st.chunks_failed++;
        s->next_frame += s->frames_per_chunk;          /* the lost frames keep their numbers */
        rc = -3;
    } else {
        sl->state = SLOT_IN_FLIGHT;
        s->head = (s->head + 1) % s->nslots;
        s->next_frame += s->frames_per_chunk;
        s->st.chunks_pushed++;
        pthread_cond_signal(&s->cv_work);
    }
    pthread_mutex_unlock(&s->mu);
    return rc;
}

int rtlws_stream_flush(rtlws_stream* s)
{
    if (!s) return -1;
    pthread_mutex_lock(&s->mu);
    while (s->st.chunks_done < s->st.chunks_pushed) pthread_cond_wait(&s->cv_free, &s->mu);
    pthread_mutex_unlock(&s->mu);
    return 0;
}

void rtlws_stream_get_stats(rtlws_stream* s, rtlws_stream_stats* out)
{
    pthread_mutex_lock(&s->mu);
    *out = s->st;
    {
        const long delivered = s->st.frames_done / s->frames_per_chunk;
        out->latency_ms_avg = delivered ? s->lat_sum / (double)delivered : 0.0;
    }
    pthread_mutex_unlock(&s->mu);
}

int rtlws_stream_topology(const rtlws_stream* s, rtlws_topo_info* out, int* cpus_pinned)
{
    if (!s) return -1;
    if (out) *out = s->topo;
    if (cpus_pinned) *cpus_pinned = s->cpus_pinned;
    return 0;
}

void rtlws_stream_close(rtlws_stream* s)
{
    int i;
    if (!s) return;
    if (s->worker) {
        rtlws_stream_flush(s);
        pthread_mutex_lock(&s->mu);
        s->closing = 1;
        pthread_cond_broadcast(&s->cv_work);
        pthread_mutex_unlock(&s->mu);
        pthread_join(s->worker, NULL);
    }
    for (i = 0; i < s->nslots; i++) {
        struct slot* sl = &s->slots[i];
        rtlws_pinned_free(sl->h_in);
        rtlws_pinned_free(sl->h_out);
        if (s->eng) {
            rtlws_dev_free(s->eng, sl->d_in);
            rtlws_dev_free(s->eng, sl->d_out);
        }
        rtlws_event_destroy(sl->done);
    }
    free(s->slots);
    for (i = 0; i < s->nq; i++) rtlws_queue_destroy(s->eng, s->q[i]);
    rtlws_engine_destroy(s->eng);
    pthread_mutex_destroy(&s->mu);
    pthread_cond_destroy(&s->cv_work);
    pthread_cond_destroy(&s->cv_free);
    free(s);
}
