/* host_stress.c -- TEST-ONLY driver of the threaded host layer over the fake shim (fake_rtlws_hip.cpp), built
 * and run by tests/test_host_sanitizers_cpu.py under -fsanitize=thread and -fsanitize=address,undefined.
 * What runs here is the PRODUCT's host code (rtl-ws_amd/host/ *.c, unmodified): the ring walk, condition
 * variables and in-order delivery of stream_gpu.c; the shard threads and command mailbox of multi_batch.c;
 * the two-slot hand-off of cbb_gpu.c with rf_decimator_set_parameters arriving from a second thread as in
 * reference src/main.c:154; the drop-in spectrum.h path from two threads.  Rows are checked against direct
 * oracle calls, so a race that corrupts data fails even where the sanitizer sees nothing. */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "cbb_main.h"
#include "rf_decimator.h"
#include "rtlws_multi.h"
#include "rtlws_oracle.h"
#include "rtlws_stream.h"
#include "signal_source.h"
#include "spectrum.h"

long fake_hip_live_objects(void);
void fake_hip_fail_after(int n);

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); __atomic_add_fetch(&g_fail, 1, __ATOMIC_RELAXED); } } while (0)

static void fill_iq(unsigned char* buf, size_t bytes, unsigned seed)
{
    unsigned x = 2463534242u + seed * 7919u;
    size_t i;
    for (i = 0; i < bytes; i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; buf[i] = (unsigned char)(x >> 11); }
}

static void nap_us(long us)
{
    struct timespec ts = {us / 1000000, (us % 1000000) * 1000};
    nanosleep(&ts, NULL);
}

/* ---- streams ------------------------------------------------------------------------------- */

#define N 1024
#define FPC 4                      /* frames per chunk */
struct sink {
    const unsigned char* src;      /* what was pushed (every chunk the same bytes), or NULL: no value check */
    long chunks, last_first, order_errors, value_errors;
    int f64;
};

static void on_rows(const void* rows, long nrows, long first_frame, double latency_ms, void* user)
{
    struct sink* s = (struct sink*)user;
    (void)latency_ms;
    if (first_frame % FPC != 0 || first_frame <= s->last_first || nrows != FPC) s->order_errors++;
    s->last_first = first_frame;
    if (s->src) {
        static __thread double ref[FPC * N];
        long i;
        orc_batch_spectra_u8(N, 1, FPC, s->src, NULL, ref, 1);
        for (i = 0; i < FPC * N; i++) {
            const double got = s->f64 ? ((const double*)rows)[i] : (double)((const float*)rows)[i];
            const double tol = s->f64 ? 1e-12 : 1e-6;
            if (fabs(got - ref[i]) > tol * (fabs(ref[i]) + 1.0)) { s->value_errors++; break; }
        }
    }
    s->chunks++;
}

static rtlws_spectra_desc desc_of(int f64)
{
    rtlws_spectra_desc d;
    memset(&d, 0, sizeof d);
    d.n_fft = N; d.k_avg = 1; d.input = RTLWS_IN_CU8; d.window = RTLWS_WIN_RECT; d.output = RTLWS_OUT_POWER_SUM;
    d.flags = f64 ? RTLWS_FLAG_F64 : 0;
    return d;
}

static void stream_sections(void)
{
    unsigned char* iq = (unsigned char*)malloc(2 * N * FPC);
    rtlws_spectra_desc d = desc_of(0), d64 = desc_of(1);
    rtlws_stream_stats st;
    struct sink k;
    rtlws_stream* s;
    int i, rc;
    fill_iq(iq, 2 * N * FPC, 1);

    /* A: in-order delivery, two queues, f32 and f64 */
    for (i = 0; i < 2; i++) {
        memset(&k, 0, sizeof k); k.last_first = -1; k.src = iq; k.f64 = i;
        s = rtlws_stream_open_q(i, i ? &d64 : &d, FPC, 4, 2, on_rows, &k);
        CHECK(s != NULL, "stream open: %s", rtlws_last_error());
        if (!s) continue;
        { int j; for (j = 0; j < 40; j++) CHECK(rtlws_stream_push(s, iq, 1) == 0, "blocking push"); }
        rtlws_stream_flush(s);
        rtlws_stream_get_stats(s, &st);
        CHECK(st.chunks_pushed == 40 && st.chunks_done == 40 && st.frames_done == 40 * FPC && st.chunks_dropped == 0,
              "A: stats %ld %ld %ld %ld", st.chunks_pushed, st.chunks_done, st.frames_done, st.chunks_dropped);
        CHECK(k.chunks == 40 && k.order_errors == 0 && k.value_errors == 0, "A: sink %ld %ld %ld", k.chunks, k.order_errors, k.value_errors);
        { rtlws_topo_info t; int pinned = -1; CHECK(rtlws_stream_topology(s, &t, &pinned) == 0 && t.device == i && pinned >= 0, "A: topology"); }
        rtlws_stream_close(s);
    }

    /* B: a full ring drops (non-blocking), the dropped frames keep their numbers */
    memset(&k, 0, sizeof k); k.last_first = -1; k.src = iq;
    s = rtlws_stream_open_q(0, &d, FPC, 2, 1, on_rows, &k);
    CHECK(s != NULL, "B: open");
    if (s) {
        long ok = 0, dropped = 0;
        for (i = 0; i < 60; i++) { rc = rtlws_stream_push(s, iq, 0); if (rc == 0) ok++; else if (rc == 1) dropped++; else CHECK(0, "B: push rc %d", rc); }
        rtlws_stream_flush(s);
        rtlws_stream_get_stats(s, &st);
        CHECK(ok + dropped == 60 && st.chunks_dropped == dropped && st.chunks_done == ok && k.chunks == ok, "B: %ld ok %ld dropped, stats %ld %ld, sink %ld", ok, dropped, st.chunks_done, st.chunks_dropped, k.chunks);
        CHECK(dropped > 0, "B: nothing was dropped (ring of 2, 60 back-to-back pushes)");
        CHECK(k.order_errors == 0 && k.value_errors == 0 && k.last_first <= 59 * FPC, "B: order %ld values %ld last %ld", k.order_errors, k.value_errors, k.last_first);
        rtlws_stream_close(s);
    }

    /* C: close with chunks in flight delivers them all */
    memset(&k, 0, sizeof k); k.last_first = -1; k.src = iq;
    s = rtlws_stream_open_q(1, &d, FPC, 8, 4, on_rows, &k);
    CHECK(s != NULL, "C: open");
    if (s) {
        for (i = 0; i < 7; i++) CHECK(rtlws_stream_push(s, iq, 1) == 0, "C: push");
        rtlws_stream_close(s);
        CHECK(k.chunks == 7 && k.order_errors == 0 && k.value_errors == 0, "C: %ld delivered", k.chunks);
    }

    /* E: a failing launch is counted, not delivered, and the stream goes on */
    memset(&k, 0, sizeof k); k.last_first = -1; k.src = iq;
    s = rtlws_stream_open_q(0, &d, FPC, 4, 1, on_rows, &k);
    CHECK(s != NULL, "E: open");
    if (s) {
        fake_hip_fail_after(3);
        for (i = 0, rc = 0; i < 10; i++) { const int r = rtlws_stream_push(s, iq, 1); if (r == -3) rc++; else CHECK(r == 0, "E: push rc %d", r); }
        fake_hip_fail_after(-1);
        rtlws_stream_flush(s);
        rtlws_stream_get_stats(s, &st);
        CHECK(rc == 1 && st.chunks_failed == 1 && st.chunks_done == 9 && k.chunks == 9 && k.value_errors == 0, "E: rc %d failed %ld done %ld sink %ld", rc, st.chunks_failed, st.chunks_done, k.chunks);
        rtlws_stream_close(s);
    }
    free(iq);
}

/* D: eight producers on eight streams (two fake devices), and four producers sharing one stream */
struct prod { int id; rtlws_stream* shared; long pushed, dropped; struct sink k; };

static void* producer(void* arg)
{
    struct prod* p = (struct prod*)arg;
    unsigned char* iq = (unsigned char*)malloc(2 * N * FPC);
    rtlws_spectra_desc d = desc_of(p->id & 1);
    rtlws_stream* s = p->shared;
    int i;
    fill_iq(iq, 2 * N * FPC, 100 + (unsigned)p->id);
    if (!s) {
        memset(&p->k, 0, sizeof p->k); p->k.last_first = -1; p->k.src = iq; p->k.f64 = p->id & 1;
        s = rtlws_stream_open_q(rtlws_stream_device_for(p->id, rtlws_device_count()), &d, FPC, 3, 1 + (p->id % 3), on_rows, &p->k);
        CHECK(s != NULL, "D: open %d", p->id);
        if (!s) { free(iq); return NULL; }
    }
    for (i = 0; i < 30; i++) {
        const int rc = rtlws_stream_push(s, iq, (i + p->id) % 3 != 0);
        if (rc == 0) p->pushed++; else if (rc == 1) p->dropped++; else CHECK(0, "D: push rc %d", rc);
        if (i % 7 == 0) nap_us(200);
    }
    if (!p->shared) {
        rtlws_stream_stats st;
        rtlws_stream_flush(s);
        rtlws_stream_get_stats(s, &st);
        CHECK(st.chunks_done == p->pushed && st.chunks_dropped == p->dropped && p->k.chunks == p->pushed && p->k.order_errors == 0 && p->k.value_errors == 0,
              "D: stream %d: pushed %ld done %ld sink %ld order %ld values %ld", p->id, p->pushed, st.chunks_done, p->k.chunks, p->k.order_errors, p->k.value_errors);
        rtlws_stream_close(s);
    }
    free(iq);
    return NULL;
}

static void many_producers(void)
{
    pthread_t th[12];
    struct prod ps[12];
    struct sink shared_sink;
    rtlws_spectra_desc d = desc_of(0);
    rtlws_stream* shared;
    rtlws_stream_stats st;
    long pushed = 0, dropped = 0;
    int i;
    memset(ps, 0, sizeof ps);
    memset(&shared_sink, 0, sizeof shared_sink); shared_sink.last_first = -1;      /* (four different inputs: order only) */
    shared = rtlws_stream_open_q(0, &d, FPC, 4, 2, on_rows, &shared_sink);
    CHECK(shared != NULL, "D: shared open");
    for (i = 0; i < 12; i++) { ps[i].id = i; ps[i].shared = (i >= 8) ? shared : NULL; pthread_create(&th[i], NULL, producer, &ps[i]); }
    for (i = 0; i < 12; i++) pthread_join(th[i], NULL);
    for (i = 8; i < 12; i++) { pushed += ps[i].pushed; dropped += ps[i].dropped; }
    if (shared) {
        rtlws_stream_flush(shared);
        rtlws_stream_get_stats(shared, &st);
        CHECK(pushed + dropped == 120 && st.chunks_done == pushed && st.chunks_dropped == dropped && shared_sink.chunks == pushed && shared_sink.order_errors == 0,
              "D: shared stream: pushed %ld dropped %ld done %ld sink %ld order %ld", pushed, dropped, st.chunks_done, shared_sink.chunks, shared_sink.order_errors);
        rtlws_stream_close(shared);
    }
}

/* ---- rtlws_multi ------------------------------------------------------------------------------- */

static void multi_section(void)
{
    const int ids[3] = {0, 1, 0};
    const int K = 2;
    const long nframes = 2 * 37 + 1;                  /* ragged: the last frame belongs to no K-group */
    unsigned char* host = (unsigned char*)malloc((size_t)nframes * 2 * N);
    int f64;
    fill_iq(host, (size_t)nframes * 2 * N, 77);
    for (f64 = 0; f64 < 2; f64++) {
        rtlws_spectra_desc d = desc_of(0);
        rtlws_multi* m;
        rtlws_multi_shard_stats st[3];
        double wall = 0.0;
        const long rows = nframes / K;
        double* ref = (double*)calloc((size_t)rows * N, sizeof(double));
        void* got = malloc((size_t)rows * N * (f64 ? 8 : 4));
        long i, bad = 0;
        int g;
        d.k_avg = K;
        m = rtlws_multi_open(3, ids, &d, nframes, f64);
        CHECK(m != NULL, "multi open: %s", rtlws_multi_error(NULL));
        if (!m) { free(ref); free(got); continue; }
        CHECK(rtlws_multi_shards(m) == 3 && rtlws_multi_frames(m) == 74, "multi: shape");
        CHECK(rtlws_multi_upload(m, host) == 0, "multi upload: %s", rtlws_multi_error(m));
        CHECK(rtlws_multi_run(m, 3, st, &wall) == 0 && wall >= 0.0, "multi run: %s", rtlws_multi_error(m));
        CHECK(rtlws_multi_run(m, 1, NULL, NULL) == 0, "multi run 2");
        CHECK(rtlws_multi_download(m, got) == 0, "multi download: %s", rtlws_multi_error(m));
        orc_batch_spectra_u8(N, K, rows * K, host, NULL, ref, 1);
        for (i = 0; i < rows * N; i++) {
            const double v = f64 ? ((double*)got)[i] : (double)((float*)got)[i];
            if (fabs(v - ref[i]) > (f64 ? 1e-12 : 1e-6) * (fabs(ref[i]) + 1.0)) bad++;
        }
        CHECK(bad == 0, "multi: %ld values differ from the oracle (f64=%d)", bad, f64);
        for (g = 0; g < 3; g++) {
            rtlws_topo_info t;
            int pinned = -1;
            CHECK(st[g].device == ids[g] && st[g].rc == 0 && st[g].launches == 3, "multi: stats of shard %d", g);
            CHECK(rtlws_multi_shard_topology(m, g, &t, &pinned) == 0 && t.device == ids[g] && t.bus_id[0] && pinned >= 0, "multi: topology of shard %d", g);
        }
        /* a failing launch on a shard thread: the code and the text reach the caller */
        fake_hip_fail_after(1);
        CHECK(rtlws_multi_run(m, 2, st, NULL) == -3, "multi: injected failure not reported");
        fake_hip_fail_after(-1);
        CHECK(strstr(rtlws_multi_error(m), "shard") && strstr(rtlws_multi_error(m), "injected"), "multi: error text '%s'", rtlws_multi_error(m));
        CHECK(rtlws_multi_run(m, 1, NULL, NULL) == 0 && rtlws_multi_error(m)[0] == 0, "multi: run after a failure");
        rtlws_multi_close(m);
        free(ref);
        free(got);
    }
    {   /* a device that does not exist: NULL, and the reason is readable */
        const int bad_ids[2] = {0, 9};
        rtlws_spectra_desc d = desc_of(0);
        CHECK(rtlws_multi_open(2, bad_ids, &d, 64, 0) == NULL, "multi: open on device 9 succeeded");
        CHECK(strstr(rtlws_multi_error(NULL), "shard 1") != NULL, "multi: open error text '%s'", rtlws_multi_error(NULL));
    }
    free(host);
}

/* ---- cbb_main.h with the decimator re-parameterised from a second thread (reference src/main.c:154) ---- */

static unsigned char g_first[6 * 1024 * 2];
static int g_have_first = 0;
static long g_decimated = 0;
static int g_stop = 0;

static void grab_first(const cmplx_u8* sig, int len)
{
    if (!__atomic_load_n(&g_have_first, __ATOMIC_ACQUIRE) && len >= 6 * 1024) {
        memcpy(g_first, sig, sizeof g_first);
        __atomic_store_n(&g_have_first, 1, __ATOMIC_RELEASE);
    }
}
static void on_decimated(const cmplx_s32* iq, int len) { (void)iq; __atomic_add_fetch(&g_decimated, len, __ATOMIC_RELAXED); }

static void* retune(void* arg)
{
    int i = 0;
    (void)arg;
    while (!__atomic_load_n(&g_stop, __ATOMIC_ACQUIRE)) {
        rf_decimator_set_parameters(cbb_rf_decimator(), 2048000.0, (i++ & 1) ? 8 : 10);
        nap_us(40000);          /* (a block is 100 ms of signal = 12.8 ms at this replay speed; every call resets the surplus) */
    }
    return NULL;
}

static void cbb_section(void)
{
    pthread_t th;
    char payload[2048];
    unsigned char want[1024];
    double ps[1024];
    int payloads = 0, i, equal = 0;
    const char* all = getenv("RTLWS_CBB_ALL_FRAMES");
    const int six_frames = !(all && atoi(all) > 0);       /* else: every frame of the buffer / of the interval */
    setenv("RTLWS_SYNTH_SPEEDUP", "8", 1);
    setenv("RTLWS_SYNTH_BUFLEN", "65536", 1);
    cbb_init(192000);
    rf_decimator_add_callback(cbb_rf_decimator(), on_decimated);
    signal_source_add_callback(grab_first);
    pthread_create(&th, NULL, retune, NULL);
    for (i = 0; i < 3000 && payloads < 4; i++) {        /* <= 3 s; four estimates take ~1.1 s unloaded */
        if (cbb_new_spectrum_available()) {
            const int n = cbb_get_spectrum_payload(payload, (int)sizeof payload, 15);
            CHECK(n == 1024, "cbb: payload of %d bytes", n);
            if (n == 1024 && __atomic_load_n(&g_have_first, __ATOMIC_ACQUIRE)) {
                const int blocks = orc_estimate_spectrum(g_first, 6 * 1024, ps);
                orc_spectrum_payload(1024, ps, blocks, 15, want);
                if (memcmp(payload, want, 1024) == 0) equal++;       /* every sensor buffer holds the same bytes */
            }
            payloads++;
        }
        nap_us(1000);
    }
    __atomic_store_n(&g_stop, 1, __ATOMIC_RELEASE);
    pthread_join(th, NULL);
    CHECK(payloads >= 3, "cbb: %d payloads in 3 s", payloads);
    if (six_frames) CHECK(equal == payloads, "cbb: %d of %d payloads equal the oracle's bytes", equal, payloads);
    CHECK(__atomic_load_n(&g_decimated, __ATOMIC_RELAXED) > 0, "cbb: the decimator callback never ran");
    cbb_close();
}

/* ---- spectrum.h from two threads, a handle each (the reference allows one thread per handle) ---- */

static void* dropin(void* arg)
{
    const unsigned seed = (unsigned)(size_t)arg;
    unsigned char* iq = (unsigned char*)malloc(6 * 2 * N);
    double ps[N], ref[N];
    struct spectrum* s = spectrum_alloc(N);
    int k, i, bad = 0;
    fill_iq(iq, 6 * 2 * N, seed);
    CHECK(s != NULL, "spectrum_alloc");
    if (s) {
        memset(ps, 0, sizeof ps);
        memset(ref, 0, sizeof ref);
        for (k = 0; k < 6; k++) {
            CHECK(spectrum_add_cmplx_u8(s, (const cmplx_u8*)(iq + (size_t)k * 2 * N), ps, N) == 0, "spectrum_add_cmplx_u8");
            orc_spectrum_add_cmplx_u8(N, iq + (size_t)k * 2 * N, NULL, ref, N);
        }
        CHECK(spectrum_add_cmplx_u8(s, (const cmplx_u8*)iq, ps, N - 1) == -1, "len != N must be -1 (src/spectrum.c:51-52)");
        for (i = 0; i < N; i++) if (fabs(ps[i] - ref[i]) > 1e-12 * (fabs(ref[i]) + 1.0)) bad++;
        CHECK(bad == 0, "drop-in: %d bins differ", bad);
        spectrum_free(s);
    }
    free(iq);
    return NULL;
}

int main(void)
{
    pthread_t a, b;
    stream_sections();
    many_producers();
    multi_section();
    CHECK(fake_hip_live_objects() == 0, "%ld engines / queues / events / buffers still alive after the stream and multi sections", fake_hip_live_objects());
    cbb_section();
    pthread_create(&a, NULL, dropin, (void*)(size_t)1);
    pthread_create(&b, NULL, dropin, (void*)(size_t)2);
    pthread_join(a, NULL);
    pthread_join(b, NULL);
    printf("host_stress: %d failure(s)\n", g_fail);
    return g_fail ? 1 : 0;
}
