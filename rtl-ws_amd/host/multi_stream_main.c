/* multi_stream_main.c -- BASELINE.json configs[4] driver: S independent IQ
 * streams, stream i on device (i % ndev), one producer thread each, no
 * collective.  Paced mode feeds every stream 131 072-sample buffers at a given
 * sample rate (2.4 MS/s -> 2343.75 spectra/s/stream) and reports drops and
 * latency; unpaced mode pushes as fast as the host link allows.
 *
 *   rtlws_multi_stream [--streams S] [--seconds T] [--rate HZ | --unpaced]
 *                      [--nfft N] [--k K] [--output f32|db|payload]
 *                      [--chunk-buffers M] [--queues Q] [--plan-only --devices D]
 *                      [--precision f32|f64|f64c_f32o]
 * Prints one JSON line: totals, and per stream its device, rate, drops and latency.
 *   --output payload   u8 rows (the 1024 bytes src/main.c:82 sends): 4x fewer D2H bytes
 *   --chunk-buffers M  M sensor buffers per chunk (unpaced throughput runs)
 *   --precision        arithmetic of the streams: the f32 fused kernel (default), the reference's f64
 *                      (rows of doubles), or f64 arithmetic with f32 rows (rtlws_stream.h, desc.flags)
 *   --plan-only        print the stream -> device plan for D devices and exit: device, PCI bus id, NUMA node and
 *                      cpuset per stream.  --bus-ids a,b,... names the devices' bus ids (no GPU is asked then),
 *                      --sysfs-root DIR reads the NUMA information under DIR instead of /sys
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rtlws_stream.h"

#define BUF_SAMPLES 131072

struct producer {
    int id, device;
    double rate_hz, seconds;
    int unpaced, chunk_buffers, queues;
    rtlws_spectra_desc desc;
    rtlws_stream* st;
    rtlws_stream_stats stats;
    double elapsed_s;
    volatile double checksum;
    long rows_seen;
    rtlws_topo_info topo;
    int cpus_pinned;
};

static void on_rows(const void* rows, long nrows, long first_frame, double latency_ms, void* user)
{
    struct producer* p = (struct producer*)user;
    (void)first_frame; (void)latency_ms;
    /* touch the data like a consumer would */
    p->checksum += (p->desc.output == RTLWS_OUT_PAYLOAD_U8) ? (double)((const unsigned char*)rows)[0]
                   : ((p->desc.flags & RTLWS_FLAG_F64) && !(p->desc.flags & RTLWS_FLAG_ROWS_F32))
                       ? ((const double*)rows)[0] : (double)((const float*)rows)[0];
    p->rows_seen += nrows;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* producer_main(void* arg)
{
    struct producer* p = (struct producer*)arg;
    const long frames = (long)p->chunk_buffers * (BUF_SAMPLES / p->desc.n_fft);
    unsigned char* buf = (unsigned char*)malloc((size_t)2 * BUF_SAMPLES * (size_t)p->chunk_buffers);
    unsigned x = 2463534242u + 977u * (unsigned)p->id;
    double t0, next;
    long i;
    /* this thread feeds device p->device for the rest of its life: stay on that device's NUMA node, so the
     * buffer below is first touched there and every push's memcpy into a ring slot is node-local */
    {
        rtlws_topo_info t;
        rtlws_topo_describe(p->device, NULL, NULL, &t);
        (void)rtlws_topo_pin_thread(&t);
    }
    for (i = 0; i < (long)BUF_SAMPLES * p->chunk_buffers; i++) {      /* tone + noise, different per stream */
        double ph = 2.0 * 3.14159265358979 * (0.05 + 0.1 * p->id) * (double)i;
        double re, im;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        re = 0.6 * cos(ph) + ((double)(x & 0xffff) / 65536.0 - 0.5) * 0.2;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        im = 0.6 * sin(ph) + ((double)(x & 0xffff) / 65536.0 - 0.5) * 0.2;
        buf[2 * i] = (unsigned char)fmin(255.0, fmax(0.0, floor(re * 128.0 + 128.5)));
        buf[2 * i + 1] = (unsigned char)fmin(255.0, fmax(0.0, floor(im * 128.0 + 128.5)));
    }
    p->st = rtlws_stream_open_q(p->device, &p->desc, frames, 4, p->queues, on_rows, p);
    if (!p->st) { fprintf(stderr, "stream %d: open failed: %s\n", p->id, rtlws_last_error()); free(buf); return NULL; }
    rtlws_stream_topology(p->st, &p->topo, &p->cpus_pinned);
    t0 = now_s();
    next = t0;
    while (now_s() - t0 < p->seconds) {
        if (!p->unpaced) {
            struct timespec ts;
            next += (double)BUF_SAMPLES * p->chunk_buffers / p->rate_hz;
            ts.tv_sec = (time_t)next;
            ts.tv_nsec = (long)((next - floor(next)) * 1e9);
            clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &ts, NULL);
            rtlws_stream_push(p->st, buf, 0);     /* a real sensor cannot wait: full ring = drop */
        } else {
            rtlws_stream_push(p->st, buf, 1);
        }
    }
    rtlws_stream_flush(p->st);
    p->elapsed_s = now_s() - t0;
    rtlws_stream_get_stats(p->st, &p->stats);
    rtlws_stream_close(p->st);
    free(buf);
    return NULL;
}

int main(int argc, char** argv)
{
    int streams = 8, nfft = 1024, k = 1, unpaced = 0, i, ndev, output = RTLWS_OUT_POWER_SUM;
    int chunk_buffers = 1, plan_only = 0, devices_override = 0, queues_override = 0;
    const char* output_name = "f32";
    const char* precision = "f32";
    const char* bus_ids = NULL;
    const char* sysfs_root = NULL;
    int flags = 0;
    double seconds = 3.0, rate = 2400000.0;
    struct producer* ps;
    pthread_t* th;
    double total_rate = 0.0, lat_max = 0.0, lat_avg = 0.0;
    long drops = 0, failed = 0, frames = 0;
    for (i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--streams") && i + 1 < argc) streams = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--seconds") && i + 1 < argc) seconds = atof(argv[++i]);
        else if (!strcmp(argv[i], "--rate") && i + 1 < argc) rate = atof(argv[++i]);
        else if (!strcmp(argv[i], "--nfft") && i + 1 < argc) nfft = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--k") && i + 1 < argc) k = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--unpaced")) unpaced = 1;
        else if (!strcmp(argv[i], "--chunk-buffers") && i + 1 < argc) chunk_buffers = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--plan-only")) plan_only = 1;
        else if (!strcmp(argv[i], "--bus-ids") && i + 1 < argc) bus_ids = argv[++i];
        else if (!strcmp(argv[i], "--sysfs-root") && i + 1 < argc) sysfs_root = argv[++i];
        else if (!strcmp(argv[i], "--queues") && i + 1 < argc) queues_override = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) devices_override = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--precision") && i + 1 < argc) {
            precision = argv[++i];
            if (!strcmp(precision, "f32")) flags = 0;
            else if (!strcmp(precision, "f64")) flags = RTLWS_FLAG_F64;
            else if (!strcmp(precision, "f64c_f32o")) flags = RTLWS_FLAG_F64 | RTLWS_FLAG_ROWS_F32;
            else { fprintf(stderr, "--precision f32|f64|f64c_f32o\n"); return 2; }
        }
        else if (!strcmp(argv[i], "--output") && i + 1 < argc) {
            output_name = argv[++i];
            if (!strcmp(output_name, "f32")) output = RTLWS_OUT_POWER_SUM;
            else if (!strcmp(output_name, "db")) output = RTLWS_OUT_MEAN_DB;
            else if (!strcmp(output_name, "payload")) output = RTLWS_OUT_PAYLOAD_U8;
            else { fprintf(stderr, "--output f32|db|payload\n"); return 2; }
        }
    }
    if (streams < 1 || chunk_buffers < 1) { fprintf(stderr, "bad --streams / --chunk-buffers\n"); return 2; }
    if (plan_only) {        /* the placement rule alone: no device is touched */
        if (devices_override < 1) { fprintf(stderr, "--plan-only needs --devices D\n"); return 2; }
        printf("{\"streams\": %d, \"devices\": %d, \"plan_only\": true, \"stream_devices\": [", streams, devices_override);
        for (i = 0; i < streams; i++) printf("%s%d", i ? ", " : "", rtlws_stream_device_for(i, devices_override));
        printf("], \"device_topology\": [");
        for (i = 0; i < devices_override; i++) {
            rtlws_topo_info t;
            char one[32];
            const char* bus = NULL;
            if (bus_ids) {                 /* the i-th comma-separated entry */
                const char* p = bus_ids;
                int skip = i;
                size_t n;
                while (skip > 0 && (p = strchr(p, ',')) != NULL) { ++p; --skip; }
                n = p ? strcspn(p, ",") : 0;
                if (p && n > 0 && n < sizeof one) { memcpy(one, p, n); one[n] = 0; bus = one; }
                else bus = "";
            }
            if (rtlws_topo_describe(bus ? -1 : i, bus, sysfs_root, &t) != 0) { memset(&t, 0, sizeof t); t.numa_node = -1; }
            printf("%s{\"device\": %d, \"bus_id\": \"%s\", \"numa_node\": %d, \"cpus\": %d, \"cpulist\": \"%s\"}",
                   i ? ", " : "", i, t.bus_id, t.numa_node, t.ncpus, t.cpulist);
        }
        printf("]}\n");
        return 0;
    }
    ndev = rtlws_device_count();
    if (ndev < 1) { fprintf(stderr, "no HIP device (there is no CPU path)\n"); return 2; }
    if (devices_override > 0 && devices_override < ndev) ndev = devices_override;    /* use fewer than there are */
    ps = (struct producer*)calloc((size_t)streams, sizeof(*ps));
    th = (pthread_t*)calloc((size_t)streams, sizeof(*th));
    for (i = 0; i < streams; i++) {
        ps[i].id = i;
        ps[i].device = rtlws_stream_device_for(i, ndev);
        ps[i].chunk_buffers = chunk_buffers;
        /* a device's only sensor gets four queues (consecutive chunks overlap), two sensors two
         * each, more than that one each (stream_gpu.c); --queues overrides */
        {
            const int on_dev = (streams - ps[i].device + ndev - 1) / ndev;     /* sensors on this device */
            ps[i].queues = queues_override > 0 ? queues_override : (on_dev >= 4 ? 1 : 4 / on_dev);
        }
        ps[i].rate_hz = rate;
        ps[i].seconds = seconds;
        ps[i].unpaced = unpaced;
        ps[i].desc.n_fft = nfft;
        ps[i].desc.k_avg = k;
        ps[i].desc.input = RTLWS_IN_CU8;
        ps[i].desc.window = RTLWS_WIN_RECT;
        ps[i].desc.output = output;
        ps[i].desc.flags = flags;
        pthread_create(&th[i], NULL, producer_main, &ps[i]);
    }
    for (i = 0; i < streams; i++) pthread_join(th[i], NULL);
    for (i = 0; i < streams; i++) {
        if (ps[i].elapsed_s > 0) total_rate += (double)ps[i].stats.frames_done / ps[i].elapsed_s;
        drops += ps[i].stats.chunks_dropped;
        failed += ps[i].stats.chunks_failed;
        frames += ps[i].stats.frames_done;
        lat_avg += ps[i].stats.latency_ms_avg / streams;
        if (ps[i].stats.latency_ms_max > lat_max) lat_max = ps[i].stats.latency_ms_max;
    }
    printf("{\"streams\": %d, \"devices\": %d, \"paced\": %s, \"rate_hz\": %.0f, \"n_fft\": %d, \"k_avg\": %d, "
           "\"output\": \"%s\", \"precision\": \"%s\", \"chunk_buffers\": %d, "
           "\"seconds\": %.2f, \"spectra_per_s_total\": %.1f, \"spectra_per_s_per_stream\": %.1f, "
           "\"frames_done\": %ld, \"chunks_dropped\": %ld, \"chunks_failed\": %ld, \"latency_ms_avg\": %.3f, \"latency_ms_max\": %.3f, "
           "\"per_stream\": [",
           streams, ndev, unpaced ? "false" : "true", rate, nfft, k, output_name, precision, chunk_buffers, seconds, total_rate,
           total_rate / streams, frames, drops, failed, lat_avg, lat_max);
    for (i = 0; i < streams; i++)
        printf("%s{\"stream\": %d, \"device\": %d, \"queues\": %d, \"spectra_per_s\": %.1f, \"chunks_dropped\": %ld, \"chunks_failed\": %ld, "
               "\"latency_ms_avg\": %.3f, \"latency_ms_max\": %.3f, \"bus_id\": \"%s\", \"numa_node\": %d, \"cpus_pinned\": %d}",
               i ? ", " : "", i, ps[i].device, ps[i].queues,
               ps[i].elapsed_s > 0 ? (double)ps[i].stats.frames_done / ps[i].elapsed_s : 0.0,
               ps[i].stats.chunks_dropped, ps[i].stats.chunks_failed, ps[i].stats.latency_ms_avg,
               ps[i].stats.latency_ms_max, ps[i].topo.bus_id, ps[i].topo.numa_node, ps[i].cpus_pinned);
    printf("]}\n");
    free(ps);
    free(th);
    return 0;
}
