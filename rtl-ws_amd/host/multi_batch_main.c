/* multi_batch_main.c -- BASELINE.json configs[1] "at 1/2/4/8 GPUs", driven from C: one
 * device-resident batch of B frames sharded over the devices of the node (rtlws_multi.h: shard g
 * = rows [g*R/G, (g+1)*R/G), one host pthread + one engine per device, no collective), W untimed
 * then L timed launches per device, per-device HIP-event times, one JSON line.
 *
 *   rtlws_multi_batch [--frames B] [--nfft N] [--k K] [--launches L] [--warmup W] [--devices D]
 *                     [--precision f32|f64|f64c_f32o] [--window hann] [--output f32|db|payload]
 *                     [--cic R] [--shards-per-device Q] [--shards-on-device0 S] [--plan-only]
 *   --devices D            use the first D devices (default: all)
 *   --shards-per-device Q  cut every device's share into Q shards with their own engine, queue and host
 *                          thread (device order 0,0,..,1,1,..): consecutive launches of different queues
 *                          overlap each other's fill and drain phases -- one MI355X: 0.66 of the HBM
 *                          roofline with Q = 1, 0.71-0.72 with Q = 2 or 3 (profiles/r04_shards_one_device.txt)
 *   --shards-on-device0 S  rehearsal: S shards, all on device 0 (the S-shard code path on one GPU;
 *                          the line says so and is not a scaling measurement)
 *   --plan-only            print the plan for --devices D and exit: frame range, PCI bus id, NUMA node and cpuset
 *                          of every device.  No GPU is touched when --bus-ids names the devices' bus ids (planning
 *                          for another host; tests); without it the HIP runtime is asked, and a host without a
 *                          device gets the ranges alone
 *   --bus-ids a,b,...      (with --plan-only) bus id of device 0, 1, ...: "0000:05:00.0,0000:15:00.0"
 *   --sysfs-root DIR       (with --plan-only) read NUMA nodes and cpusets under DIR instead of /sys
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtlws_multi.h"

static void synth_iq(unsigned char* buf, long samples)
{
    unsigned x = 2463534242u;
    long i;
    for (i = 0; i < samples; i++) {      /* tone + noise, quantised like an RTL2832U sample */
        const double ph = 2.0 * 3.14159265358979 * 0.1373 * (double)i;
        double re, im;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        re = 0.6 * cos(ph) + ((double)(x & 0xffff) / 65536.0 - 0.5) * 0.2;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        im = 0.6 * sin(ph) + ((double)(x & 0xffff) / 65536.0 - 0.5) * 0.2;
        buf[2 * i] = (unsigned char)fmin(255.0, fmax(0.0, floor(re * 128.0 + 128.5)));
        buf[2 * i + 1] = (unsigned char)fmin(255.0, fmax(0.0, floor(im * 128.0 + 128.5)));
    }
}

int main(int argc, char** argv)
{
    long frames = 65536;
    int nfft = 1024, k = 1, launches = 200, warmup = 500, devices = 0, plan_only = 0, rehearsal = 0, cic = 0, per_dev = 1, i, g, n;
    int output = RTLWS_OUT_POWER_SUM, window = RTLWS_WIN_RECT, flags = 0, f64 = 0, rc;
    const char* precision = "f32";
    const char* output_name = "f32";
    const char* bus_ids = NULL;
    const char* sysfs_root = NULL;
    rtlws_spectra_desc d;
    rtlws_multi* m;
    rtlws_multi_shard_stats* st;
    int* ids = NULL;
    unsigned char* host;
    double wall_ms = 0.0, total = 0.0, bytes_per_frame;
    for (i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atol(argv[++i]);
        else if (!strcmp(argv[i], "--nfft") && i + 1 < argc) nfft = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--k") && i + 1 < argc) k = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--launches") && i + 1 < argc) launches = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) devices = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cic") && i + 1 < argc) cic = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--shards-on-device0") && i + 1 < argc) rehearsal = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--shards-per-device") && i + 1 < argc) per_dev = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--plan-only")) plan_only = 1;
        else if (!strcmp(argv[i], "--bus-ids") && i + 1 < argc) bus_ids = argv[++i];
        else if (!strcmp(argv[i], "--sysfs-root") && i + 1 < argc) sysfs_root = argv[++i];
        else if (!strcmp(argv[i], "--window") && i + 1 < argc) window = !strcmp(argv[++i], "hann") ? RTLWS_WIN_HANN : RTLWS_WIN_RECT;
        else if (!strcmp(argv[i], "--precision") && i + 1 < argc) {
            precision = argv[++i];
            if (!strcmp(precision, "f32")) { f64 = 0; flags = 0; }
            else if (!strcmp(precision, "f64")) { f64 = 1; flags = 0; }
            else if (!strcmp(precision, "f64c_f32o")) { f64 = 1; flags = RTLWS_FLAG_ROWS_F32; }
            else { fprintf(stderr, "--precision f32|f64|f64c_f32o\n"); return 2; }
        } else if (!strcmp(argv[i], "--output") && i + 1 < argc) {
            output_name = argv[++i];
            if (!strcmp(output_name, "f32")) output = RTLWS_OUT_POWER_SUM;
            else if (!strcmp(output_name, "db")) output = RTLWS_OUT_MEAN_DB;
            else if (!strcmp(output_name, "payload")) output = RTLWS_OUT_PAYLOAD_U8;
            else { fprintf(stderr, "--output f32|db|payload\n"); return 2; }
        } else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (per_dev < 1 || per_dev > 4) { fprintf(stderr, "--shards-per-device 1..4\n"); return 2; }
    if (frames < 0 || k < 1 || launches < 1 || warmup < 0) { fprintf(stderr, "bad --frames / --k / --launches\n"); return 2; }
    if (plan_only) {                       /* the partition alone: no device is touched */
        if (devices < 1) { fprintf(stderr, "--plan-only needs --devices D\n"); return 2; }
        printf("{\"frames\": %ld, \"frames_used\": %ld, \"k_avg\": %d, \"devices\": %d, \"plan_only\": true, \"shards\": [",
               frames, frames - frames % k, k, devices);
        for (g = 0; g < devices; g++) {
            long first, count;
            rtlws_topo_info t;
            char one[32];
            const char* bus = NULL;
            rtlws_multi_partition(frames, k, devices, g, &first, &count);
            if (bus_ids) {                 /* the g-th comma-separated entry */
                const char* p = bus_ids;
                int skip = g;
                size_t n;
                while (skip > 0 && (p = strchr(p, ',')) != NULL) { ++p; --skip; }
                n = p ? strcspn(p, ",") : 0;
                if (p && n > 0 && n < sizeof one) { memcpy(one, p, n); one[n] = 0; bus = one; }
                else bus = "";
            }
            if (rtlws_topo_describe(bus ? -1 : g, bus, sysfs_root, &t) != 0) memset(&t, 0, sizeof t), t.numa_node = -1;
            printf("%s{\"device\": %d, \"first_frame\": %ld, \"frames\": %ld, \"bus_id\": \"%s\", \"numa_node\": %d, "
                   "\"cpus\": %d, \"cpulist\": \"%s\"}", g ? ", " : "", g, first, count, t.bus_id, t.numa_node, t.ncpus, t.cpulist);
        }
        printf("]}\n");
        return 0;
    }
    memset(&d, 0, sizeof d);
    d.n_fft = nfft; d.k_avg = k; d.input = RTLWS_IN_CU8; d.window = window; d.output = output; d.cic_r = cic; d.flags = flags;
    n = rtlws_device_count();
    if (n < 1) { fprintf(stderr, "no HIP device (there is no CPU path)\n"); return 2; }
    if (devices > n) { fprintf(stderr, "--devices %d but this host has %d HIP device(s)\n", devices, n); return 2; }
    if (devices > 0) n = devices;
    if (rehearsal > 0) {
        n = rehearsal;
        ids = (int*)calloc((size_t)n, sizeof(int));       /* all zero: device 0 */
    } else if (per_dev > 1) {
        const int ndev = n;
        n = ndev * per_dev;
        ids = (int*)calloc((size_t)n, sizeof(int));
        for (g = 0; g < n; g++) ids[g] = g / per_dev;     /* contiguous frame ranges stay on one device */
    }
    m = rtlws_multi_open(n, ids, &d, frames, f64);
    if (!m) { fprintf(stderr, "%s\n", rtlws_multi_error(NULL)); return 3; }
    {
        const size_t fb = rtlws_multi_frame_bytes(m);
        host = (unsigned char*)malloc((size_t)frames * fb + 1);
        if (!host) return 3;
        synth_iq(host, (long)((size_t)frames * fb / 2));
    }
    st = (rtlws_multi_shard_stats*)calloc((size_t)n, sizeof(*st));
    rc = rtlws_multi_upload(m, host);
    if (!rc && warmup) rc = rtlws_multi_run(m, warmup, NULL, NULL);     /* clock governor + code objects */
    if (!rc) rc = rtlws_multi_run(m, launches, st, &wall_ms);
    if (rc) { fprintf(stderr, "run failed (%d): %s\n", rc, rtlws_multi_error(m)); return 3; }
    frames = rtlws_multi_frames(m);                 /* whole K-groups */
    total = (double)frames * launches / (wall_ms * 1e-3);
    bytes_per_frame = (double)rtlws_multi_frame_bytes(m) + (double)rtlws_multi_row_bytes(m) / k;
    printf("{\"metric\": \"spectra/s (%d-pt IQ frames)\", \"workload\": \"multi_batch\", \"frames_used\": %ld, \"n_fft\": %d, \"k_avg\": %d, "
           "\"cic_r\": %d, \"precision\": \"%s\", \"output\": \"%s\", \"shards\": %d, \"shards_per_device\": %d, \"devices_of_host\": %d, "
           "\"rehearsal_all_on_device0\": %s, \"launches\": %d, \"warmup\": %d, \"wall_ms\": %.4f, "
           "\"spectra_per_s_total\": %.1f, \"algorithmic_bytes_per_frame\": %.0f, \"per_shard\": [",
           nfft, frames, nfft, k, cic, precision, output_name, n, rehearsal > 0 ? 0 : per_dev, rtlws_device_count(), rehearsal > 0 ? "true" : "false",
           launches, warmup, wall_ms, total, bytes_per_frame);
    for (g = 0; g < n; g++) {
        const double ev_s = st[g].event_ms * 1e-3 / launches;
        rtlws_topo_info t;
        int pinned = 0;
        rtlws_multi_shard_topology(m, g, &t, &pinned);
        printf("%s{\"shard\": %d, \"device\": %d, \"first_frame\": %ld, \"frames\": %ld, \"event_ms_per_launch\": %.5f, "
               "\"wall_ms\": %.4f, \"spectra_per_s\": %.1f, \"hbm_gbs\": %.1f, \"bus_id\": \"%s\", \"numa_node\": %d, "
               "\"cpus_pinned\": %d}", g ? ", " : "", g, st[g].device,
               st[g].first_frame, st[g].frames, st[g].event_ms / launches, st[g].wall_ms,
               ev_s > 0 ? (double)st[g].frames / ev_s : 0.0, ev_s > 0 ? bytes_per_frame * (double)st[g].frames / ev_s / 1e9 : 0.0,
               t.bus_id, t.numa_node, pinned);
    }
    printf("]}\n");
    rtlws_multi_close(m);
    free(st);
    free(host);
    free(ids);
    return 0;
}
