/*
 * rtlws_stream.h -- host-fed streaming front end of the spectrum engine.
 *
 * What the live server needs once it stops throwing away 122 of every 128
 * frames (reference src/cbb_main.c:46-49 transforms at most 6 frames per
 * 250 ms): sensor buffers arrive on the host (reference
 * src/signal_source.c:29-35: 131 072 samples per callback with librtlsdr's
 * defaults), are staged in a pinned ring, copied to the device
 * asynchronously, transformed by the fused kernel and the results copied
 * back -- each chunk on the stream's own HIP stream, several chunks in
 * flight, completion reported in order on a worker thread.
 *
 * One stream per sensor; streams on different devices (or on the same one) are
 * independent: BASELINE.json configs[4] is eight of these, one per GPU, no
 * collective.  The throughput of this path is PCIe-bound (2 bytes in and
 * 4/K bytes out per sample); it is never what bench.py reports as `value`.
 */
#ifndef RTLWS_STREAM_H
#define RTLWS_STREAM_H

#include "rtlws_hip.h"
#include "rtlws_topo.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtlws_stream rtlws_stream;

/* Called on the stream's worker thread, in push order, once a chunk's results
 * are in host memory (a chunk whose device work failed is counted in
 * chunks_failed and NOT delivered).  `rows` points at rows_in_chunk * n_fft outputs (f32; bytes
 * for RTLWS_OUT_PAYLOAD_U8; f64 for a stream opened with desc->flags & RTLWS_FLAG_F64 and without
 * RTLWS_FLAG_ROWS_F32) valid only during the call. */
typedef void (*rtlws_stream_callback)(const void* rows, long nrows, long first_frame,
                                      double latency_ms, void* user);

typedef struct rtlws_stream_stats {
    long chunks_pushed;
    long chunks_done;          /* retired by the worker: delivered + failed in flight */
    long chunks_dropped;       /* rtlws_stream_push(..., block = 0) found the ring full */
    long frames_done;          /* frames whose rows reached the callback */
    double latency_ms_avg;     /* push -> results on the host */
    double latency_ms_max;
    long chunks_failed;        /* device failure at enqueue (push returned -3) or in flight;
                                  such a chunk is never handed to the callback */
} rtlws_stream_stats;

/* frames_per_chunk must be a multiple of desc->k_avg; ring_slots >= 2 chunks may
 * be in flight.  desc->flags selects the arithmetic: 0 = the f32 fused kernel (rtlws_spectra_batch),
 * RTLWS_FLAG_F64 = the reference's f64 (rtlws_spectra_batch_f64; + RTLWS_FLAG_ROWS_F32: f32 rows).
 * Opening a stream also warms it: tables built, code object loaded, one chunk of mid-scale samples
 * through every ring slot, so the first real chunk has the latency of any other.
 * NUMA: for the length of the call the CALLING thread is pinned to the CPUs of the device's NUMA node
 * (rtlws_topo.h) -- the pinned ring slots are allocated and first touched there, and the worker thread,
 * created meanwhile, inherits that mask and keeps it; the caller's own mask is put back before the call
 * returns.  A producer that wants its pushes (a memcpy into a ring slot each) local too pins itself the
 * same way (rtlws_topo_pin_thread; rtl-ws_amd/host/multi_stream_main.c does).
 * NULL on failure (rtlws_last_error). */
rtlws_stream* rtlws_stream_open(int device, const rtlws_spectra_desc* desc, long frames_per_chunk,
                                int ring_slots, rtlws_stream_callback cb, void* user);

/* The same with the number of in-order device queues stated: 1 <= queues <= min(ring_slots, 8).
 * With more than one, consecutive chunks go to different queues and their copy-in, transform
 * and copy-out overlap (a chunk's own three steps stay in order on its queue).  Worth +88 % for
 * a device's only sensor; several sensors sharing one device do best with one queue each
 * (rtl-ws_amd/host/stream_gpu.c has the measurements).  rtlws_stream_open() uses
 * RTLWS_STREAM_QUEUES, default 1. */
rtlws_stream* rtlws_stream_open_q(int device, const rtlws_spectra_desc* desc, long frames_per_chunk,
                                  int ring_slots, int queues, rtlws_stream_callback cb, void* user);

/* Hand over one chunk of frames_per_chunk frames of host IQ (copied before the
 * call returns).  block != 0: wait for a free ring slot; block == 0: count a
 * drop and return 1 when the ring is full.  0 on success, -1 bad argument,
 * -3 device failure. */
int rtlws_stream_push(rtlws_stream* s, const void* iq_host, int block);

/* Wait until every pushed chunk has been delivered. */
int rtlws_stream_flush(rtlws_stream* s);

void rtlws_stream_get_stats(rtlws_stream* s, rtlws_stream_stats* out);

/* Where the stream's worker thread and pinned slots live: the device's PCI bus id, NUMA node and cpuset
 * (unknown: "" / -1 / 0) and the number of CPUs the worker is pinned to (0: not pinned).  0 / -1. */
int rtlws_stream_topology(const rtlws_stream* s, rtlws_topo_info* out, int* cpus_pinned);

void rtlws_stream_close(rtlws_stream* s);

/* BASELINE.json configs[4]'s placement rule -- "GPU g <- stream g", independent streams,
 * no collective: the device of stream `stream_index` on a host with
 * `device_count` devices is stream_index mod device_count (8 streams on 8 devices: one
 * each; on fewer devices they share round-robin).  -1 when either argument is invalid.
 * Pure arithmetic: callable (and tested) without a GPU. */
int rtlws_stream_device_for(int stream_index, int device_count);

#ifdef __cplusplus
}
#endif
#endif /* RTLWS_STREAM_H */
