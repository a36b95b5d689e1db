/*
 * rtlws_multi.h -- one device-resident batch of frames sharded over the GPUs of a node.
 *
 * BASELINE.json configs[1] "at 1/2/4/8 GPUs" and SURVEY.md §8e: spectrum frames are
 * independent, so a batch of B frames splits into contiguous frame ranges
 * [g*B/G, (g+1)*B/G) -- aligned to K so that no K-frame average (src/cbb_main.c:50-59: the
 * accumulation and the DC-slot weights of src/spectrum.c:25-33 couple the K frames of one
 * output row, and nothing else) ever spans two devices -- one range per device, one host
 * pthread and one engine (own HIP stream, own tables) per device, NO collective and no
 * peer access: xGMI is not used.  The host side is plain C over rtlws_hip.h.
 *
 * Lifetime: open (one thread per shard is created and lives until close: it pins itself to the CPUs
 * of its device's NUMA node -- rtlws_topo.h -- and THEN creates its engine and allocates its device's
 * share of the input and of the rows and two pinned staging buffers) -> upload (each shard thread
 * copies its own range, through its pinned staging buffers) -> run any number of times (each shard
 * thread enqueues `launches` launches of its range on its engine's stream between two HIP events;
 * the threads are released together by one broadcast) -> download -> close.
 */
#ifndef RTLWS_MULTI_H
#define RTLWS_MULTI_H

#include "rtlws_hip.h"
#include "rtlws_topo.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The frame range of shard g of G: rows (K-groups) are dealt as evenly as whole rows allow,
 * in order -- shard g owns rows [g*R/G, (g+1)*R/G), R = nframes / k_avg (floor), i.e. frames
 * [k_avg * (g*R/G), k_avg * ((g+1)*R/G)).  Every frame of the R whole K-groups belongs to exactly
 * one shard, ranges are contiguous and ascending, sizes differ by at most one row, and a shard
 * may be empty (more devices than rows).  The nframes % k_avg frames after the last whole
 * K-group belong to no shard -- the rule of src/cbb_main.c:49 (blocks = len / 1024, the rest of
 * the buffer is not transformed).  Pure arithmetic: callable (and tested) without a GPU.
 * 0, or -1 for nframes < 0, k_avg < 1, G < 1 or g outside [0, G). */
int rtlws_multi_partition(long nframes, int k_avg, int shards, int g, long* first_frame, long* frame_count);

typedef struct rtlws_multi rtlws_multi;

typedef struct rtlws_multi_shard_stats {
    int device;               /* HIP device of this shard */
    long first_frame, frames; /* its range of the batch */
    int launches;             /* launches timed by the last rtlws_multi_run */
    double event_ms;          /* HIP-event time around those launches on the shard's stream */
    double wall_ms;           /* host clock of the shard's thread: barrier -> its stream drained */
    int rc;                   /* 0, or the first failing call's code (-1 / -3) */
} rtlws_multi_shard_stats;

/* One shard per entry of device_ids (n_shards >= 1): shard g runs on HIP device device_ids[g].
 * device_ids == NULL: shards on devices 0 .. n_shards-1 (n_shards == 0: every device of the
 * host).  The same device may be named more than once (a rehearsal of the N-shard path on fewer
 * GPUs: the shards then share that device, each with its own engine and stream).
 * f64 != 0: rtlws_spectra_batch_f64 (rows of doubles, or floats with RTLWS_FLAG_ROWS_F32).
 * nframes need not be a multiple of desc->k_avg: the whole K-groups are used (rtlws_multi_frames).
 * NULL on failure (rtlws_last_error; there is no CPU path). */
rtlws_multi* rtlws_multi_open(int n_shards, const int* device_ids, const rtlws_spectra_desc* desc,
                              long nframes, int f64);
int rtlws_multi_shards(const rtlws_multi* m);
/* Why the last upload / run / download of `m` returned non-zero: "shard g (device d): <call>: <the HIP
 * shim's text>" of its first failing shard ("" after a success).  The failing call ran on a shard's own
 * thread, so the CALLER's rtlws_last_error() does not hold it.  m == NULL: the same for the last
 * rtlws_multi_open of the CALLING THREAD that returned NULL (per thread, like rtlws_last_error()).
 * A handle is single-caller: upload / run / download / error / close of one rtlws_multi must come from one thread
 * at a time (its shards share one command mailbox); different handles are independent. */
const char* rtlws_multi_error(const rtlws_multi* m);
/* Where shard g runs: its device's PCI bus id, NUMA node and cpuset as the shard thread found them
 * (rtlws_topo.h; unknown fields "" / -1 / 0) and the number of CPUs the thread pinned itself to
 * (0: not pinned -- no NUMA information, or the node's CPUs lie outside the job's mask).  0 / -1. */
int rtlws_multi_shard_topology(const rtlws_multi* m, int g, rtlws_topo_info* out, int* cpus_pinned);
/* frames the shards cover: k_avg * (nframes / k_avg) */
long rtlws_multi_frames(const rtlws_multi* m);
/* bytes of one input frame / one output row, as the batch API lays them out */
size_t rtlws_multi_frame_bytes(const rtlws_multi* m);
size_t rtlws_multi_row_bytes(const rtlws_multi* m);

/* host_frames: the batch's frames, contiguous; every shard copies its own range.  0 / -3. */
int rtlws_multi_upload(rtlws_multi* m, const void* host_frames);
/* `launches` launches per shard, all shards concurrently; stats: n_shards entries (may be NULL).
 * *wall_ms_max (may be NULL) = the longest shard's wall clock: the job's time.  0 / -1 / -3. */
int rtlws_multi_run(rtlws_multi* m, int launches, rtlws_multi_shard_stats* stats, double* wall_ms_max);
/* host_rows: nframes / k_avg (floor) rows, contiguous, in frame order.  0 / -3. */
int rtlws_multi_download(rtlws_multi* m, void* host_rows);
void rtlws_multi_close(rtlws_multi* m);

#ifdef __cplusplus
}
#endif
#endif /* RTLWS_MULTI_H */
