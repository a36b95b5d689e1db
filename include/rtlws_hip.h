/*
 * rtlws_hip.h -- the thin extern-"C" HIP shim under the drop-in headers, and
 * the batch API that the roofline configurations use.
 *
 * Everything here is plain C: pointers, sizes, ints.  No HIP or torch types
 * appear in a signature; a stream is passed as `void*` (a hipStream_t; see
 * "Streams" below for what NULL means -- it is NOT HIP's default stream).
 *
 * Why a batch API exists at all: the reference interface
 * (src/spectrum.h:11-15) hands over one N-sample frame per call and wants
 * f64 back in host memory -- 2 KiB in, 8 KiB out per call.  That cannot get
 * near an HBM roofline, so the same arithmetic is also exposed over
 * device-resident batches: many frames in, one f32 (or dB, or payload-byte)
 * spectrum out per K consecutive frames.  spectrum_add_* (spectrum.h) is the
 * batch-of-one, K=1 case of rtlws_spectra_batch plus a host-side add.
 *
 * Reference semantics each entry point reproduces (paths under the
 * reference tree):
 *   rtlws_spectra_batch   src/spectrum.c:15-35,47-99 applied K times to a
 *                         zeroed row (the loop of src/cbb_main.c:50-59), with
 *                         optional src/cbb_main.c:121-130 dB/clamp epilogue
 *   rtlws_cic_block_sums  src/resample.c:21-40 (state handled by the caller)
 *   rtlws_halfband        src/resample.c:53-64
 *   rtlws_fm_demod        src/audio_main.c:110-131, src/common_sp.h:40-76
 */
#ifndef RTLWS_HIP_H
#define RTLWS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* librtlws_hip.so is built with -fvisibility=hidden and linked with a version script: these declarations, and
 * nothing else, are its dynamic symbols (tests/test_abi_cpu.py checks both directions) */
#pragma GCC visibility push(default)

typedef struct rtlws_engine rtlws_engine;

/* ---- Streams -------------------------------------------------------------
 *
 * Every asynchronous entry point takes a `void* stream`:
 *   a non-zero hipStream_t   the work is enqueued on exactly that stream;
 *   RTLWS_STREAM_ENGINE      (NULL) the ENGINE'S OWN stream.  It is created with
 *                            hipStreamNonBlocking, so it is NOT ordered against
 *                            HIP's default (null) stream: work the caller queued
 *                            on the default stream -- filling d_in, say -- may
 *                            still be running when the kernel starts, and memory
 *                            a stream-ordered allocator recycled on the default
 *                            stream may be written while its previous user is
 *                            still reading it.  Use rtlws_stream_sync(e, NULL) /
 *                            events to order against it;
 *   RTLWS_STREAM_DEFAULT     HIP's legacy default stream itself (hipStreamLegacy).
 *
 * The trap this spells out: torch.cuda.current_stream().cuda_stream is 0 for
 * torch's default stream, the same bits as NULL.  A torch caller on the default
 * stream passes RTLWS_STREAM_DEFAULT (rtlws.torch_stream_handle() does), or runs
 * producer and launches on one torch.cuda.Stream() (non-zero handle), as bench.py
 * does. */
#define RTLWS_STREAM_ENGINE  ((void*)0)
#define RTLWS_STREAM_DEFAULT ((void*)1)

/* ---- engine / device plumbing ---------------------------------------- */

/* Number of usable HIP devices (0 when none; never negative). */
int rtlws_device_count(void);

/* The device's PCI bus id, "0000:05:00.0" (domain:bus:device.function), NUL-terminated into buf[len],
 * len >= 16: what rtlws_topo.h maps to a NUMA node.  0 / -1 / -3. */
int rtlws_device_pci_bus_id(int device, char* buf, int len);

/* One engine per device per user: owns a stream and the twiddle/window
 * tables.  Returns NULL (and sets rtlws_last_error) if the device cannot be
 * used -- there is no CPU fallback behind this library. */
rtlws_engine* rtlws_engine_create(int device);
void rtlws_engine_destroy(rtlws_engine* e);
int rtlws_engine_device(const rtlws_engine* e);

/* Kernel-selection switches, for experiments and A/B tests.  Each is read from the environment
 * ONCE, when the engine is created (the variable in brackets), and can be changed afterwards only
 * here -- no launch path reads the environment:
 *   "v2"                [RTLWS_V2]                -1 default rule, 0 / 1 never / always the
 *                                                 two-virtual-threads-per-lane f32 kernel
 *   "blocks_per_cu"     [RTLWS_BLOCKS_PER_CU]     > 0: workgroups per CU of the f32 fused kernels
 *   "f64_fused"         [RTLWS_F64_FUSED]         0: f64 batches on the row-per-workgroup kernel
 *   "f64_blocks_per_cu" [RTLWS_F64_BLOCKS_PER_CU]
 *   "f64_x1024"         [RTLWS_F64_X1024]         0: rectangular 1024-point cmplx_u8 frames stay on the
 *                                                 two-transposition f64 kernel
 *   "f64_x_waves"       [RTLWS_F64_X_WAVES]       wavefronts per workgroup of the one-transposition f64 kernel:
 *                                                 0 by batch size (8 from 32 rows per CU on, where the device's
 *                                                 LDS limit per workgroup holds their 136 KiB), 1, 8
 *   "cic_direct"        [RTLWS_CIC_DIRECT]        1: per-lane loads for every CIC factor but 8
 *   "cic_round"         [RTLWS_CIC_ROUND]         1 | 2 | 4: LDS staging depth of the generic factors
 * (Removed in round 6: "split" = Q, a batch's rows as Q concurrent launches on engine-owned queues joined back
 * into the caller's stream.  Measured slower than one launch at every Q -- 0.42 against 0.46 of the HBM roofline
 * in f64 arithmetic, 0.54 against 0.65 in f32, profiles/r05_split_priority_queues.txt: the fork and the join are
 * cross-queue dependencies that cost more than the overlapped fill and drain phases win.  The overlap pays only
 * for INDEPENDENT batches on independent queues: rtlws_multi.h, shards per device.)
 * set: 0, -1 for an unknown name.  get: the value ("cu_count" is readable too), -2 if unknown.  An option must not be changed while another thread launches on the same engine
 * (the launch paths read the options without the engine's lock). */
int rtlws_engine_set_option(rtlws_engine* e, const char* name, int value);
int rtlws_engine_get_option(const rtlws_engine* e, const char* name);

/* Build the twiddle/window tables for an FFT size now (they are otherwise built
 * on the first rtlws_spectra_batch call, which allocates and copies and is
 * therefore not legal inside a hipGraph capture).  After this, batch launches
 * for that size only enqueue a kernel and may be captured.  0 / -1 / -3. */
int rtlws_engine_prepare(rtlws_engine* e, int n_fft);
/* The same for rtlws_spectra_batch_f64's tables (2 <= n_fft <= 8192); it also raises the dynamic-LDS limit of
 * every instantiation of that size that needs more than 64 KiB (hipFuncSetAttribute, once per instantiation
 * and device), which the first launch of such an instantiation would otherwise do. */
int rtlws_engine_prepare_f64(rtlws_engine* e, int n_fft);

/* Last error text of the calling thread ("" when none). */
const char* rtlws_last_error(void);

void* rtlws_dev_alloc(rtlws_engine* e, size_t bytes);
void rtlws_dev_free(rtlws_engine* e, void* dptr);
void* rtlws_pinned_alloc(size_t bytes);
void rtlws_pinned_free(void* hptr);
/* Asynchronous on `stream` (NULL: the engine's own non-blocking stream, see "Streams"). 0 on success. */
int rtlws_copy_h2d(rtlws_engine* e, void* dst_dev, const void* src_host, size_t bytes, void* stream);
int rtlws_copy_d2h(rtlws_engine* e, void* dst_host, const void* src_dev, size_t bytes, void* stream);
int rtlws_memset_dev(rtlws_engine* e, void* dst_dev, int value, size_t bytes, void* stream);
int rtlws_stream_sync(rtlws_engine* e, void* stream);

/* Additional in-order queues on the engine's device (hipStreamNonBlocking), for callers that
 * overlap copies with kernels: rtlws_stream.h runs copy-in, transform and copy-out of
 * consecutive chunks on three of them, ordered by events.  NULL on failure. */
void* rtlws_queue_create(rtlws_engine* e);
void rtlws_queue_destroy(rtlws_engine* e, void* queue);
/* Work enqueued on `stream` after this call starts only once `ev` (recorded earlier, on any
 * stream of the same device) has completed.  0 / -1 / -3. */
int rtlws_queue_wait_event(rtlws_engine* e, void* stream, void* ev);
/* An event whose rtlws_event_sync() sleeps instead of spinning (hipEventBlockingSync): for
 * worker threads that wait on many chunks per second and should leave their core to the
 * producers.  Not usable with rtlws_event_elapsed_ms (timing disabled). */
void* rtlws_event_create_blocking(void);

/* hipEvent timing on a stream, for bench.py's roofline leg.  A handle binds to
 * the device of the engine it is first recorded on (any thread, whatever its
 * current device is). */
void* rtlws_event_create(void);
void rtlws_event_destroy(void* ev);
int rtlws_event_record(void* ev, rtlws_engine* e, void* stream);
/* Block the calling thread until the event has completed. 0 / -3. */
int rtlws_event_sync(void* ev);
/* Synchronises on `stop`; returns milliseconds, or a negative value. */
float rtlws_event_elapsed_ms(void* start, void* stop);

/* ---- batched power spectra ------------------------------------------- */

enum rtlws_input {
    RTLWS_IN_CU8 = 0,       /* cmplx_u8, (x-128)/128      src/spectrum.c:54-58 */
    RTLWS_IN_CS32 = 1,      /* cmplx_s32, x/128           src/spectrum.c:72-76 */
    RTLWS_IN_RF32 = 2       /* real f32, (x, 0)           src/spectrum.c:90-94 */
};

enum rtlws_window {
    RTLWS_WIN_RECT = 0,     /* reference behaviour: no window */
    RTLWS_WIN_HANN = 1      /* periodic Hann; build extension */
};

enum rtlws_output {
    RTLWS_OUT_POWER_SUM = 0,   /* f32[N]: sum over the K frames, what K calls of
                                  spectrum_add_* leave in a zeroed buffer */
    RTLWS_OUT_MEAN_DB = 1,     /* f32[N]: 10*log10(sum / K) */
    RTLWS_OUT_PAYLOAD_U8 = 2   /* u8[N]: clamp((int)(10*log10(|g*sum/K|)), 0, 255),
                                  g = 10^(gain_db/10) with C integer division,
                                  src/cbb_main.c:112,125-128 */
};

typedef struct rtlws_spectra_desc {
    int n_fft;        /* 1024, 2048, 4096 (fused kernel) or any N >= 2 (direct DFT kernel) */
    int k_avg;        /* >= 1 consecutive frames accumulated per output spectrum */
    int input;        /* enum rtlws_input */
    int window;       /* enum rtlws_window */
    int output;       /* enum rtlws_output */
    int cic_r;        /* 0 or 1: none.  R > 1 (RTLWS_IN_CU8 only): every FFT input
                         sample is the CIC block sum of R consecutive cmplx_u8,
                         fed as spectrum_add_cmplx_s32 would (sum/128) */
    int gain_db;      /* RTLWS_OUT_PAYLOAD_U8 only */
    int flags;        /* RTLWS_FLAG_*; 0 = none (was `reserved`, always 0) */
} rtlws_spectra_desc;

/* rtlws_spectra_batch_f64 only: f64 ARITHMETIC, f32 ROWS.  Everything is computed in double
 * exactly as without the flag (conversion, DFT, |X|^2, K-frame sums, DC-slot rule, dB); each
 * output value is rounded to f32 once, on the store.  Rows are then n_fft floats -- the
 * 2N + 4N/K bytes per frame SURVEY.md §8d prices the contract with -- and the error against
 * the f64 reference is that one rounding (<= 6e-8 relative, any bin, any dynamic range),
 * where the f32 transform of rtlws_spectra_batch holds 1e-4 only within 50 dB of a row's
 * maximum.  No effect on RTLWS_OUT_PAYLOAD_U8 (bytes either way) or on rtlws_spectra_batch. */
#define RTLWS_FLAG_ROWS_F32 1
/* rtlws_stream.h only (a stream has ONE entry point for both arithmetics): transform this
 * stream's chunks with rtlws_spectra_batch_f64 -- the reference's own precision for a live
 * sensor (src/signal_source.c:29-35 -> src/cbb_main.c:52-59).  Rows are then doubles, or floats
 * with RTLWS_FLAG_ROWS_F32 as well.  Ignored by rtlws_spectra_batch / rtlws_spectra_batch_f64,
 * where the function called IS the choice. */
#define RTLWS_FLAG_F64 2

/* Precision: f32 arithmetic (inputs are exact in f32; twiddles are f64-computed and
 * rounded once).  Against an f64 evaluation: power within 1e-4 relative for every
 * bin within 50 dB of its row's maximum on a single frame, within 3e-5 under the
 * strict metric (floor 1e-9 of the row maximum) on K >= 6 averages of noise-like
 * input; mean dB within 2e-4 dB; payload bytes equal except +-1 where the f64 dB
 * value lies within 1e-3 of an integer.  A caller that needs the
 * reference's f64 results uses rtlws_spectra_batch_f64 below -- which is what
 * spectrum.h and cbb_main.h do.
 *
 * d_in : nframes * n_fft * max(cic_r,1) input samples, device memory.
 * d_out: (nframes / k_avg) rows of n_fft outputs, device memory.
 * nframes must be a multiple of k_avg; d_in and d_out 16-byte aligned
 * (any hipMalloc pointer, or one offset by whole frames / rows).
 * Asynchronous on `stream` (NULL = the engine's own stream, which is not ordered
 * against HIP's default stream: "Streams" above).
 * Returns 0; -1 bad descriptor/size; -3 HIP failure (see rtlws_last_error). */
int rtlws_spectra_batch(rtlws_engine* e, const rtlws_spectra_desc* desc, const void* d_in,
                        long nframes, void* d_out, void* stream);

/* The dB epilogue on its own (reference src/cbb_main.c:112,121-130) for sums
 * that already live on the device: out[i] = clamp((int)(10*log10(|g*sums[i]/count|)),
 * 0, 255), g = 10^(gain_db/10) with C integer division.  n values in, n bytes
 * out; count must be > 0.  0 / -1 / -3. */
int rtlws_payload_from_sums(rtlws_engine* e, const float* d_sums, int n, int count, int gain_db,
                            void* d_out_u8, void* stream);

/* ---- the same, in the reference's own precision (f64) -----------------------
 *
 * The reference converts to double, runs an f64 DFT and accumulates into the
 * caller's double buffer (src/spectrum.c:21-34,54-58); cbb_main.c does the
 * dB / truncate / clamp in double (src/cbb_main.c:112,121-130).  The
 * reference-API paths -- spectrum_add_* (spectrum.h) and cbb_main.h -- move
 * one to six frames per call, so they go through this entry point: same
 * descriptor, same semantics and frame layout as rtlws_spectra_batch, f64
 * arithmetic, any 2 <= n_fft <= 8192.  1024- / 2048- / 4096-point frames of any input
 * kind, and cmplx_u8 through the CIC-fused input stage for cic_r = 8, 10, 12, run the
 * fused throughput kernel (spectrum_f64_fused.hip; needs d_out -- and d_in when cic_r > 1
 * -- 16-byte aligned, else the general kernel is used); everything else one workgroup per
 * output row (radix-2 in LDS for powers of two, the direct sum otherwise).
 *   RTLWS_OUT_POWER_SUM / RTLWS_OUT_MEAN_DB : d_out rows of n_fft doubles
 *       (desc->flags & RTLWS_FLAG_ROWS_F32: rows of n_fft floats, see the flag)
 *   RTLWS_OUT_PAYLOAD_U8                    : d_out rows of n_fft bytes,
 *       clamp((int)(10*log10(fabs(g*sum/K))), 0, 255) evaluated in double
 * d_in 8-byte aligned; d_out 8-byte (4-byte for f32 rows and payload bytes).  0 / -1 / -3. */
int rtlws_spectra_batch_f64(rtlws_engine* e, const rtlws_spectra_desc* desc, const void* d_in,
                            long nframes, void* d_out, void* stream);

/* src/cbb_main.c:112,121-130 on f64 sums that live on the device, in double. */
int rtlws_payload_from_sums_f64(rtlws_engine* e, const double* d_sums, int n, int count, int gain_db,
                                void* d_out_u8, void* stream);

/* Averaging over SEVERAL launches with the reference's semantics (cbb_main.h's
 * RTLWS_CBB_ALL_FRAMES=2 mode: every sensor buffer of a 250 ms interval).  Each
 * launch j of rtlws_spectra_batch_f64 (RTLWS_OUT_POWER_SUM, k_avg = K_j, one row)
 * is folded into a running row with
 *   rtlws_welch_accumulate_f64(e, d_acc, d_row_j, n, frames_end_j, d_b)
 * where frames_end_j = K_0 + ... + K_j, and the interval is closed with
 *   rtlws_welch_finish_f64(e, d_acc, n, total_frames, d_b)
 * after which d_acc holds exactly what total_frames sequential spectrum_add_*
 * calls leave in a zeroed buffer -- including slot n/2, whose weights
 * (src/spectrum.c:25-33) depend on the position of a frame in the whole sequence --
 * and *d_b is zero again.  d_acc (n doubles) and d_b (one double) start zeroed.
 * 0 / -1 / -3. */
int rtlws_welch_accumulate_f64(rtlws_engine* e, double* d_acc, const double* d_part, int n,
                               long frames_end, double* d_b, void* stream);
int rtlws_welch_finish_f64(rtlws_engine* e, double* d_acc, int n, long total, double* d_b, void* stream);

/* Which kernel a descriptor selects: 1 fused, 2 direct DFT, 0 unsupported. */
int rtlws_spectra_kernel_kind(const rtlws_spectra_desc* desc);

/* ---- decimators ------------------------------------------------------ */

/* dst[m] = sum_{n<R} (src[m*R+n] - 128) per component, int32, for m < dst_len.
 * Stateless block sums; resample.h's cic_decimate adds the delay-line
 * bookkeeping on the host.  Device pointers.  0 / -1 / -3. */
int rtlws_cic_block_sums(rtlws_engine* e, int R, const void* d_src_cu8, long dst_len,
                         void* d_dst_cs32, void* stream);

/* y[n] = 0.5*x[2n-5] + sum_{k even} h[k]*x[2n-k] over a buffer that already
 * has the 10 history samples in front: d_x holds 10 + 2*out_len floats and
 * x[i] here means d_x[10 + i].  Device pointers.  0 / -1 / -3. */
int rtlws_halfband(rtlws_engine* e, const float* d_x, float* d_y, long out_len, void* stream);

/* FM-demodulator front end of the audio chain (reference src/audio_main.c:110-131
 * with atan2_approx of src/common_sp.h:40-76):
 *   phase_i = atan2_approx((float)im_i, (float)re_i)
 *   out_i   = clamp(phase_i - phase_{i-1}, -1, 1),   phase_{-1} = *d_prev_in
 * and *d_prev_out = phase_{len-1} (len = 0: *d_prev_out = *d_prev_in).  d_iq: len cmplx_s32; all pointers device;
 * d_prev_in and d_prev_out must differ.  Bit-identical to an IEEE evaluation
 * of the reference's expressions.  0 / -1 / -3. */
int rtlws_fm_demod(rtlws_engine* e, const void* d_iq_cs32, long len, const float* d_prev_in,
                   float* d_prev_out, float* d_out, void* stream);

/* ---- measurement: the shader clock OVER a series of launches ----------------------------
 * rtlws_clock_stamp enqueues `slots` one-wavefront workgroups on `stream`; workgroup i writes {s_memtime (shader
 * clocks), s_memrealtime (100 MHz), place, 0x5354414d50} to d_out[4 i .. 4 i + 3] -- device memory, 8-byte aligned --
 * and leaves; place = XCC_ID << 16 | HW_ID & 0xff30 (SE, SH, CU, SIMD).  s_memtime is a counter of the place it is read
 * at, so two such launches in one stream, before and after the launches being measured, are paired BY PLACE:
 * (memtime1 - memtime0) / (memrealtime1 - memrealtime0) x 100 MHz per place is the clock the package power governor
 * gave that interval (take the median; 2 048 slots cover the 1 024 SIMDs of an MI355X).  Nothing is resident beside
 * the launches in between: bench.py's roofline.sclk_ghz uses this since round 6.  0 / -1 / -3. */
int rtlws_clock_stamp(rtlws_engine* e, unsigned long long* d_out, int slots, void* stream);

/* The earlier instrument, kept for kernels with registers to spare and for the record of what it costs
 * (profiles/r06_clock_probe_perturbation.txt: +1 .. 7 % on the launches it sits beside, +38 % on kernels that fill a
 * SIMD's registers -- one of their workgroups then cannot be resident -- because it IS resident, on a hardware queue
 * of its own):
 * rtlws_clock_probe_start puts ONE wavefront on a queue of its own beside whatever the caller
 * enqueues next; it records s_memtime (shader clocks) and s_memrealtime (100 MHz) when it starts and
 * again when told to leave (rtlws_clock_probe_signal: returns at once; rtlws_clock_probe_stop:
 * signals if that has not been done, then waits for the wavefront) or after ~10 s by itself, sleeping
 * in between.  *sclk_ghz = d(memtime) / d(memrealtime) x 100 MHz: the clock the package power governor
 * actually gave the kernels that ran in that interval (bench.py's roofline.valu_issue_frac uses
 * it); *seconds = the interval.  start: NULL on failure; stop: 0 / -1 / -3 (the handle is consumed). */
void* rtlws_clock_probe_start(rtlws_engine* e);
void rtlws_clock_probe_signal(void* probe);
/* The same signal, given by the device: written when everything enqueued on `stream` so far has
 * completed (a stream write-value packet), so the host need not wait for the stream first.  0 / -1 / -3. */
int rtlws_clock_probe_signal_on_stream(void* probe, void* stream);
int rtlws_clock_probe_stop(void* probe, double* sclk_ghz, double* seconds);

/* Device-to-device copy on `stream` (delay-line upkeep of chained kernels). */
int rtlws_copy_d2d(rtlws_engine* e, void* dst_dev, const void* src_dev, size_t bytes, void* stream);

/* Bytes of LDS, VGPR count etc. are in DESIGN.md; this returns the grid the
 * fused kernel would launch for a descriptor and nframes (for tests). */
int rtlws_spectra_grid(rtlws_engine* e, const rtlws_spectra_desc* desc, long nframes,
                       int* blocks, int* threads, int* lds_bytes);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* RTLWS_HIP_H */
