/*
 * rtlws_oracle.h -- CPU oracle for the rtl-ws IQ -> power-spectrum hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into it.  The product path (rtl-ws_amd/) never links or loads it.
 *
 * It restates, in plain C with f64 arithmetic and its own FFT, what the
 * reference computes (citations are path:line under /root/reference):
 *
 *   spectrum stage   src/spectrum.c:15-35 (fft-shift, |X|^2 accumulate, DC-slot
 *                    rule), :47-63 (u8), :65-81 (s32), :83-99 (real f32)
 *   CIC              src/resample.c:6-45
 *   half-band        src/resample.c:4,47-67
 *   re-blocker       src/rf_decimator.c:53-119
 *   FM front end     src/common_sp.h:40-76, src/audio_main.c:74-145 (next row §8f.3)
 *   dB payload       src/cbb_main.c:106-135
 *   frame harness    src/cbb_main.c:40-70 (blocks = min(len/1024, 6))
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - CIC, half-band, rf_decimator, FM front end: pinned bit-exact against the
 *     reference's own object code (oracle/_ref, built from /root/reference/src/
 *     resample.c, rf_decimator.c, audio_main.c, list.c, rate_logger.c, common.c)
 *     and against tests/golden/ fixtures made from it.
 *   - spectrum stage and dB payload: PARITY UNPINNED by reference execution.
 *     src/spectrum.c needs FFTW3 (un-vendored, unpinned "-lfftw3",
 *     Makefile:21) which is absent from this image (no header, no library),
 *     and the reference ships no tests or golden vectors.  The FFT here is
 *     pinned by mathematics instead: a forward, unnormalised DFT with
 *     exp(-2*pi*i*n*k/N) (what fftw_plan_dft_1d(..., FFTW_FORWARD, ...)
 *     documents), checked in tests against a long-double O(N^2) DFT and
 *     against numpy's pocketfft.
 */
#ifndef RTLWS_ORACLE_H
#define RTLWS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- DFT ------------------------------------------------------------- */

/* Forward unnormalised DFT of N interleaved (re,im) f64 points; any N >= 1.
 * Power-of-two N: iterative radix-2; otherwise a long-double direct DFT. */
void orc_dft_forward(int N, const double* in, double* out);

/* Direct O(N^2) long-double DFT (used by tests to pin orc_dft_forward). */
void orc_dft_direct(int N, const double* in, double* out);

/* ---- spectrum.c semantics ------------------------------------------- */

/* One frame, accumulate into ps[N] exactly as src/spectrum.c:15-35 does:
 * ps[i] += |X[(i+N/2)%N]|^2 for (i+N/2)%N != 0, and the slot whose bin index
 * is 0 does ps[i] += ps[i-1] (already updated neighbour).
 * Returns 0, or -1 when len != N (src/spectrum.c:51-52).
 * window may be NULL (reference behaviour) or N f64 weights applied to the
 * converted samples (build extension, not in the reference). */
int orc_spectrum_add_cmplx_u8(int N, const uint8_t* src_iq, const double* window,
                              double* ps, int len);
int orc_spectrum_add_cmplx_s32(int N, const int32_t* src_iq, const double* window,
                               double* ps, int len);
int orc_spectrum_add_real_f32(int N, const float* src, const double* window,
                              double* ps, int len);

/* Batch: nframes frames of N cmplx_u8 each, grouped K consecutive frames per
 * output row; every row starts from zero and receives K sequential
 * orc_spectrum_add_cmplx_u8 calls (the loop of src/cbb_main.c:50-59).
 * out is [nframes/K][N] f64.  nthreads >= 1 splits rows across pthreads. */
int orc_batch_spectra_u8(int N, int K, long nframes, const uint8_t* src_iq,
                         const double* window, double* out, int nthreads);

/* Same, with a CIC (block sum of R samples, gain R) in front: each FFT input
 * sample is sum_{i<R}(u8-128) and is converted like spectrum_add_cmplx_s32
 * (value/128, no offset).  src holds nframes*N*R cmplx_u8. */
int orc_batch_spectra_cic_u8(int N, int K, int R, long nframes, const uint8_t* src_iq,
                             const double* window, double* out, int nthreads);

/* ---- resample.c semantics ------------------------------------------- */

/* state = {integrator_prev_out.re, .im, comb_prev_in.re, .im}; int32 wraps. */
int orc_cic_decimate(int R, const uint8_t* src_iq, int src_len, int32_t* dst_iq,
                     int dst_len, int32_t state[4]);

/* 11-tap half-band 2:1, f32, same operation order as src/resample.c:53-66
 * (compile this file with -ffp-contract=off to keep mul and add separate). */
void orc_halfband_decimate(const float* input, float* output, int output_len,
                           float* delay10);

/* ---- audio front end ---------------------------- */

/* src/common_sp.h:40-76, f32 with the double-precision +-M_PI of the original. */
float orc_atan2_approx(float y, float x);

/* src/audio_main.c:110-131: phase = atan2_approx(im, re); out = phase - prev;
 * hard limit to [-1, 1]; *prev carries the last phase across calls. */
void orc_fm_demod(const int32_t* iq, int len, float* prev_phase, float* out);

/* Whole chain of src/audio_main.c:74-145 for one decimator block of `len`
 * samples: FM demod, half-band 2:1, half-band 2:1 -> len/4 audio samples.
 * state = {prev_phase, delay1[10], delay2[10]} (21 floats, zero-initialised). */
void orc_audio_block(const int32_t* iq, int len, float state[21], float* audio_out);

/* ---- rf_decimator.c semantics --------------------------------------- */

struct orc_rfdec;
typedef void (*orc_rfdec_cb)(const int32_t* iq, int len, void* user);
struct orc_rfdec* orc_rfdec_new(void);
int orc_rfdec_set_parameters(struct orc_rfdec* d, double sample_rate, int down_factor);
int orc_rfdec_decimate(struct orc_rfdec* d, const uint8_t* iq, int len,
                       orc_rfdec_cb cb, void* user);
int orc_rfdec_input_len(const struct orc_rfdec* d);
int orc_rfdec_resampled_len(const struct orc_rfdec* d);
void orc_rfdec_free(struct orc_rfdec* d);

/* ---- cbb_main.c semantics ------------------------------------------- */

/* src/cbb_main.c:106-135.  Returns bytes written: N when count > 0 else 0. */
int orc_spectrum_payload(int N, const double* ps, int count, int spectrum_gain_db,
                         uint8_t* buf);

/* src/cbb_main.c:40-70 without the time gate: blocks = min(len/1024, 6),
 * zero, add each block, return blocks. ps is double[1024]. */
int orc_estimate_spectrum(const uint8_t* iq, int len, double* ps);

/* f64 mean-power dB (build extension used by config 3): 10*log10(ps/count). */
void orc_mean_db(int N, const double* ps, int count, double* db);

#ifdef __cplusplus
}
#endif
#endif
