/*
 * spectrum.h -- drop-in boundary #1: the power-spectrum estimator.
 *
 * Same five entry points, argument meaning and return codes as the reference
 * (reference src/spectrum.h:7-17, implemented by src/spectrum.c:37-107), but
 * implemented by librtlws_amd.so on an MI355X: u8/s32/f32 -> complex DOUBLE,
 * N-point forward DFT, |X|^2 and fft-shift all run in one HIP kernel in the
 * reference's own precision (rtl-ws_amd/csrc/spectrum_f64.hip, reached through
 * rtlws_spectra_batch_f64 of rtlws_hip.h); only the accumulation into the
 * caller's host `double` buffer and the order-dependent DC-slot rule
 * (src/spectrum.c:25-33) stay on the calling thread, because that buffer is
 * host memory owned by the caller.
 *
 * Contract kept from the reference:
 *   - spectrum_alloc(N): one handle per estimator; no thread safety inside a
 *     handle (one caller at a time, src/cbb_main.c:54).
 *   - spectrum_add_*: returns 0, or -1 when len != N (src/spectrum.c:51-52,
 *     69-70, 87-88).  `power_spectrum` is read-modify-write: results are
 *     ADDED to what the caller left there (the caller zeroes it,
 *     src/cbb_main.c:50).  Output is fft-shifted, unnormalised, and slot N/2
 *     receives the running value of slot N/2-1 instead of the DC bin.
 *   - Synchronous: the result is in `power_spectrum` on return.
 *
 * Differences, stated so a maintainer is not surprised:
 *   - 2 <= N <= 8192 (a frame lives in the 160 KiB LDS of one compute unit as
 *     complex doubles): the fused radix-16 kernel at N = 1024 / 2048 / 4096, radix-2
 *     for the other powers of two, a direct O(N^2) sum for any other N.  There is no CPU path: if no HIP device is usable, or N is out of
 *     range, spectrum_alloc returns NULL (the reference never reports failure).
 *   - Arithmetic is f64 on the device, like the reference (f64 via FFTW):
 *     increments agree with an f64 FFT to <= 1e-10 relative per bin under the
 *     strict metric (floor 1e-9 of the row maximum), K = 1 included
 *     (tests/test_spectrum_gpu.py::test_dropin_*).  The summation order inside
 *     the transform differs from FFTW's, so the last bits do.
 *
 * For throughput use the batch API in rtlws_hip.h (rtlws_spectra_batch: f32
 * arithmetic, device-resident frames, its own stated error budget): this one
 * moves at most N samples per call across PCIe.
 */
#ifndef SPECTRUM_H
#define SPECTRUM_H

#include <stdint.h>
#include "common_sp.h"

#ifdef __cplusplus
extern "C" {
#endif

struct spectrum;

/* reference src/spectrum.h:9 / src/spectrum.c:37-45 */
struct spectrum* spectrum_alloc(int N);

/* reference src/spectrum.h:11 / src/spectrum.c:47-63: in = (u8 - 128) / 128 */
int spectrum_add_cmplx_u8(struct spectrum* s, const cmplx_u8* src, double* power_spectrum, int len);

/* reference src/spectrum.h:13 / src/spectrum.c:65-81: in = s32 / 128 (no offset) */
int spectrum_add_cmplx_s32(struct spectrum* s, const cmplx_s32* src, double* power_spectrum, int len);

/* reference src/spectrum.h:15 / src/spectrum.c:83-99: in = (f32, 0) */
int spectrum_add_real_f32(struct spectrum* s, const float* src, double* power_spectrum, int len);

/* reference src/spectrum.h:17 / src/spectrum.c:101-107 */
void spectrum_free(struct spectrum* s);

#ifdef __cplusplus
}
#endif
#endif /* SPECTRUM_H */
