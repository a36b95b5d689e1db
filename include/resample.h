/*
 * resample.h -- drop-in boundary #1b: the two decimators of the reference
 * (reference src/resample.h:6-17, src/resample.c:4-67), served by HIP kernels
 * in rtl-ws_amd/csrc/resample_kernels.hip.
 *
 * cic_decimate: first-order CIC == exact int32 sums of R consecutive
 *   (u8 - 128) samples per component (src/resample.c:21-40).  The public
 *   delay-line struct is kept byte-compatible and is updated exactly as the
 *   reference updates it (running int32 integrator, wrapping), so callers
 *   that chain calls or inspect the state see identical values.
 *   Returns 0, or -1 when dst_len * R != src_len (src/resample.c:18-19);
 *   -3 (ours) when the device call fails.
 *
 * halfband_decimate: 11-tap half-band 2:1 on real f32 with a 10-sample delay
 *   line (src/resample.c:47-67); products and sums are taken in the
 *   reference's source order without fused multiply-add.
 */
#ifndef RESAMPLE_H
#define RESAMPLE_H

#include "common_sp.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HALF_BAND_N 11                    /* reference src/resample.h:6 */

struct cic_delay_line {                   /* reference src/resample.h:8-12 */
    cmplx_s32 integrator_prev_out;
    cmplx_s32 comb_prev_in;
};

/* reference src/resample.h:14 */
int cic_decimate(int R, const cmplx_u8* src, int src_len, cmplx_s32* dst, int dst_len,
                 struct cic_delay_line* delay);

/* reference src/resample.h:17 -- `delay` holds HALF_BAND_N - 1 floats */
void halfband_decimate(const float* input, float* output, int output_len, float* delay);

#ifdef __cplusplus
}
#endif
#endif /* RESAMPLE_H */
