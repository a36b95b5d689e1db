/*
 * audio_main.h -- the FM-demodulator front end of the audio chain, declaration-
 * compatible with the reference (reference src/audio_main.h:6-14,
 * src/audio_main.c:1-161).  SURVEY.md §8f row 3: a "next" row widened into after
 * the spectrum path met its bar; the audio chain is not on the IQ -> spectrum
 * path and moves ~200 kS/s, so this is a completeness row, not a roofline one.
 *
 * audio_fm_demodulator is an rf_decimator callback (reference src/main.c:205):
 * per decimated block of `len` cmplx_s32 it computes atan2_approx phase, first
 * difference, hard limit (src/audio_main.c:110-131), then two 11-tap half-band
 * 2:1 decimators (:133,139) -- three kernels chained on the device, delay lines
 * resident there -- and queues len/4 float samples.
 *
 * audio_get_audio_payload hands the queued floats out in order.  It does NOT
 * reproduce two defects of the reference's version (src/audio_main.c:40-72: the
 * first buffer is delivered twice, and reads spanning two buffers land at a
 * byte offset computed from a sample count): a plain FIFO is what the browser
 * code (resources/rtl_ui.js:34-69) assumes.
 */
#ifndef AUDIO_MAIN_H
#define AUDIO_MAIN_H

#include "common_sp.h"

#ifdef __cplusplus
extern "C" {
#endif

void audio_init(void);                                        /* src/audio_main.h:6  */
int audio_new_audio_available(void);                          /* :8  */
int audio_get_audio_payload(char* buf, int buf_len);          /* :10: returns bytes written */
void audio_fm_demodulator(const cmplx_s32* signal, int len);  /* :12 */
void audio_close(void);                                       /* :14 */

#ifdef __cplusplus
}
#endif
#endif /* AUDIO_MAIN_H */
