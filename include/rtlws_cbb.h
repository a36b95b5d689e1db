/*
 * rtlws_cbb.h -- diagnostic counters of librtlws_cbb.so, beside the reference's own cbb_main.h (which stays as the
 * reference wrote it, src/cbb_main.h:7-17): what tests and an operator's status page read; the server never needs
 * them.
 */
#ifndef RTLWS_CBB_H
#define RTLWS_CBB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Frames averaged into the spectrum currently published (src/cbb_main.c:49: at most 6 in the reference's mode). */
int rtlws_cbb_published_frames(void);
/* IQ samples the sensor callback has seen since cbb_init (src/cbb_main.c:42-44 counts them for the rate logger). */
uint64_t rtlws_cbb_samples_seen(void);

#ifdef __cplusplus
}
#endif
#endif /* RTLWS_CBB_H */
