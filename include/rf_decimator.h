/*
 * rf_decimator.h -- drop-in boundary #1b: the re-blocking CIC front end
 * (reference src/rf_decimator.h:6-21, src/rf_decimator.c:38-136).
 *
 * Behaviour kept: arbitrary-length u8 IQ chunks are re-blocked into 100 ms
 * blocks (resampled_len = (int)(fs / R * 100 / 1000), input_len =
 * resampled_len * R, src/rf_decimator.c:65-66); each full block is decimated
 * by R and handed to every registered callback, synchronously, on the calling
 * thread, while the decimator's mutex is held (src/rf_decimator.c:86,105);
 * the pointer a callback receives is only valid during that callback.
 * Return codes: 0, -1 unconfigured / bad parameters, -2 decimation failed.
 *
 * Ours differs only where the reference has latent defects:
 * the mutex is released on every return path.
 */
#ifndef RF_DECIMATOR_H
#define RF_DECIMATOR_H

#include "common_sp.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void (*rf_decimator_callback)(const cmplx_s32*, int);    /* src/rf_decimator.h:6 */

struct rf_decimator;

struct rf_decimator* rf_decimator_alloc(void);                                             /* :11 */
void rf_decimator_add_callback(struct rf_decimator* d, rf_decimator_callback callback);    /* :13 */
int rf_decimator_set_parameters(struct rf_decimator* d, double sample_rate, int down_factor); /* :15 */
int rf_decimator_decimate_cmplx_u8(struct rf_decimator* d, const cmplx_u8* complex_signal, int len); /* :17 */
void rf_decimator_remove_callbacks(struct rf_decimator* d);                                /* :19 */
void rf_decimator_free(struct rf_decimator* d);                                            /* :21 */

#ifdef __cplusplus
}
#endif
#endif /* RF_DECIMATOR_H */
