/*
 * signal_source.h -- worker thread + callback fan-out, declaration-compatible
 * with the reference (reference src/signal_source.h:7-16).  Out of scope for
 * the engine; rtl-ws_amd/host/synth_signal_source.c supplies a
 * minimal implementation so that cbb_main can be exercised end to end without
 * the reference tree.  Callbacks run serially on the worker thread, one round
 * per sensor buffer (reference src/signal_source.c:29-35).
 */
#ifndef SIGNAL_SOURCE_H
#define SIGNAL_SOURCE_H

#include "common_sp.h"
#include "rtl_sensor.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void (*signal_source_callback)(const cmplx_u8*, int);     /* src/signal_source.h:7 */

void signal_source_start(struct rtl_dev* dev);                     /* :10 */
void signal_source_add_callback(signal_source_callback callback);  /* :12 */
void signal_source_remove_callbacks(void);                         /* :14 */
void signal_source_stop(void);                                     /* :16 */

#ifdef __cplusplus
}
#endif
#endif /* SIGNAL_SOURCE_H */
