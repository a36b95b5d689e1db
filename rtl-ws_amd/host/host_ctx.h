/* host_ctx.h -- process-wide device context shared by the handle-less entry
 * points (cic_decimate, halfband_decimate) and by rf_decimator.  Plain C; the
 * GPU is reached only through include/rtlws_hip.h. */
#ifndef RTLWS_HOST_CTX_H
#define RTLWS_HOST_CTX_H

#include <stddef.h>
#include <stdint.h>
#include "rtlws_hip.h"
#include "rtlws_host.h"
#include "resample.h"

/* Record a failure of an entry point that cannot return one (include/rtlws_host.h):
 * sticky first message, counter, stderr for the first and every 1024th. */
void rtlws_host_fail(const char* where, const char* what);

/* Device index used by the drop-in entry points: $RTLWS_DEVICE or 0. */
int rtlws_host_device(void);

/* Lazily created engine + growable staging buffers, serialised by a mutex.
 * Returns NULL when no HIP device is usable (callers then fail loudly). */
struct rtlws_host_ctx;
struct rtlws_host_ctx* rtlws_host_ctx_get(void);

/* CIC of one block on the GPU plus the reference's delay-line bookkeeping
 * (reference src/resample.c:15-16,35-36,42-43).  Returns 0, -1 (length
 * mismatch), -3 (device failure). */
int rtlws_host_cic(int R, const cmplx_u8* src, int src_len, cmplx_s32* dst, int dst_len,
                   struct cic_delay_line* delay);

#endif
