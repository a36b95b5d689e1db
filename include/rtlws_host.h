/*
 * rtlws_host.h -- failure reporting of the drop-in host layer (librtlws_amd / librtlws_cbb).
 *
 * Several entry points of the reference API cannot report a failure through their
 * signature: halfband_decimate, cbb_init, audio_init and audio_fm_demodulator return void
 * (src/resample.h:17, src/cbb_main.c:72, src/audio_main.c:53,106).  There is no CPU
 * implementation behind this library, so when the device is missing or a launch fails
 * those functions do NOT compute anything -- but they do not take the host process (the
 * WebSocket server) down either.  They
 *   - produce a DEFINED result: halfband_decimate writes zeros (silence) and still
 *     advances the caller's delay line; audio_fm_demodulator queues nothing for that
 *     block; cbb_init leaves the spectrum side inert (cbb_new_spectrum_available() stays
 *     0, cbb_get_spectrum_payload() returns 0 bytes, the sample counter and the
 *     decimator callback still run; cic_decimate underneath returns -3 as before);
 *   - record the failure here, STICKY: the first message is kept, every failure counted,
 *     the first (and every 1024th) also goes to stderr.
 * Entry points that can return an error still do (spectrum_alloc NULL, spectrum_add_* -3,
 * cic_decimate -3, rf_decimator_decimate_cmplx_u8 -2).
 */
#ifndef RTLWS_HOST_H
#define RTLWS_HOST_H

#ifdef __cplusplus
extern "C" {
#endif

/* Text of the FIRST failure since start (or since the last clear); "" when none. */
const char* rtlws_host_error(void);
/* Number of failures recorded since start (or since the last clear). */
long rtlws_host_error_count(void);
void rtlws_host_error_clear(void);

/* For companion libraries built on the host layer (librtlws_cbb.so uses both): record a failure of an entry point
 * that cannot return one -- sticky first message, counter, stderr for the first and every 1024th -- and the device
 * index the drop-in entry points use ($RTLWS_DEVICE, read once; default 0). */
void rtlws_host_fail(const char* where, const char* what);
int rtlws_host_device(void);

#ifdef __cplusplus
}
#endif
#endif /* RTLWS_HOST_H */
