/*
 * cbb_main.h -- drop-in boundary #2: what the untouched server (reference
 * src/main.c) calls (reference src/cbb_main.h:7-17, src/cbb_main.c:72-153).
 *
 *   cbb_init(decimated_bw_target_hz)  -- note: the reference header declares
 *       `void cbb_init();` but defines and calls it with one int
 *       (src/cbb_main.c:72, src/main.c:203); the int form is exported.
 *       Opens the sensor, configures the decimator with R = fs / target
 *       (integer division, src/cbb_main.c:80), creates the GPU spectrum engine,
 *       starts the signal source and registers the per-buffer callbacks.
 *   cbb_new_spectrum_available()      -- src/cbb_main.c:101-104
 *   cbb_get_spectrum_payload(buf, buf_len, gain_db)
 *       -- src/cbb_main.c:106-135: average of the accumulated frames, gain in
 *       10 dB steps (C integer division), 10*log10, truncate, clamp to
 *       [0,255]; writes exactly 1024 bytes when a spectrum of >= 1 frame is
 *       held, else 0; returns the byte count.  Unlike the reference it never
 *       writes more than buf_len bytes.
 *   cbb_rf_decimator(), cbb_get_rtl_dev(), cbb_close()
 *
 * Per sensor buffer the engine does what src/cbb_main.c:40-70 does: at most
 * every 250 ms, the first min(len/1024, 6) frames of the buffer become ONE
 * launch of the f64 kernel (K = blocks; the reference keeps doubles,
 * src/cbb_main.c:27); the 1024 double sums stay on the device and the dB/clamp
 * conversion, in double, is a second small kernel at payload time.  The sensor
 * thread never waits for the device.
 *
 * Extensions, chosen by RTLWS_CBB_ALL_FRAMES in the environment at cbb_init
 * (same payload format, same calls):
 *   1  every whole frame of the buffer that passes the 250 ms gate is averaged
 *      (128 at librtlsdr's default buffer size) instead of its first 6;
 *   2  Welch averaging over the whole interval: EVERY buffer is transformed, all
 *      frames of it, and the gate only decides when the running average is
 *      published -- no sample of the stream is thrown away.  The published row
 *      is exactly what the reference's loop would leave after that many frames.
 */
#ifndef CBB_MAIN_H
#define CBB_MAIN_H

#include "rf_decimator.h"
#include "rtl_sensor.h"

#ifdef __cplusplus
extern "C" {
#endif

void cbb_init(int decimated_bw_target_hz);                                   /* src/cbb_main.h:7 */
struct rf_decimator* cbb_rf_decimator(void);                                 /* :9  */
struct rtl_dev* cbb_get_rtl_dev(void);                                       /* :11 */
int cbb_new_spectrum_available(void);                                        /* :13 */
int cbb_get_spectrum_payload(char* buf, int buf_len, int spectrum_gain_db);  /* :15 */
void cbb_close(void);                                                        /* :17 */

#ifdef __cplusplus
}
#endif
#endif /* CBB_MAIN_H */
