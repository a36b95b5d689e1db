/*
 * rtl_sensor.h -- the sensor seam, declaration-compatible with the reference
 * (reference src/rtl_sensor.h:9-27).  The engine does not implement the
 * dongle wrapper (out of scope); this header exists because
 * cbb_main (boundary #2) is written against it and because the synthetic
 * source for BASELINE.json configs[0] (rtl-ws_amd/host/synth_sensor.c)
 * implements exactly this interface: rtl_read_async() blocks on the calling
 * thread and delivers buffers of interleaved u8 I/Q (len bytes, len/2 complex
 * samples, reference src/signal_source.c:29-35) until rtl_cancel().
 */
#ifndef RTL_SENSOR_H
#define RTL_SENSOR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct rtl_dev;

int rtl_init(struct rtl_dev** dev, int dev_index);                 /* src/rtl_sensor.h:9  */
int rtl_set_frequency(struct rtl_dev* dev, uint32_t f);            /* :11 */
int rtl_set_sample_rate(struct rtl_dev* dev, uint32_t fs);         /* :13 */
int rtl_set_gain(struct rtl_dev* dev, double gain);                /* :15 */
uint32_t rtl_freq(const struct rtl_dev* dev);                      /* :17 */
uint32_t rtl_sample_rate(const struct rtl_dev* dev);               /* :19 */
double rtl_gain(const struct rtl_dev* dev);                        /* :21 */
int rtl_read_async(struct rtl_dev* dev, void (*callback)(unsigned char*, uint32_t, void*),
                   void* user);                                    /* :23 */
void rtl_cancel(struct rtl_dev* dev);                              /* :25 */
void rtl_close(struct rtl_dev* dev);                               /* :27 */

#ifdef __cplusplus
}
#endif
#endif /* RTL_SENSOR_H */
