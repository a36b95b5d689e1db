/* synth_sensor.c -- a synthetic implementation of rtl_sensor.h for
 * BASELINE.json configs[0] ("CPU plumbing": no dongle, no librtlsdr).
 *
 * The reference's own stub (REAL_SENSOR undefined) returns from
 * rtl_read_async immediately and never delivers data (reference
 * src/rtl_sensor.c:146-153), so end-to-end runs need a source that behaves
 * like the dongle: blocks on the calling thread and hands out buffers of
 * interleaved u8 I/Q, paced at the sample rate, until rtl_cancel().
 *
 * Data: $RTLWS_SYNTH_FILE (raw u8 IQ, replayed cyclically) or, if unset, a
 * generated tone + noise.  $RTLWS_SYNTH_BUFLEN bytes per buffer (default
 * 262144 = librtlsdr's default), $RTLWS_SYNTH_SPEEDUP (default 1.0; 0 = no
 * pacing), $RTLWS_SYNTH_MAXBUFS (stop delivering after that many, then idle
 * until cancelled).
 */
#include "rtl_sensor.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct rtl_dev {
    uint32_t f;
    uint32_t fs;
    double gain;
    int cancel;                    /* set by rtl_cancel on another thread: accessed atomically */
};

int rtl_init(struct rtl_dev** dev, int dev_index)
{
    (void)dev_index;
    *dev = (struct rtl_dev*)calloc(1, sizeof(struct rtl_dev));
    if (!*dev) return -1;
    (*dev)->fs = 2048000;          /* reference defaults, src/rtl_sensor.c:12-14 */
    (*dev)->f = 100000000;
    (*dev)->gain = 25.4;
    return 0;
}

int rtl_set_frequency(struct rtl_dev* dev, uint32_t f) { dev->f = f; return 0; }
int rtl_set_sample_rate(struct rtl_dev* dev, uint32_t fs) { dev->fs = fs; return 0; }
int rtl_set_gain(struct rtl_dev* dev, double gain) { dev->gain = gain; return 0; }
uint32_t rtl_freq(const struct rtl_dev* dev) { return dev->f; }
uint32_t rtl_sample_rate(const struct rtl_dev* dev) { return dev->fs; }
double rtl_gain(const struct rtl_dev* dev) { return dev->gain; }

static double env_double(const char* name, double dflt)
{
    const char* s = getenv(name);
    return s ? atof(s) : dflt;
}

static unsigned char* make_signal(size_t* nbytes, size_t buflen)
{
    const char* path = getenv("RTLWS_SYNTH_FILE");
    unsigned char* data;
    if (path) {
        FILE* f = fopen(path, "rb");
        long sz;
        if (!f) { fprintf(stderr, "synth_sensor: cannot open %s\n", path); return NULL; }
        fseek(f, 0, SEEK_END);
        sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz < 2) { fclose(f); return NULL; }
        data = (unsigned char*)malloc((size_t)sz);
        if (!data) { fclose(f); return NULL; }
        if (fread(data, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(data); return NULL; }
        fclose(f);
        *nbytes = (size_t)sz - ((size_t)sz & 1u);
        return data;
    }
    {   /* tone at fs/8 offset, amplitude 0.6, plus uniform dither noise */
        size_t n = buflen / 2, i;
        uint32_t x = 2463534242u;
        data = (unsigned char*)malloc(buflen);
        if (!data) return NULL;
        for (i = 0; i < n; i++) {
            double ph = 2.0 * 3.14159265358979323846 * 0.125 * (double)i;
            double nr, ni;
            x ^= x << 13; x ^= x >> 17; x ^= x << 5; nr = ((double)(x & 0xffff) / 65536.0 - 0.5) * 0.2;
            x ^= x << 13; x ^= x >> 17; x ^= x << 5; ni = ((double)(x & 0xffff) / 65536.0 - 0.5) * 0.2;
            {
                double re = 0.6 * cos(ph) + nr, im = 0.6 * sin(ph) + ni;
                double qr = floor(re * 128.0 + 128.0 + 0.5), qi = floor(im * 128.0 + 128.0 + 0.5);
                data[2 * i] = (unsigned char)(qr < 0 ? 0 : qr > 255 ? 255 : qr);
                data[2 * i + 1] = (unsigned char)(qi < 0 ? 0 : qi > 255 ? 255 : qi);
            }
        }
        *nbytes = buflen;
        return data;
    }
}

int rtl_read_async(struct rtl_dev* dev, void (*callback)(unsigned char*, uint32_t, void*), void* user)
{
    size_t buflen = (size_t)env_double("RTLWS_SYNTH_BUFLEN", 262144.0);
    const double speedup = env_double("RTLWS_SYNTH_SPEEDUP", 1.0);
    const long maxbufs = (long)env_double("RTLWS_SYNTH_MAXBUFS", 0.0);
    size_t nbytes = 0, pos = 0;
    unsigned char *data, *buf;
    long delivered = 0;
    struct timespec next;

    buflen -= buflen & 1u;
    if (buflen < 2) return -1;
    data = make_signal(&nbytes, buflen);
    if (!data) return -1;
    buf = (unsigned char*)malloc(buflen);
    if (!buf) { free(data); return -1; }
    clock_gettime(CLOCK_MONOTONIC, &next);

    while (!__atomic_load_n(&dev->cancel, __ATOMIC_ACQUIRE)) {
        if (maxbufs > 0 && delivered >= maxbufs) {       /* drained: idle until cancelled */
            struct timespec nap = {0, 2000000};
            nanosleep(&nap, NULL);
            continue;
        }
        {   /* next buffer, wrapping around the recording */
            size_t done = 0;
            while (done < buflen) {
                size_t chunk = nbytes - pos < buflen - done ? nbytes - pos : buflen - done;
                memcpy(buf + done, data + pos, chunk);
                done += chunk;
                pos = (pos + chunk) % nbytes;
            }
        }
        if (speedup > 0.0) {                             /* pace like the dongle would */
            const double dt = ((double)(buflen / 2) / (double)dev->fs) / speedup;
            next.tv_nsec += (long)((dt - floor(dt)) * 1e9);
            next.tv_sec += (time_t)floor(dt) + next.tv_nsec / 1000000000L;
            next.tv_nsec %= 1000000000L;
            clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &next, NULL);
        }
        if (__atomic_load_n(&dev->cancel, __ATOMIC_ACQUIRE)) break;
        callback(buf, (uint32_t)buflen, user);
        delivered++;
    }
    free(buf);
    free(data);
    return 0;
}

void rtl_cancel(struct rtl_dev* dev) { __atomic_store_n(&dev->cancel, 1, __ATOMIC_RELEASE); }

void rtl_close(struct rtl_dev* dev) { free(dev); }
