/* rf_decimator_gpu.c -- rf_decimator.h (drop-in boundary #1b) over the HIP shim.
 *
 * Replaces reference src/rf_decimator.c:38-136: re-blocks arbitrary-length
 * cmplx_u8 chunks into 100 ms blocks, decimates every full block by R on the
 * GPU (block sums, exact int32) and fans the result out to the registered
 * callbacks on the calling thread with the mutex held.
 */
#include "rf_decimator.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "host_ctx.h"
#include "resample.h"

#define BLOCK_MS 100            /* reference src/rf_decimator.c:10 */
#define RATE_EPS 0.0001         /* reference src/rf_decimator.c:11 */

struct rf_decimator {
    pthread_mutex_t mutex;
    rf_decimator_callback* cbs;
    int ncbs, cap_cbs;

    double sample_rate;
    int down_factor;

    cmplx_u8* block_in;         /* one block being filled */
    int block_in_len;           /* samples per full block  */
    int filled;                 /* samples already in block_in ("surplus") */

    cmplx_s32* block_out;
    int block_out_len;

    struct cic_delay_line delay;
};

struct rf_decimator* rf_decimator_alloc(void)
{
    struct rf_decimator* d = (struct rf_decimator*)calloc(1, sizeof(*d));
    if (d) pthread_mutex_init(&d->mutex, NULL);
    return d;
}

void rf_decimator_add_callback(struct rf_decimator* d, rf_decimator_callback callback)
{
    pthread_mutex_lock(&d->mutex);
    if (d->ncbs == d->cap_cbs) {
        int cap = d->cap_cbs ? 2 * d->cap_cbs : 4;
        rf_decimator_callback* n = (rf_decimator_callback*)realloc(d->cbs, cap * sizeof(*n));
        if (n) { d->cbs = n; d->cap_cbs = cap; }
    }
    if (d->ncbs < d->cap_cbs) d->cbs[d->ncbs++] = callback;
    pthread_mutex_unlock(&d->mutex);
}

int rf_decimator_set_parameters(struct rf_decimator* d, double sample_rate, int down_factor)
{
    int rc = -1;
    pthread_mutex_lock(&d->mutex);
    if (sample_rate > 0 && down_factor > 0) {
        if (fabs(d->sample_rate - sample_rate) > RATE_EPS || d->down_factor != down_factor) {
            /* reference src/rf_decimator.c:63-71 */
            d->sample_rate = sample_rate;
            d->down_factor = down_factor;
            d->block_out_len = (int)((sample_rate / down_factor) * BLOCK_MS / 1000);
            d->block_in_len = d->block_out_len * down_factor;
            d->block_out = (cmplx_s32*)realloc(d->block_out, (size_t)d->block_out_len * sizeof(cmplx_s32) + 8);
            d->block_in = (cmplx_u8*)realloc(d->block_in, (size_t)d->block_in_len * sizeof(cmplx_u8) + 8);
            d->filled = 0;
        }
        rc = 0;
    }
    pthread_mutex_unlock(&d->mutex);
    return rc;
}

int rf_decimator_decimate_cmplx_u8(struct rf_decimator* d, const cmplx_u8* complex_signal, int len)
{
    int pos = 0, rc = 0, i;
    pthread_mutex_lock(&d->mutex);
    if (!d->block_out || !d->block_in || d->block_in_len <= 0) {
        /* unconfigured (reference src/rf_decimator.c:90-91), or a rate so low
         * that a 100 ms block is empty -- the reference would spin forever */
        pthread_mutex_unlock(&d->mutex);
        return -1;
    }
    /* complete and flush blocks while the chunk can fill the current one */
    while (len - pos >= d->block_in_len - d->filled) {
        const int take = d->block_in_len - d->filled;
        memcpy(d->block_in + d->filled, complex_signal + pos, (size_t)take * sizeof(cmplx_u8));
        pos += take;
        if (rtlws_host_cic(d->down_factor, d->block_in, d->block_in_len, d->block_out,
                           d->block_out_len, &d->delay)) {
            rc = -2;                              /* reference src/rf_decimator.c:99-103 */
            break;
        }
        for (i = 0; i < d->ncbs; i++) d->cbs[i](d->block_out, d->block_out_len);
        d->filled = 0;
    }
    if (rc == 0 && pos < len) {                   /* keep the tail for the next call */
        memcpy(d->block_in + d->filled, complex_signal + pos, (size_t)(len - pos) * sizeof(cmplx_u8));
        d->filled += len - pos;
    }
    pthread_mutex_unlock(&d->mutex);
    return rc;
}

void rf_decimator_remove_callbacks(struct rf_decimator* d)
{
    pthread_mutex_lock(&d->mutex);
    d->ncbs = 0;
    pthread_mutex_unlock(&d->mutex);
}

void rf_decimator_free(struct rf_decimator* d)
{
    if (!d) return;
    pthread_mutex_destroy(&d->mutex);
    free(d->cbs);
    free(d->block_in);
    free(d->block_out);
    free(d);
}
