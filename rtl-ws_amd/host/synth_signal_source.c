/* synth_signal_source.c -- a minimal signal_source.h for end-to-end runs
 * without the reference tree: one worker thread blocking in rtl_read_async,
 * every buffer handed to each registered callback in order, under a mutex
 * (behaviour of reference src/signal_source.c:29-97).
 */
#include "signal_source.h"

#include <pthread.h>
#include <stdlib.h>

#define MAX_CALLBACKS 16

static pthread_t g_worker;
static pthread_mutex_t g_cb_mu = PTHREAD_MUTEX_INITIALIZER;
static signal_source_callback g_cbs[MAX_CALLBACKS];
static int g_ncbs = 0;
static struct rtl_dev* g_sensor = NULL;
static int g_running = 0;           /* start / stop come from one thread; the worker reads it once, after its creation */

static void on_buffer(unsigned char* buf, uint32_t len_bytes, void* user)
{
    int i;
    (void)user;
    pthread_mutex_lock(&g_cb_mu);
    for (i = 0; i < g_ncbs; i++) g_cbs[i]((const cmplx_u8*)buf, (int)(len_bytes / 2));
    pthread_mutex_unlock(&g_cb_mu);
}

static void* worker_main(void* arg)
{
    if (g_running) rtl_read_async((struct rtl_dev*)arg, on_buffer, NULL);
    return NULL;
}

void signal_source_start(struct rtl_dev* dev)
{
    if (g_running) return;
    g_running = 1;
    g_sensor = dev;
    g_ncbs = 0;
    pthread_create(&g_worker, NULL, worker_main, dev);
}

void signal_source_add_callback(signal_source_callback callback)
{
    pthread_mutex_lock(&g_cb_mu);
    if (g_ncbs < MAX_CALLBACKS) g_cbs[g_ncbs++] = callback;
    pthread_mutex_unlock(&g_cb_mu);
}

void signal_source_remove_callbacks(void)
{
    pthread_mutex_lock(&g_cb_mu);
    g_ncbs = 0;
    pthread_mutex_unlock(&g_cb_mu);
}

void signal_source_stop(void)
{
    if (!g_running) return;
    g_running = 0;
    rtl_cancel(g_sensor);
    pthread_join(g_worker, NULL);
}
