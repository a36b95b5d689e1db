/* topology.h -- internal side of include/rtlws_topo.h: pin the calling thread next to a device for the
 * length of a set-up phase and put its mask back afterwards (rtlws_stream_open_q runs on the CALLER's
 * thread, which is not ours to keep pinned). */
#ifndef RTLWS_TOPOLOGY_H
#define RTLWS_TOPOLOGY_H

#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <sched.h>

#include "rtlws_topo.h"

#define RTLWS_TOPO_MAX_CPUS 4096

/* as rtlws_topo_pin_thread; *saved / *have_saved receive the mask to restore (have_saved = 0: nothing changed) */
int rtlws_topo_pin_save(const rtlws_topo_info* info, cpu_set_t* saved, int* have_saved);
void rtlws_topo_restore(const cpu_set_t* saved, int have_saved);

#endif
