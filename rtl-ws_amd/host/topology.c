/* topology.c -- rtlws_topo.h: device -> PCI bus id -> NUMA node -> cpuset, and thread pinning. */
#define _GNU_SOURCE
#include "rtlws_topo.h"

#include <ctype.h>
#include <errno.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtlws_hip.h"
#include "topology.h"

static int read_line(const char* path, char* buf, size_t len)
{
    FILE* f = fopen(path, "r");
    size_t n;
    if (!f) return -1;
    if (!fgets(buf, (int)len, f)) { fclose(f); return -1; }
    fclose(f);
    n = strlen(buf);
    while (n && (buf[n - 1] == '\n' || buf[n - 1] == ' ' || buf[n - 1] == '\r')) buf[--n] = 0;
    return 0;
}

int rtlws_topo_parse_cpulist(const char* list, unsigned char* cpus, int max)
{
    const char* p = list;
    int count = 0;
    if (!list || !cpus || max < 1) return -1;
    memset(cpus, 0, (size_t)max);
    while (*p) {
        long a, b;
        char* end;
        while (*p == ' ' || *p == ',') ++p;
        if (!*p) break;
        if (!isdigit((unsigned char)*p)) return -1;
        errno = 0;
        a = strtol(p, &end, 10);
        if (errno == ERANGE) return -1;
        b = a;
        p = end;
        if (*p == '-') {
            ++p;
            if (!isdigit((unsigned char)*p)) return -1;
            b = strtol(p, &end, 10);
            if (errno == ERANGE) return -1;
            p = end;
        }
        if (b < a) return -1;
        if (*p && *p != ',' && *p != ' ' && *p != '\n') return -1;
        /* CPUs from `max` on are not representable: ignored, never iterated over (a list such as
         * "0-9223372036854775807" from a corrupt or caller-supplied sysfs tree must not spin) */
        if (b >= max) b = (long)max - 1;
        for (; a <= b; ++a)
            if (!cpus[a]) { cpus[a] = 1; ++count; }
    }
    return count;
}

int rtlws_topo_describe(int device, const char* bus_id, const char* sysfs_root, rtlws_topo_info* out)
{
    char path[640], line[512];
    unsigned char cpus[RTLWS_TOPO_MAX_CPUS];
    const char* root = (sysfs_root && *sysfs_root) ? sysfs_root : "/sys";
    size_t i;
    if (!out) return -1;
    memset(out, 0, sizeof *out);
    out->device = device;
    out->numa_node = -1;
    if (bus_id) {
        if (strlen(bus_id) >= sizeof out->bus_id) return -1;
        strcpy(out->bus_id, bus_id);
    } else if (device < 0 || rtlws_device_pci_bus_id(device, out->bus_id, (int)sizeof out->bus_id) != 0) {
        out->bus_id[0] = 0;                     /* no device to ask: nothing is known, nothing is pinned */
        return device < 0 ? -1 : 0;
    }
    for (i = 0; out->bus_id[i]; ++i) out->bus_id[i] = (char)tolower((unsigned char)out->bus_id[i]);   /* sysfs spells hex in lower case */
    if (strchr(out->bus_id, '/') || strstr(out->bus_id, "..")) return -1;
    snprintf(path, sizeof path, "%s/bus/pci/devices/%s/numa_node", root, out->bus_id);
    if (read_line(path, line, sizeof line) == 0) out->numa_node = atoi(line);
    if (out->numa_node < 0) out->numa_node = -1;
    line[0] = 0;
    if (out->numa_node >= 0) {
        snprintf(path, sizeof path, "%s/devices/system/node/node%d/cpulist", root, out->numa_node);
        if (read_line(path, line, sizeof line) != 0) line[0] = 0;
    }
    if (!line[0]) {
        snprintf(path, sizeof path, "%s/bus/pci/devices/%s/local_cpulist", root, out->bus_id);
        if (read_line(path, line, sizeof line) != 0) line[0] = 0;
    }
    if (line[0] && strlen(line) < sizeof out->cpulist) {
        const int n = rtlws_topo_parse_cpulist(line, cpus, RTLWS_TOPO_MAX_CPUS);
        if (n > 0) {
            strcpy(out->cpulist, line);
            out->ncpus = n;
        }
    }
    return 0;
}

int rtlws_topo_pin_save(const rtlws_topo_info* info, cpu_set_t* saved, int* have_saved)
{
    unsigned char cpus[RTLWS_TOPO_MAX_CPUS];
    cpu_set_t now, want;
    int c, n = 0;
    if (have_saved) *have_saved = 0;
    if (!info || !info->cpulist[0]) return 0;
    if (rtlws_topo_parse_cpulist(info->cpulist, cpus, RTLWS_TOPO_MAX_CPUS) <= 0) return 0;
    if (pthread_getaffinity_np(pthread_self(), sizeof now, &now) != 0) return -1;
    CPU_ZERO(&want);
    for (c = 0; c < RTLWS_TOPO_MAX_CPUS && c < CPU_SETSIZE; ++c)
        if (cpus[c] && CPU_ISSET(c, &now)) { CPU_SET(c, &want); ++n; }
    if (n == 0) return 0;                       /* the node's CPUs are outside this job's mask: leave it */
    if (pthread_setaffinity_np(pthread_self(), sizeof want, &want) != 0) return -1;
    if (saved) *saved = now;
    if (have_saved) *have_saved = 1;
    return n;
}

void rtlws_topo_restore(const cpu_set_t* saved, int have_saved)
{
    if (have_saved && saved) (void)pthread_setaffinity_np(pthread_self(), sizeof *saved, saved);
}

int rtlws_topo_pin_thread(const rtlws_topo_info* info) { return rtlws_topo_pin_save(info, NULL, NULL); }
