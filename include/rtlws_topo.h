/*
 * rtlws_topo.h -- which CPUs sit next to a GPU: device -> PCI bus id -> NUMA node -> cpuset.
 *
 * The multi-device drivers (rtlws_multi.h: one host thread per shard; rtlws_stream.h: one worker thread
 * per stream, pinned ring slots) feed their device from host memory.  On a two-socket node half the
 * GPUs hang off each socket: a thread that runs -- and a pinned buffer that was first touched -- on the
 * other socket crosses the inter-socket link with every byte of the ~45 GB/s a device's copy engine
 * moves.  So each such thread pins itself to the CPUs of its device's NUMA node BEFORE
 * its first allocation, and allocates its pinned memory itself.  The reference has no counterpart (one
 * dongle, one thread: src/signal_source.c:29-35); this belongs to SURVEY.md §8e.
 *
 * Everything here is host-side C; the only GPU question asked is the device's PCI bus id
 * (rtlws_device_pci_bus_id, rtlws_hip.h).  The sysfs root is a parameter so that the mapping is testable
 * against a fake tree without a GPU (tests/test_topo_cpu.py) and plannable offline.
 */
#ifndef RTLWS_TOPO_H
#define RTLWS_TOPO_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtlws_topo_info {
    int device;          /* HIP device index asked for (-1: none, a bus id was given) */
    char bus_id[32];     /* "0000:05:00.0" -- domain:bus:device.function, as sysfs spells it; "" unknown */
    int numa_node;       /* the device's NUMA node; -1: unknown, or the platform reports none */
    int ncpus;           /* CPUs in `cpulist` (0: unknown) */
    char cpulist[512];   /* the CPUs local to the device in the kernel's list format, "0-31,128-159" */
} rtlws_topo_info;

/* Fill *out for HIP device `device`.  bus_id: NULL = ask the HIP runtime (needs the device), else use this
 * one (planning for another host, tests).  sysfs_root: NULL = "/sys".  Reads
 *   <root>/bus/pci/devices/<bus_id>/numa_node                 the node (-1 when the firmware names none)
 *   <root>/devices/system/node/node<N>/cpulist                its CPUs
 *   <root>/bus/pci/devices/<bus_id>/local_cpulist             the fallback when there is no node
 * Returns 0 with whatever could be learnt (unknown fields as documented above): a host without NUMA
 * information is not an error, the callers then simply do not pin.  -1: bad arguments. */
int rtlws_topo_describe(int device, const char* bus_id, const char* sysfs_root, rtlws_topo_info* out);

/* Parse a kernel cpu list ("0-3,8,10-11") into cpus[0 .. max): 1 for a listed CPU.  Returns how many
 * are listed (CPUs >= max are ignored), -1 for a malformed list. */
int rtlws_topo_parse_cpulist(const char* list, unsigned char* cpus, int max);

/* Restrict the CALLING thread to info->cpulist intersected with the CPUs it may run on now.  Returns the
 * number of CPUs it is then pinned to; 0 = nothing done (no list, or the intersection is empty: the
 * thread keeps its mask); -1 = the system call failed. */
int rtlws_topo_pin_thread(const rtlws_topo_info* info);

#ifdef __cplusplus
}
#endif
#endif /* RTLWS_TOPO_H */
