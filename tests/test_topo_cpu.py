"""include/rtlws_topo.h on fake sysfs trees (no GPU): device -> PCI bus id -> NUMA node -> cpuset for 1 / 2 / 8
devices on 1 / 2 nodes, the cpulist parser, thread pinning, and the drivers' --plan-only output.  The reference
has no counterpart (one dongle, one thread: src/signal_source.c:29-35); this belongs to SURVEY.md 8e."""
import ctypes as C
import json
import os
import subprocess
import threading

import pytest


def make_sysfs(root, devices, nodes):
    """devices: {bus_id: node or None}; nodes: {node: cpulist}.  node None: numa_node = -1 + a local_cpulist."""
    for bus, node in devices.items():
        d = root / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % (-1 if node is None else node))
        (d / "local_cpulist").write_text("0-1\n")
    for node, cpus in nodes.items():
        d = root / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    return str(root)


EIGHT = ["0000:%02x:00.0" % b for b in (0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5)]


def test_cpulist_parser(built):
    L = built.amd_lib()
    buf = (C.c_ubyte * 64)()
    assert L.rtlws_topo_parse_cpulist(b"0-3,8,10-11", buf, 64) == 7
    assert [i for i in range(64) if buf[i]] == [0, 1, 2, 3, 8, 10, 11]
    assert L.rtlws_topo_parse_cpulist(b"5", buf, 64) == 1 and buf[5] == 1 and buf[0] == 0
    assert L.rtlws_topo_parse_cpulist(b"60-70", buf, 64) == 4          # CPUs beyond the buffer are ignored
    assert L.rtlws_topo_parse_cpulist(b"", buf, 64) == 0
    for bad in (b"3-1", b"a", b"1-", b"1;2", b"-3"):
        assert L.rtlws_topo_parse_cpulist(bad, buf, 64) == -1, bad
    # a huge range (a corrupt or caller-supplied sysfs file) is clamped, not iterated over; an
    # unrepresentable number is malformed
    assert L.rtlws_topo_parse_cpulist(b"0-9223372036854775807", buf, 64) == 64
    assert L.rtlws_topo_parse_cpulist(b"70-9223372036854775807", buf, 64) == 0
    assert L.rtlws_topo_parse_cpulist(b"0-99999999999999999999999", buf, 64) == -1
    assert L.rtlws_topo_parse_cpulist(b"99999999999999999999999", buf, 64) == -1


@pytest.mark.parametrize("ndev,nnodes", [(1, 1), (2, 1), (2, 2), (8, 1), (8, 2)])
def test_device_to_node_to_cpuset(built, tmp_path, ndev, nnodes):
    buses = EIGHT[:ndev]
    per = 64 // nnodes
    nodes = {n: "%d-%d,%d-%d" % (n * per, n * per + per - 1, 64 + n * per, 64 + n * per + per - 1) for n in range(nnodes)}
    # the first half of the devices on node 0, the second on the last node (a two-socket 8-GPU host)
    node_of = {b: (i * nnodes) // ndev for i, b in enumerate(buses)}
    root = make_sysfs(tmp_path / "sys", node_of, nodes)
    for i, b in enumerate(buses):
        t = built.topo_describe(bus_id=b.upper(), sysfs_root=root)       # the runtime may spell hex in upper case
        assert t is not None and t.bus_id.decode() == b and t.device == -1
        assert t.numa_node == node_of[b] and t.cpulist.decode() == nodes[node_of[b]] and t.ncpus == 2 * per
    # devices of different nodes get disjoint cpusets
    if nnodes == 2:
        a = built.topo_describe(bus_id=buses[0], sysfs_root=root).cpulist
        z = built.topo_describe(bus_id=buses[-1], sysfs_root=root).cpulist
        assert a != z


def test_unknown_node_and_unknown_device(built, tmp_path):
    root = make_sysfs(tmp_path / "sys", {"0000:05:00.0": None}, {})
    t = built.topo_describe(bus_id="0000:05:00.0", sysfs_root=root)
    assert t.numa_node == -1 and t.cpulist == b"0-1" and t.ncpus == 2          # local_cpulist as the fallback
    t = built.topo_describe(bus_id="0000:99:00.0", sysfs_root=root)             # not in the tree: nothing known, no error
    assert t is not None and t.numa_node == -1 and t.ncpus == 0 and t.cpulist == b""
    assert built.topo_describe(bus_id="../../etc", sysfs_root=root) is None     # not a bus id
    assert built.topo_describe(device=-1) is None                               # neither a device nor a bus id
    # without a GPU the runtime has no bus id to give: nothing known, nothing pinned, no failure
    if built.device_count() == 0:
        t = built.topo_describe(device=0)
        assert t is not None and t.bus_id == b"" and t.numa_node == -1
        assert built.amd_lib().rtlws_topo_pin_thread(C.byref(t)) == 0


def test_pin_thread_intersects_with_the_jobs_mask(built, tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two CPUs")
    keep = allowed[:max(1, len(allowed) // 2)]
    root = make_sysfs(tmp_path / "sys", {"0000:05:00.0": 0, "0000:15:00.0": 1},
                      {0: ",".join(str(c) for c in keep) + ",4000", 1: "4001-4005"})
    got = {}

    def work():
        t = built.topo_describe(bus_id="0000:05:00.0", sysfs_root=root)
        got["n"] = built.amd_lib().rtlws_topo_pin_thread(C.byref(t))
        got["mask"] = sorted(os.sched_getaffinity(0))            # (the calling thread's)
        far = built.topo_describe(bus_id="0000:15:00.0", sysfs_root=root)
        got["far"] = built.amd_lib().rtlws_topo_pin_thread(C.byref(far))   # no CPU of that node is ours: untouched
        got["mask_after_far"] = sorted(os.sched_getaffinity(0))

    th = threading.Thread(target=work)
    th.start()
    th.join()
    assert got["n"] == len(keep) and got["mask"] == keep
    assert got["far"] == 0 and got["mask_after_far"] == keep
    assert sorted(os.sched_getaffinity(0)) == allowed             # this thread was never touched


def test_drivers_print_the_plan_without_a_gpu(built, tmp_path):
    root = make_sysfs(tmp_path / "sys", {b: (0 if i < 4 else 1) for i, b in enumerate(EIGHT)}, {0: "0-63", 1: "64-127"})
    out = subprocess.run([os.path.join(built.LIB_DIR, "rtlws_multi_batch"), "--plan-only", "--devices", "8", "--frames", "65536",
                          "--bus-ids", ",".join(EIGHT), "--sysfs-root", root], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout)
    assert [s["numa_node"] for s in plan["shards"]] == [0] * 4 + [1] * 4
    assert [s["bus_id"] for s in plan["shards"]] == EIGHT
    assert {s["cpulist"] for s in plan["shards"][:4]} == {"0-63"} and {s["cpulist"] for s in plan["shards"][4:]} == {"64-127"}
    assert sum(s["frames"] for s in plan["shards"]) == 65536 and all(s["cpus"] == 64 for s in plan["shards"])
    out = subprocess.run([os.path.join(built.LIB_DIR, "rtlws_multi_stream"), "--plan-only", "--devices", "2", "--streams", "8",
                          "--bus-ids", ",".join(EIGHT[3:5]), "--sysfs-root", root], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout)
    assert plan["stream_devices"] == [0, 1] * 4
    assert [(d["device"], d["numa_node"], d["cpulist"]) for d in plan["device_topology"]] == [(0, 0, "0-63"), (1, 1, "64-127")]
