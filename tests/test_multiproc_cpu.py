"""N > 1 plumbing of bench.py on CPU: two processes, gloo, 127.0.0.1.
The data path has no collective (independent frames per GPU); what must be
right is the rendezvous, the barrier-bracketed timing, the MAX over ranks and
the whole-job rate.  The GPU step itself is replaced by a sleep here."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch
    import bench
    dist, world, rank = bench.init_distributed(torch, backend="gloo")
    assert world == 2 and dist is not None
    steps, frames = 5, 1000
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        time.sleep(0.02 * (rank + 1))          # rank 1 is the slow one
    dist.barrier()
    elapsed = time.perf_counter() - t0
    mine = elapsed
    own = [0.011, 0.033][rank]                 # what a rank would time around its own launches
    elapsed, other = bench.max_over_ranks(torch, dist, [elapsed, float(rank)])
    per_rank = bench.gather_ranks(torch, dist, own)
    if rank == 0:
        print(json.dumps({"value": bench.whole_job_rate(world, steps, frames, elapsed),
                          "elapsed": elapsed, "max_rank": other, "world": world, "per_rank": per_rank}))
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_timing_and_aggregation(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["world"] == 2 and res["max_rank"] == 1.0
    assert res["per_rank"] == [0.011, 0.033]        # every rank's own figure, in rank order (bench line: per_rank)
    # both ranks are bracketed by barriers, so the job lasts as long as the slow rank
    assert 0.19 < res["elapsed"] < 1.0
    assert abs(res["value"] - 2 * 5 * 1000 / res["elapsed"]) < 1e-6


def test_single_process_identity():
    sys.path.insert(0, ROOT)
    import bench
    import torch
    os.environ.pop("WORLD_SIZE", None)
    dist, world, rank = bench.init_distributed(torch, backend="gloo")
    assert dist is None and world == 1 and rank == 0
    assert bench.max_over_ranks(torch, None, [1.5, 2.5]) == [1.5, 2.5]
    assert bench.whole_job_rate(8, 10, 65536, 2.0) == 8 * 10 * 65536 / 2.0
    assert bench.algorithmic_bytes_per_frame(1024, 1, 0) == 6144
    assert bench.algorithmic_bytes_per_frame(4096, 8, 0) == 10240
    assert bench.algorithmic_bytes_per_frame(2048, 1, 8) == 40960


def test_bench_gpus_n_fans_out_by_itself():
    """`python bench.py --gpus 2` (no torch.distributed.run around it, no WORLD_SIZE):
    the file starts its own two ranks as a child process and relays rank 0's line.
    --plumbing-cpu swaps the GPU step for a sleep and RCCL for gloo, nothing else."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5",
                          "--plumbing-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout                      # the contract: ONE line on stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["asked_gpus"] == 2 and res["slowest_rank"] == 1.0
    assert abs(res["value"] - 2 * 5 * 65536 / (res["ms_per_step"] * 5e-3)) / res["value"] < 1e-9
    # every rank's own time is in the line: rank 1 sleeps twice as long as rank 0, and the job's
    # ms_per_step (MAX over ranks, closing barrier included) is at least the slow rank's own
    own = res["per_rank"]["ms_per_step_own"]
    assert len(own["all"]) == 2 and own["min"] == own["all"][0] < own["all"][1] == own["max"] <= res["ms_per_step"] * 1.001
    assert 1.5 < own["all"][1] / own["all"][0] < 2.6


def test_two_ranks_record_where_they_ran(tmp_path):
    """every rank contributes device, PCI bus id, NUMA node, CPUs pinned and its own times to
    per_rank.ranks, and rank 0 refuses an N-GPU line over ranks that share a device.  Two gloo ranks on a fake
    sysfs tree (include/rtlws_topo.h takes the root as a parameter): two devices on two nodes whose CPUs are this
    job's own, so that the pinning really happens.  Unmeasured on hardware -- this is the code path, on CPU."""
    from test_topo_cpu import make_sysfs
    cpus = sorted(os.sched_getaffinity(0))
    lo, hi = cpus[:max(1, len(cpus) // 2)], cpus[max(1, len(cpus) // 2):] or cpus[:1]
    as_list = lambda c: ",".join(str(x) for x in c)
    buses = ["0000:05:00.0", "0000:85:00.0"]
    root = make_sysfs(tmp_path / "sys", {buses[0]: 0, buses[1]: 1}, {0: as_list(lo), 1: as_list(hi)})
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RTLWS_BENCH_BUS_IDS=",".join(buses), RTLWS_BENCH_SYSFS_ROOT=root)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--plumbing-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    ranks = res["per_rank"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and [r["local_rank"] for r in ranks] == [0, 1]
    assert [r["bus_id"] for r in ranks] == buses and [r["numa_node"] for r in ranks] == [0, 1]
    assert [r["cpus_pinned"] for r in ranks] == [len(lo), len(hi)]
    assert all(r["host"] and r["ms_per_step_own"] > 0 for r in ranks)
    assert res["per_rank"]["device_clashes"] == []
    # ... and two ranks on ONE device are refused: rank 0 exits 6 (torch.distributed.run turns any failing rank
    # into its own non-zero code), the clash named in the line
    env["RTLWS_BENCH_BUS_IDS"] = buses[0]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--plumbing-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["per_rank"]["device_clashes"] == [{"ranks": [0, 1], "host": ranks[0]["host"], "device": buses[0]}]


def test_distinct_device_check_is_pure():
    sys.path.insert(0, ROOT)
    import bench
    rec = lambda r, host, bus, dev: {"rank": r, "host": host, "bus_id": bus, "device": dev}
    eight = [rec(r, "n0", "0000:%02x:00.0" % (5 + 16 * r), r) for r in range(8)]
    assert bench.check_distinct_devices(eight) == []
    assert bench.check_distinct_devices(eight + [rec(8, "n1", eight[0]["bus_id"], 0)]) == []      # another host
    bad = bench.check_distinct_devices(eight + [rec(8, "n0", eight[3]["bus_id"], 3)])
    assert bad == [{"ranks": [3, 8], "host": "n0", "device": eight[3]["bus_id"]}]
    assert bench.check_distinct_devices([rec(0, "n0", "", 0), rec(1, "n0", "", 0)]) != []          # no bus id: by index
    assert bench.check_distinct_devices([rec(0, "n0", "", 0), rec(1, "n0", "", 0)], allow_shared=True) == []
    # a launcher that shows every rank ONE device: all say "device 0", their visibility masks tell them apart
    masked = [dict(rec(r, "n0", "", 0), visible_devices=str(r)) for r in range(8)]
    assert bench.check_distinct_devices(masked) == []
    assert bench.check_distinct_devices(masked + [dict(rec(8, "n0", "", 0), visible_devices="3")]) != []


def test_bench_gpus_n_without_the_devices_fails_loudly():
    """More GPUs asked for than the host has: non-zero exit and no result line --
    never a silent n_gpus=1 under the name of an N-GPU run."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this host has the devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stdout.strip() == ""
    assert "--gpus 2" in out.stderr


def test_fan_out_decision():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.needs_fan_out(2, {}) and bench.needs_fan_out(8, {"HOME": "/"})
    assert not bench.needs_fan_out(1, {})
    assert not bench.needs_fan_out(2, {"WORLD_SIZE": "2", "RANK": "0"})     # already a rank


def test_rank_and_stream_placement_for_1_2_8_devices(monkeypatch):
    """VERDICT r2 "next" #5: the rank -> device and stream -> device rules as functions, checked
    for device counts this container cannot have.  bench.py's rule and the C driver's
    (rtlws_stream_device_for, include/rtlws_stream.h; rtlws_multi_stream --plan-only) must agree."""
    import ctypes
    import json
    import subprocess
    import bench
    import rtlws
    L = rtlws.amd_lib()
    L.rtlws_stream_device_for.argtypes = [ctypes.c_int, ctypes.c_int]
    L.rtlws_stream_device_for.restype = ctypes.c_int
    exe = os.path.join(rtlws.LIB_DIR, "rtlws_multi_stream")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    for n in (1, 2, 8):
        plan = bench.multi_stream_plan(8, n)
        assert plan == [i % n for i in range(8)]
        assert [L.rtlws_stream_device_for(i, n) for i in range(8)] == plan
        out = subprocess.run([exe, "--plan-only", "--devices", str(n)], capture_output=True, text=True, timeout=30)
        assert out.returncode == 0 and json.loads(out.stdout)["stream_devices"] == plan
        assert [bench.device_for_rank(r, n) for r in range(n)] == list(range(n))      # one rank, one device
        with pytest.raises(SystemExit):
            bench.device_for_rank(n, n)                                              # no silent sharing
    # ... except where a launcher shows every rank one device of its own through a visibility mask
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5")
    assert bench.device_for_rank(5, 1) == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    with pytest.raises(SystemExit):
        bench.device_for_rank(5, 1)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert bench.multi_stream_plan(8, 8) == list(range(8))                           # configs[4]: GPU g <- stream g
    assert L.rtlws_stream_device_for(3, 0) == -1 and L.rtlws_stream_device_for(-1, 8) == -1
    with pytest.raises(SystemExit):
        bench.device_for_rank(0, 0)
