#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory into files fit for profiles/.

usage: summarize_profile.py gpurun_out/prof_<tag> <dst prefix> [workload]
       summarize_profile.py --merge-traffic profiles/hbm_traffic.json <x_pmc.json> [...]

Writes
  <dst>_kernel_stats.csv   rocprofv3 --kernel-trace --stats, our kernels only
  <dst>_timed_launches.json  the dominant kernel's launches from the per-dispatch
                           trace, in start order, WITHOUT the first 500 (the settle /
                           warm-up launches bench.py always runs): their average is
                           what must agree with the bench line's avg_launch_us and
                           stay under its ms_per_step; the bench line of the same
                           (profiled) run is copied in beside it
  <dst>_pmc.json           per-launch means of every counter for the dominant kernel,
                           HBM bytes per launch: FETCH_SIZE and WRITE_SIZE are in KiB;
                           on gfx950 FETCH_SIZE counts a wide coalesced read at half
                           its bytes (MI355X_MICROARCH.md, HBM), so the read side is
                           doubled -- both raw and corrected values are kept.
"""
import csv
import glob
import json
import os
import re
import sys

SETTLE = 500


def short(name):
    name = re.sub(r"\(rtlws::SpectraParams\)", "", name)
    return name if len(name) < 120 else name[:117] + "..."


def merge_traffic(table_path, pmc_files):
    allt = json.load(open(table_path)) if os.path.exists(table_path) else {}
    for f in pmc_files:
        d = json.load(open(f))
        if "hbm" in d and d.get("workload"):
            allt[d["workload"]] = d["hbm"]
    json.dump(allt, open(table_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(allt, indent=1, sort_keys=True))


def main():
    if sys.argv[1] == "--merge-traffic":
        return merge_traffic(sys.argv[2], sys.argv[3:])
    src, dst = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "batched_1024pt_64k_frames"
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)

    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    dominant = None
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        ours = [r for r in rows if "rtlws::" in r["Name"] and "clock_probe_kernel" not in r["Name"]]
        ours.sort(key=lambda r: -float(r["TotalDurationNs"]))
        with open(dst + "_kernel_stats.csv", "w") as f:
            f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,StdDev\n")
            for r in ours:
                f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                                                      r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]))
            other = sum(float(r["TotalDurationNs"]) for r in rows if "rtlws::" not in r["Name"])
            f.write('"(all non-rtlws kernels: torch input generation etc.)",,%d,,,,\n' % other)
        if ours:
            dominant = ours[0]["Name"]

    # per-dispatch trace of the same run: average over the TIMED launches only
    traces = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
    if traces and dominant:
        d = []
        for r in csv.DictReader(open(traces[0])):
            if r["Kernel_Name"] == dominant:
                d.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        d.sort()
        timed = d[SETTLE:]
        out = {"kernel": short(dominant), "launches_total": len(d), "dropped_first": min(SETTLE, len(d)),
               "timed_launches": len(timed), "workload": workload}
        if timed:
            durs = [e - s for s, e in timed]
            out["timed_avg_ns"] = sum(durs) / len(durs)
            out["timed_min_ns"] = min(durs)
            out["timed_max_ns"] = max(durs)
            # back-to-back launches: span / count includes the dispatch gap between kernels
            out["timed_span_per_launch_ns"] = (timed[-1][1] - timed[0][0]) / len(timed)
            all_durs = [e - s for s, e in d]
            out["all_launches_avg_ns"] = sum(all_durs) / len(all_durs)
        bj = os.path.join(src, "bench_trace.json")
        if os.path.exists(bj):
            try:
                line = [l for l in open(bj).read().splitlines() if l.startswith("{")][-1]
                b = json.loads(line)
                out["bench_line_same_run"] = {"value": b["value"], "ms_per_step": b["ms_per_step"],
                                              "steps": b["steps"], "warmup": b["warmup"],
                                              "settle_launches": b["settle_launches"],
                                              "roofline": b["roofline"]}
                if timed:
                    bytes_pl = b["roofline"]["algorithmic_bytes_per_launch"]
                    out["achieved_GBs_from_trace"] = bytes_pl / out["timed_avg_ns"]
                    out["trace_over_bench_events"] = out["timed_avg_ns"] / (1e3 * b["roofline"]["avg_launch_us"])
                    out["trace_avg_le_ms_per_step"] = out["timed_avg_ns"] <= 1e6 * b["ms_per_step"]
            except Exception as ex:        # keep the rest of the summary
                out["bench_line_same_run"] = "unreadable: %s" % ex
        json.dump(out, open(dst + "_timed_launches.json", "w"), indent=1, sort_keys=True)
        print(json.dumps(out, indent=1, sort_keys=True))

    pmc = {}
    meta = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            acc = {}
            for r in csv.DictReader(open(f)):
                if dominant and r["Kernel_Name"] != dominant:
                    continue
                if not dominant and "rtlws::" not in r["Kernel_Name"]:
                    continue
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                meta = {"VGPR_Count": r["VGPR_Count"], "SGPR_Count": r["SGPR_Count"],
                        "LDS_Block_Size": r["LDS_Block_Size"], "Scratch_Size": r["Scratch_Size"],
                        "Workgroup_Size": r["Workgroup_Size"], "Grid_Size": r["Grid_Size"]}
            for k, v in acc.items():
                pmc[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}

    out = {"kernel": short(dominant or ""), "workload": workload, "dispatch": meta, "counters": pmc}
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        fetch_kib = pmc["FETCH_SIZE"]["mean_per_launch"]
        write_kib = pmc["WRITE_SIZE"]["mean_per_launch"]
        out["hbm"] = {"fetch_bytes_raw": fetch_kib * 1024, "fetch_bytes_corrected_x2": 2 * fetch_kib * 1024,
                      "write_bytes": write_kib * 1024,
                      "bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
                      "note": "FETCH_SIZE/WRITE_SIZE in KiB per launch; FETCH doubled per the gfx950 "
                              "correction in MI355X_MICROARCH.md (HBM); separate --pmc passes",
                      "source": os.path.basename(dst)}
    json.dump(out, open(dst + "_pmc.json", "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
