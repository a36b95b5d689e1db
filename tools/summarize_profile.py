#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory into files fit for profiles/.

usage: summarize_profile.py gpurun_out/prof_<tag> profiles/<name> [workload]

Writes <name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, our kernels
only, names shortened), <name>_pmc.json (per-launch means of every counter for
the dominant kernel) and updates profiles/hbm_traffic.json with the HBM bytes
per launch: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a
wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM), so the read
side is doubled -- both raw and corrected values are kept.
"""
import csv
import glob
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "batched_1024pt_64k_frames"
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)


def short(name):
    name = re.sub(r"\(rtlws::SpectraParams\)", "", name)
    return name if len(name) < 120 else name[:117] + "..."


stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
dominant = None
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    ours = [r for r in rows if "rtlws::" in r["Name"]]
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

out = {"kernel": short(dominant or ""), "dispatch": meta, "counters": pmc}
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fetch_kib = pmc["FETCH_SIZE"]["mean_per_launch"]
    write_kib = pmc["WRITE_SIZE"]["mean_per_launch"]
    hbm = {"fetch_bytes_raw": fetch_kib * 1024, "fetch_bytes_corrected_x2": 2 * fetch_kib * 1024,
           "write_bytes": write_kib * 1024,
           "bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
           "note": "FETCH_SIZE/WRITE_SIZE in KiB per launch; FETCH doubled per the gfx950 "
                   "correction in MI355X_MICROARCH.md (HBM); separate --pmc passes",
           "source": os.path.basename(dst)}
    out["hbm"] = hbm
    tpath = os.path.join(os.path.dirname(dst) or ".", "hbm_traffic.json")
    allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
    allt[workload] = hbm
    json.dump(allt, open(tpath, "w"), indent=1, sort_keys=True)
json.dump(out, open(dst + "_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
