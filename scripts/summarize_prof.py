#!/usr/bin/env python3
"""Summarise a scripts/gpu_profile.sh output directory: per-kernel time (kernel-trace stats) and
per-kernel PMC sums / per-launch means.  HBM bytes follow MI355X_MICROARCH.md section HBM:
FETCH_SIZE and WRITE_SIZE are in KiB... (rocprofv3 reports them in units of 1 KiB? no: in bytes/1024
on this stack) -- we report the raw counter and bytes = raw * 1024, and for reads ALSO the gfx950
correction (x2 for wide coalesced reads, 128-B requests tallied as 64 B)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {"kernels": {}, "pmc": {}}


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].strip()


for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        out["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "total_ns": int(r["TotalDurationNs"]),
                                            "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"]),
                                            "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"])}

for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        ndisp = defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[k].add(r["Dispatch_Id"])
        for k, cs in acc.items():
            e = out["pmc"].setdefault(k, {})
            e["dispatches"] = len(ndisp[k])
            for c, v in cs.items():
                e[c] = v
                e[c + "_per_launch"] = v / max(len(ndisp[k]), 1)

for k, e in out["pmc"].items():
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
        fetch = e.get("FETCH_SIZE_per_launch", 0.0) * 1024.0
        write = e.get("WRITE_SIZE_per_launch", 0.0) * 1024.0
        e["hbm_read_bytes_per_launch_raw"] = fetch
        e["hbm_read_bytes_per_launch_gfx950_x2"] = 2.0 * fetch
        e["hbm_write_bytes_per_launch"] = write
        e["hbm_bytes_per_launch"] = 2.0 * fetch + write

json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1, sort_keys=True)
print("== kernel time")
for k, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["total_ns"]):
    print(f"{k[:60]:60s} calls {e['calls']:6d} total {e['total_ns']/1e6:10.3f} ms avg {e['avg_ns']/1e3:9.2f} us  {e['pct']:5.1f}%")
print("== pmc (per launch)")
for k, e in sorted(out["pmc"].items()):
    print(k[:70], "dispatches", e.get("dispatches"))
    for c in sorted(e):
        if c.endswith("_per_launch") and not c.startswith("hbm"):
            print(f"    {c:45s} {e[c]:18.1f}")
    for c in ("hbm_read_bytes_per_launch_raw", "hbm_read_bytes_per_launch_gfx950_x2", "hbm_write_bytes_per_launch", "hbm_bytes_per_launch"):
        if c in e:
            print(f"    {c:45s} {e[c]:18.1f}")
