#!/usr/bin/env python3
"""dev tool: join scripts/ubench/fetch_calib's own byte counts with the rocprofv3 --pmc CSVs of the same runs.
usage: calib_fetch.py <dir with calib_fetch/ calib_write/ [calib_*]/ and calib_stdout.txt> <out json>
For every kernel name: counter value per launch (rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB) x 1024 / the bytes the
kernel asked for = the factor a reader has to DIVIDE the counter by to get requested bytes (0.5 = the guide's x2 case)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
known = {}
for line in open(os.path.join(root, "calib_stdout.txt")):
    if not line.startswith("CALIB kernel="):
        continue
    kv = dict(p.split("=", 1) for p in line.split()[1:])
    known[kv["kernel"].replace(" ", "")] = {"launches": int(kv["launches"]), "bytes_per_launch": int(kv["bytes_per_launch"]),
                                            "records_per_launch": int(kv["records_per_launch"]), "ms": float(kv["ms"])}


def short(name):
    n = name.replace("void ", "").split("(")[0].strip().replace(" ", "")
    return n


counters = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "calib_*", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[(short(r["Kernel_Name"]), r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per.items():
        for c, v in cs.items():
            counters[k][c].append(v)

rec = {"source": "scripts/ubench/fetch_calib.hip under rocprofv3 --pmc <counter> --kernel-trace, one run per counter",
       "unit_note": "FETCH_SIZE / WRITE_SIZE are reported in KiB; ratio = counter x 1024 / bytes the kernel requested",
       "kernels": {}}
for k, kn in known.items():
    e = dict(kn)
    e["requested_GBps"] = round(kn["bytes_per_launch"] / (kn["ms"] * 1e-3) / 1e9, 1)
    if kn["records_per_launch"]:
        e["record_bytes_64"] = kn["records_per_launch"] * 64
        e["line_bytes_128"] = kn["records_per_launch"] * 128
        e["G_records_per_s"] = round(kn["records_per_launch"] / (kn["ms"] * 1e-3) / 1e9, 2)
    for c, vals in counters.get(k, {}).items():
        v = sorted(vals)[len(vals) // 2]          # median launch
        e[c + "_per_launch"] = v
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            e[c + "_bytes"] = v * 1024.0
            e[c + "_over_requested"] = round(v * 1024.0 / kn["bytes_per_launch"], 4)
            if kn["records_per_launch"]:
                e[c + "_over_record_bytes_64"] = round(v * 1024.0 / (kn["records_per_launch"] * 64), 4)
                e[c + "_over_line_bytes_128"] = round(v * 1024.0 / (kn["records_per_launch"] * 128), 4)
    rec["kernels"][k] = e
json.dump(rec, open(out, "w"), indent=1)
for k, e in rec["kernels"].items():
    print(k, {x: e[x] for x in e if "over" in x or x.endswith("GBps") or x.startswith("G_")})
