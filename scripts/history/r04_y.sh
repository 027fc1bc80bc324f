#!/bin/bash
# round 4, GPU session Y: where the 46 ms of a 2047-triangle scene build go (kernel trace)
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python scripts/probe_build_small.py | tee $O/r04y_build_small.log
rocprofv3 --kernel-trace --stats -f csv -d $O/prof_r04y -o b -- python scripts/probe_build_small.py > /dev/null 2>&1
python - <<'PY' | tee -a $O/r04y_build_small.log
import csv, glob
f = glob.glob("gpurun_out/prof_r04y/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(r["Name"][:50], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), "ms")
PY
rm -rf $O/prof_r04y
