#!/bin/bash
# round 4, GPU session AL: what one blocking fused frame is made of (kernel trace), by sample-group count
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for G in 0 4 8 16; do python scripts/probe_fused_k1.py $G; done | tee $O/r04al_fused_k1.log
rocprofv3 --kernel-trace --stats -f csv -d $O/prof_r04al -o b -- python scripts/probe_fused_k1.py 0 > /dev/null 2>&1
python - <<'PY' | tee -a $O/r04al_fused_k1.log
import csv, glob
f = glob.glob("gpurun_out/prof_r04al/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:6]:
    print(r["Name"][:60], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1))
PY
rm -rf $O/prof_r04al
