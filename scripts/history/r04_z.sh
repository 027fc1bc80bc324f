#!/bin/bash
# round 4, GPU session Z: the surface-area builder with its working set in LDS: the committed trees, the tests that read trees, build time
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "sah or bvh or tree or cornell or c1 or instanced" 2>&1 | tail -5 | tee $O/r04z_pytest.log
python scripts/probe_build_small.py | tee $O/r04z_build_small.log
rocprofv3 --kernel-trace --stats -f csv -d $O/prof_r04z -o b -- python scripts/probe_build_small.py > /dev/null 2>&1
python - <<'PY' | tee -a $O/r04z_build_small.log
import csv, glob
f = glob.glob("gpurun_out/prof_r04z/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:6]:
    print(r["Name"][:50], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), "ms")
PY
rm -rf $O/prof_r04z
