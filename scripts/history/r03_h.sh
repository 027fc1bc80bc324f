#!/bin/bash
# round 3, session h (re-entry after the container was replaced): full GPU suite on the restored tree, default bench line,
# rocprofv3 kernel stats of the very same command
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03h_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/r03h_pytest.txt; grep -n "Error" -B2 -A6 $O/r03h_pytest.txt | head -40
timeout 900 python bench.py > $O/r03h_bench_default.json 2> $O/r03h_bench_default.err; echo "bench rc=$?"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_r03h_default -o stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/r03h_bench_default_rocprof.json 2>$O/r03h_rocprof.err )
find $O/prof_r03h_default -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03h_default_kernel_stats.csv
rm -rf $O/prof_r03h_default
python - <<'PY'
import json
for f in ("gpurun_out/r03h_bench_default.json", "gpurun_out/r03h_bench_default_rocprof.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["value_min"], d["value_max"], d.get("cpu_baseline"))
    print(" roofline", {k: d["roofline"][k] for k in ("kernel","frac","avg_launch_us","achieved") if k in d["roofline"]})
    for k in ("roofline_c4","roofline_c5","roofline_c5x"):
        x=d.get(k) or {}; print(" ", k, x.get("error") or (x.get("mrays_per_s"), x.get("kernel"), x.get("frac"), x.get("avg_launch_us"), x.get("active_lanes")))
PY
head -12 $O/r03h_default_kernel_stats.csv
