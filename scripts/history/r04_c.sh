#!/bin/bash
# round 4, GPU session C: fused pipeline -- refill sweep upward, the whole GPU suite on the new workspace code, rocprofv3 stats +
# PMC of k_fused at its best setting.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for t in "refill=40" "refill=48" "refill=56" "refill=64" "refill=44,extend_blocks=5"; do
  PT_TUNE="$t" timeout 300 python bench.py --pipeline fused --no-extra-legs --no-cpu-baseline --reps 5 > $O/r04c_fused_$(echo $t | tr '=,' '__').json 2> $O/r04c_fused.err
  python - "$t" $O/r04c_fused_$(echo $t | tr '=,' '__').json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("fused", sys.argv[1] or "default", d["value"], d["value_min"], d["value_max"], "ms/frame", d["ms_per_step"], "kernel us", d.get("roofline", {}).get("avg_launch_us"))
except Exception as e:
    print("fused", sys.argv[1], "ERR", e, open("gpurun_out/r04c_fused.err").read()[-600:])
PY
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r04c_pytest.log
export PT_TUNE="refill=40"
bash scripts/gpu_profile.sh r04c_fused --pipeline fused --steps 16 --warmup 0 --reps 1 --no-cpu-baseline --no-extra-legs > $O/r04c_prof_fused.log 2>&1
python scripts/make_pmc_json.py $O/prof_r04c_fused $O/r04c_pmc_fused_c2.json "--pipeline fused --steps 16 --no-extra-legs PT_TUNE=$PT_TUNE" --kernel=k_fused || echo "pmc json failed"
cp $O/prof_r04c_fused/summary.txt $O/r04c_fused_rocprofv3_summary.txt
find $O/prof_r04c_fused -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04c_fused_kernel_stats.csv
rm -rf $O/prof_r04c_fused
