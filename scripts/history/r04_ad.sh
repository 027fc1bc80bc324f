#!/bin/bash
# round 4, GPU session AD: the default bench line with roofline.traffic measured live (two nested rocprofv3 --pmc passes); wall time of the whole default run
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
S=$(date +%s)
timeout 1500 python bench.py > $O/r04ad_bench_default.json 2> $O/r04ad_bench_default.err; echo "rc=$? wall_s=$(( $(date +%s) - S ))" | tee $O/r04ad_wall.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04ad_bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], "traffic", r.get("traffic"), "frac_counted", r.get("frac_counted"), r.get("traffic_live"))
print("pmc_profile", r.get("pmc_profile"))
print("c4", d["roofline_c4"].get("mrays_per_s"), d["roofline_c4"].get("fused"))
PY
tail -3 $O/r04ad_bench_default.err
