#!/bin/bash
# round 3, session g: AUTO = 8-wide tree beyond L2, NEE x instances, instanced independent pin -- full GPU suite, stadium check, default bench
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03g_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/r03g_pytest.txt; grep -n "Error" -B2 -A6 $O/r03g_pytest.txt | head -40
for t in hbm8=0 hbm8=-1; do echo "stadium $t:"; PT_TUNE=$t timeout 600 python scripts/probe_stress_scene.py 2>&1 | grep stadium | cut -c1-420; done | tee $O/r03g_stadium_hbm8.txt
timeout 900 python bench.py > $O/r03g_bench_default.json 2> $O/r03g_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03g_bench_default.json"))
print(d["value"], d["value_min"], d["value_max"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["scaling_efficiency"])
for k in ("roofline_c4","roofline_c5","roofline_c5x"):
    x=d[k]; print(k, x.get("error") or (x["mrays_per_s"], x["kernel"], x["frac"], x["active_lanes"], x["gather"]["bvh_nodes_per_ray"]))
PY
