#!/bin/bash
# round 3, session x: three pipelines by default for C2-class shapes -- full GPU suite, default bench line x3 processes, C4/C5 sanity
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03x_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03x_pytest.txt
for i in 1 2 3 4; do python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('default', d['value'], 'min', d['value_min'], 'max', d['value_max'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'], 'frac', r['frac'], 'valu', r.get('valu_wave_instr_per_64_rays'))"; done 2>&1 | tee $O/r03x_default_4_processes.log
timeout 900 python bench.py > $O/r03x_bench_default.json 2> $O/r03x_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03x_bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["value_min"], d["value_max"], d["c2_exact"]["mrays_per_s"], d["latency_ms_1frame"], d["frame0_film_bit_exact"])
for k in ("roofline_c4","roofline_c5","roofline_c5x"):
    x=d[k]; print(k, x.get("error") or (x["mrays_per_s"], x["kernel"], x["frac"], x["avg_launch_us"]))
PY
