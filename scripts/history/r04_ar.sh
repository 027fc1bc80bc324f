#!/bin/bash
# round 4, GPU session AR: the final tree once more: suite, smoke, default line (live counter passes), C5x line
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04ar_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/r04ar_pytest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); timeout 900 python bench.py > $O/r04ar_bench_default.json 2> /dev/null; echo "default bench wall_s=$(( $(date +%s) - S ))"
timeout 600 python bench.py --config c5x --steps 2 --no-cpu-baseline --no-extra-legs > $O/r04ar_bench_c5x.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04ar_bench_default.json").read().strip().splitlines()[-1])
print("default", d["value"], d["ms_per_step"], "fused", d["c2_fused"]["k_steps"]["mrays_per_s"], "c4", d["roofline_c4"]["mrays_per_s"], d["roofline_c4"]["fused"]["mrays_per_s"],
      "c5", d["roofline_c5"]["mrays_per_s"], d["roofline_c5"]["frac"], d["roofline_c5"]["frac_counted"], "c5x", d["roofline_c5x"]["mrays_per_s"], d["roofline_c5x"]["frac"], d["roofline_c5x"]["frac_counted"],
      "frame0", d.get("frame0_film_bit_exact"), "cpu", d["cpu_baseline"]["value"])
x = json.loads(open("gpurun_out/r04ar_bench_c5x.json").read().strip().splitlines()[-1])
print("c5x line", x["value"], x["ms_per_step"], x["roofline"]["frac"], x["roofline"].get("frac_counted"))
PY
