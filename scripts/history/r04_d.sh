#!/bin/bash
# round 4, GPU session D: the split build (wavefront.hip -> six units) through the whole GPU suite and the default bench line;
# fused kernel: 256 threads x 5 waves against 512 threads x 6 waves (interleaved A/B).
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04d_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r04d_pytest.log
timeout 600 python bench.py > $O/r04d_bench_default.json 2> $O/r04d_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04d_bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["value_min"], d["value_max"], "frac", d["roofline"]["frac"], "shade frac", d["roofline_shade"]["frac"], d["roofline_shade"].get("frac_counted"), "c2_exact", d["c2_exact"]["mrays_per_s"], "lat", d["latency_ms_1frame"],
      "c4", d["roofline_c4"]["mrays_per_s"], "c5", d["roofline_c5"]["mrays_per_s"], d["roofline_c5"]["frac"], d["roofline_c5"].get("frac_counted"), "c5x", d["roofline_c5x"]["mrays_per_s"], d["roofline_c5x"].get("frac_counted"),
      "\nfused leg:", json.dumps(d.get("c2_fused")), "\ncpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread_mrays"], d["cpu_baseline"]["scaling_efficiency"])
PY
AB_ROUNDS=3 bash scripts/ab_many.sh "--pipeline fused --reps 3" build/fused_tb256.so.bin build/fused_tb512_w6.so.bin 2>&1 | tee $O/r04d_ab_fused_tb.log
