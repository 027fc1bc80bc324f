#!/bin/bash
# round 4, GPU session AZ: the default line after the counter-CSV refactor (live passes still measured)
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py > $O/r04az_bench_default.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04az_bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["traffic_live"].get("hbm_bytes_per_ray"), d["roofline_shade"]["traffic_live"].get("hbm_bytes_per_ray"),
      [d["roofline_" + l]["traffic_live"].get("hbm_bytes_per_ray") for l in ("c4", "c5", "c5x")], d["c2_fused"]["k_steps"]["mrays_per_s"], d["c2_fused"]["latency_1frame"]["median_ms"])
PY
