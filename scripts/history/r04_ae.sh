#!/bin/bash
# round 4, GPU session AE: the default bench line with roofline.traffic measured live (two nested rocprofv3 --pmc passes); wall time of the whole default run
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
S=$(date +%s)
timeout 1500 python bench.py > $O/r04ae_bench_default.json 2> $O/r04ae_bench_default.err; echo "rc=$? wall_s=$(( $(date +%s) - S ))" | tee $O/r04ae_wall.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04ae_bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], "traffic", r.get("traffic"), "frac_counted", r.get("frac_counted"), r.get("traffic_live"))
print("pmc_profile", r.get("pmc_profile"))
print("shade", d["roofline_shade"].get("traffic"), d["roofline_shade"].get("frac_counted"), d["roofline_shade"].get("traffic_live"))
for leg in ("c4", "c5", "c5x"):
    l = d["roofline_" + leg]
    print(leg, l.get("mrays_per_s"), "frac", l.get("frac"), "counted", l.get("frac_counted"), l.get("traffic_live"), "committed", (l.get("pmc_profile") or {}).get("hbm_bytes_per_ray"))
PY
tail -3 $O/r04ae_bench_default.err
