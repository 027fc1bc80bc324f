#!/bin/bash
# round 4, GPU session E: the builders' split through the GPU suite; VALU issue rates of the byte-decode candidates; C3 shard
# efficiency on the final code (both pipelines); the default bench line with the 512-thread fused kernel.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04e_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r04e_pytest.log
mkdir -p scripts/ubench/bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/ubench/valu_rate.hip -o scripts/ubench/bin/valu_rate && scripts/ubench/bin/valu_rate > $O/r04_valu_rate_ubench.txt 2>&1; grep -i "ubyte\|perm\|fma_mix\|bfe\|v_fma_f32 \|v_add_f32" $O/r04_valu_rate_ubench.txt
timeout 900 python scripts/probe_shard_efficiency.py 32 wavefront > $O/r04_shard_efficiency.json 2> $O/r04_shard_efficiency.err; cat $O/r04_shard_efficiency.err
timeout 900 python scripts/probe_shard_efficiency.py 32 fused > $O/r04_shard_efficiency_fused.json 2> $O/r04_shard_efficiency_fused.err; cat $O/r04_shard_efficiency_fused.err
timeout 600 python bench.py --no-cpu-baseline > $O/r04e_bench_default.json 2> $O/r04e_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04e_bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["value_min"], d["value_max"], "c4", d["roofline_c4"]["mrays_per_s"], "c5", d["roofline_c5"]["mrays_per_s"], "c5x", d["roofline_c5x"]["mrays_per_s"], "\nfused leg:", json.dumps(d.get("c2_fused")))
PY
