#!/bin/bash
# round 4, GPU session S: counters of k_fused_inst on config C4 (what bounds it: issue, waiting, lanes)
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
bash scripts/gpu_profile.sh r04s_c4_fused --config c4 --pipeline fused --steps 8 --no-extra-legs --warmup 0 --no-cpu-baseline > $O/r04s_c4_fused_profile.log 2>&1
python scripts/make_pmc_json.py $O/prof_r04s_c4_fused $O/r04s_pmc_fused_c4.json "--config c4 --pipeline fused --steps 8 --no-extra-legs" --kernel=k_fused_inst || echo "pmc json failed"
cat $O/r04s_pmc_fused_c4.json
