#!/bin/bash
# round 3, session bd: PMC of k_extend8 on C5 with the new vote / refill defaults
O=gpurun_out; mkdir -p $O; TAG=r03bd
export TMPDIR=/tmp
for c in c5; do
  extra="--config $c"; steps="--steps 4"
  bash scripts/gpu_profile.sh ${TAG}_$c $extra $steps --warmup 0 --reps 1 --no-cpu-baseline --no-extra-legs > $O/${TAG}_prof_$c.log 2>&1
  python scripts/make_pmc_json.py $O/prof_${TAG}_$c $O/${TAG}_pmc_extend_$c.json "$extra $steps --no-extra-legs" > /dev/null || echo "pmc json failed for $c"
  cp $O/prof_${TAG}_$c/summary.txt $O/${TAG}_${c}_rocprofv3_summary.txt
  rm -rf $O/prof_${TAG}_$c
done
python - <<'PY'
import json
p=json.load(open("gpurun_out/r03bd_pmc_extend_c5.json"))
print(json.dumps(p, indent=1)[:3000])
PY
