#!/bin/bash
# dev tool (GPU box): rocprofv3 stats + PMC passes for the three bench configs, condensed PMC records,
# and plain bench lines.  usage: bash scripts/profile_all.sh <tag>
TAG=${1:-r01x}
for c in c2 c4 c5; do
  extra=""; [ $c != c2 ] && extra="--config $c"
  steps=""; [ $c = c5 ] && steps="--steps 4"; [ $c = c4 ] && steps="--steps 8"
  bash scripts/gpu_profile.sh ${TAG}_$c $extra $steps --warmup 0 --no-cpu-baseline > gpurun_out/prof_${TAG}_$c.log 2>&1
  python scripts/make_pmc_json.py gpurun_out/prof_${TAG}_$c gpurun_out/pmc_${TAG}_$c.json "$extra $steps" > /dev/null || echo "pmc json failed for $c"
done
rm -rf gpurun_out/prof_*/pmc_* gpurun_out/prof_*/stats/*/*agent_info* 2>/dev/null
du -sh gpurun_out
