#!/bin/bash
TAG=${1:-r02c}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > $O/${TAG}_pytest.log
for r in 8 16 24 32 48; do echo -n "refill $r: "; PT_TUNE_REFILL=$r python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['extend_ms'], r['shade_ms'], r.get('valu_wave_instr_per_64_rays'), r['active_lanes'])"; done > $O/${TAG}_refill.log 2>&1
bash scripts/gpu_profile.sh ${TAG}_c2 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs > $O/${TAG}_prof_c2.log 2>&1
python scripts/make_pmc_json.py $O/prof_${TAG}_c2 $O/${TAG}_pmc_c2.json "--steps 1" > /dev/null
cp $O/prof_${TAG}_c2/stats.log $O/${TAG}_stats_c2.log
rm -rf $O/prof_${TAG}_c2/pmc_*
cat $O/${TAG}_pytest.log $O/${TAG}_refill.log; cat $O/${TAG}_pmc_c2.json
