#!/bin/bash
# round 3, session ai: the shade rule as the default of three pipelines -- full GPU suite, default command x6 processes, timeline
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/r03ai_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03ai_pytest.txt
for i in 1 2 3 4 5 6; do python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('default', d['value'], 'min', d['value_min'], 'max', d['value_max'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'], 'frac', r['frac'], r['frac_all_launches_over_device_time'])"; done 2>&1 | tee $O/r03ai_default_6_processes.log
bash scripts/gpu_timeline.sh r03ai_c2 --steps 16 --warmup 1 2>&1 | tail -6
