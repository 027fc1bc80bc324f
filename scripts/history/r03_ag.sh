#!/bin/bash
# round 3, session ag: distribution over processes: free-running three pipelines vs the shade rule (2) vs both rules (4)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2 3 4 5 6 7 8; do for v in "" stagger=2 stagger=4; do echo -n "c2 [$v]: "; PT_TUNE=$v python bench.py --steps 16 --warmup 2 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'])"; done; done 2>&1 | tee $O/r03ag_c2_rules_distribution.log
