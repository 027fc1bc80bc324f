#!/bin/bash
# round 3, session ah: variants of the shade rule: which pipeline a shade launch waits for, and with 2 / 3 / 4 pipelines
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2 3 4; do for v in "pipes=3,stagger=2" "pipes=3,stagger=5" "pipes=4,stagger=5" "pipes=4,stagger=2" "pipes=2,stagger=2" "pipes=2"; do echo -n "c2 [$v]: "; PT_TUNE=$v python bench.py --steps 16 --warmup 2 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'])"; done; done 2>&1 | tee $O/r03ah_c2_shade_rule_variants.log
