#!/bin/bash
# round 3, session v: 2 vs 3 vs 4 free-running pipelines on C2 (K = 16, K = 2, K = 1), interleaved
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2 3 4; do for v in pipes=2 pipes=3 pipes=4 "pipes=3,stagger=0"; do echo -n "c2 K=16 [$v]: "; PT_TUNE=$v python bench.py --steps 16 --warmup 2 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'])"; done; done 2>&1 | tee $O/r03v_c2_pipes.log
for r in 1 2; do for v in pipes=2 pipes=3; do for k in 2 1; do echo -n "c2 K=$k [$v]: "; PT_TUNE=$v python bench.py --steps $k --warmup 2 --reps 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'ms/frame', d['ms_per_step'])"; done; done; done 2>&1 | tee -a $O/r03v_c2_pipes.log
