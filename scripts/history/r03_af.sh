#!/bin/bash
# round 3, session af: experiment -- "at most two of three pipelines in one phase" by events (stagger = 2: shade, 3: traversal, 4: both)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2 3; do for v in "" stagger=2 stagger=3 stagger=4; do echo -n "c2 [$v]: "; PT_TUNE=$v python bench.py --steps 16 --warmup 2 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'], 'pipes', d['config']['pipelines'])"; done; done 2>&1 | tee $O/r03af_c2_two_of_three.log
PT_TUNE_TL=stagger=4 bash scripts/gpu_timeline.sh r03af_s4 --steps 16 --warmup 1 2>&1 | tail -4
