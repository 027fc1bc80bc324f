#!/bin/bash
# round 3, session u: with the pipelines free-running (lagged poll), do the grid cap / pipeline count / stagger knobs read differently?
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2; do for v in "" stagger=0 extend_blocks=6 extend_blocks=5 extend_blocks=4 pipes=3 "pipes=3,extend_blocks=5" refill=24 refill=8; do echo -n "c2 [$v]: "; PT_TUNE=$v python bench.py --steps 16 --warmup 2 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'])"; done; done 2>&1 | tee $O/r03u_c2_knobs_free_running.log
