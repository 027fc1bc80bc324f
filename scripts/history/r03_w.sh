#!/bin/bash
# round 3, session w: where do three free-running pipelines pay?  C2 by frames in flight, C4, C5
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
one() { PT_TUNE=$2 python bench.py $1 --warmup 1 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'ms/frame', d['ms_per_step'])"; }
for r in 1 2; do
for k in 4 8 12 16 32; do for v in pipes=2 "pipes=3,stagger=0"; do echo -n "c2 K=$k [$v]: "; one "--steps $k" "$v"; done; done
for k in 8 16; do for v in pipes=2 "pipes=3,stagger=0" pipes=3; do echo -n "c4 K=$k [$v]: "; one "--config c4 --steps $k" "$v"; done; done
for v in pipes=2 "pipes=3,stagger=0"; do echo -n "c5 K=4 [$v]: "; one "--config c5 --steps 4" "$v"; done
done 2>&1 | tee $O/r03w_pipes_by_shape.log
