#!/bin/bash
# round 3, session aj: three pipelines + shade rule on the shapes that keep two: C4, C5, C2 at K = 2 and K = 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
one() { PT_TUNE=$2 python bench.py $1 --warmup 1 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'pipes', c.get('pipelines'), 'ms/frame', d['ms_per_step'])"; }
for r in 1 2 3; do
for v in "" "pipes=3,stagger=2"; do echo -n "c4 K=8 [$v]: "; one "--config c4 --steps 8" "$v"; done
for v in "" "pipes=3,stagger=2"; do echo -n "c4 K=16 [$v]: "; one "--config c4 --steps 16" "$v"; done
for v in "" "pipes=3,stagger=2"; do echo -n "c5 K=4 [$v]: "; one "--config c5 --steps 4" "$v"; done
for v in "" "pipes=3,stagger=2"; do echo -n "c2 K=2 [$v]: "; one "--steps 2" "$v"; done
for v in "" "pipes=3,stagger=2"; do echo -n "c2 K=1 [$v]: "; one "--steps 1" "$v"; done
for v in "" "pipes=3,stagger=2"; do echo -n "c2 K=4 [$v]: "; one "--steps 4" "$v"; done
for v in "" "pipes=2"; do echo -n "c2 K=8 [$v]: "; one "--steps 8" "$v"; done
done 2>&1 | tee $O/r03aj_shade_rule_other_shapes.log
