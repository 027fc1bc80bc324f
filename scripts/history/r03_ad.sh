#!/bin/bash
# round 3, session ad: batch shape of the default bench (16 frames) under the free-running schedule: sample groups x pipelines
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
one() { PT_TUNE=$2 python bench.py $1 --warmup 1 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'pipes', c.get('pipelines'), 'rounds', d['rounds'], 'ws GB', round(d['workspace_bytes']/2**30,1))"; }
for r in 1 2; do
for g in 2 4 8; do for v in "" pipes=2 pipes=3; do echo -n "c2 K=16 G=$g [$v]: "; one "--steps 16 --sample-groups $g" "$v"; done; done
for f in 8; do for g in 4 8; do echo -n "c2 K=16 fif=$f G=$g []: "; one "--steps 16 --frames-in-flight $f --sample-groups $g" ""; done; done
done 2>&1 | tee $O/r03ad_c2_batch_shapes.log
