#!/bin/bash
# round 3, session ap: one sample group (no term logs: a colour accumulator per slot) under the three-pipeline schedule, by frames in flight
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
one() { python bench.py $1 --warmup 1 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'pipes', c.get('pipelines'), 'rounds', d['rounds'], 'ws GB', round(d['workspace_bytes']/2**30,1), 'ms/frame', d['ms_per_step'])"; }
for r in 1 2 3; do
for k in 16 32; do for g in 0 1 2; do echo -n "c2 K=$k G=$g: "; one "--steps $k --sample-groups $g"; done; done
for k in 8 4 2 1; do for g in 0 1; do echo -n "c2 K=$k G=$g: "; one "--steps $k --sample-groups $g"; done; done
done 2>&1 | tee $O/r03ap_c2_one_sample_group.log
