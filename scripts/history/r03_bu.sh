#!/bin/bash
# round 3, session bu: C2 -- one sample per slot (32 groups: 8 rounds, no regeneration) at growing numbers of frames in flight, and 16 frames as two batches of 8
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { echo -n "$*: "; python bench.py --no-cpu-baseline --no-extra-legs --reps 5 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], '[%s..%s]' % (d.get('value_min'), d.get('value_max')), 'ms/frame', d['ms_per_step'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'pipes', c.get('pipelines'), 'rounds', d.get('rounds'), 'ws GB', round(d.get('workspace_bytes',0)/2**30,1))"; }
for r in 1 2 3; do
run --steps 16 --sample-groups 32
run --steps 16 --frames-in-flight 8 --sample-groups 32
run --steps 16 --frames-in-flight 4 --sample-groups 32
run --steps 16 --sample-groups 16
run --steps 4 --sample-groups 32
run --steps 4
run --steps 16 --frames-in-flight 8 --sample-groups 16
done 2>&1 | tee $O/r03bu_c2_one_sample_per_slot.log
