#!/bin/bash
# round 3, session bx: AUTO shapes at the frame counts a driver may ask for (--steps 20 in round 2), and C5 at 8 frames
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { echo -n "$*: "; python bench.py --no-cpu-baseline --no-extra-legs --reps 5 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], '[%s..%s]' % (d.get('value_min'), d.get('value_max')), 'ms/frame', d['ms_per_step'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'pipes', c.get('pipelines'), 'rounds', d.get('rounds'), 'ws GB', round(d.get('workspace_bytes',0)/2**30,1))"; }
(for r in 1 2; do
run --steps 20 --warmup 2
run --steps 20 --warmup 2 --sample-groups 4
run --steps 20 --warmup 2 --sample-groups 16
run --steps 10 --warmup 2
run --steps 12 --warmup 2
run --steps 24 --warmup 2
run --steps 5 --warmup 1
run --steps 40 --warmup 2
run --config c5 --steps 8 --warmup 1
run --config c5 --steps 4 --warmup 1
done) 2>&1 | tee $O/r03bx_steps_a_driver_may_ask_for.log
