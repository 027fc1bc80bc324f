#!/bin/bash
# round 3, session cs: C4 batch shapes once more on the final code (AUTO: 8 frames x 4 groups)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { echo -n "$*: "; python bench.py --no-cpu-baseline --no-extra-legs --reps 5 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], '[%s..%s]' % (d.get('value_min'), d.get('value_max')), 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'pipes', c.get('pipelines'))"; }
(for r in 1 2 3; do run --config c4 --steps 8; run --config c4 --steps 8 --sample-groups 8; run --config c4 --steps 8 --sample-groups 16; run --config c4 --steps 8 --sample-groups 2; done
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" base:- p3:-:pipes=3,stagger=2 p3f:-:pipes=3,stagger=0) 2>&1 | tee $O/r03cs_c4_shapes_final.log
