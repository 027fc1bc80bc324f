#!/bin/bash
# round 3, session cq: with nt queue stores the shade launches are shorter -- pipelines / shade rule / sample groups of the C2 default once more
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" base:- p2:-:pipes=2 p4:-:pipes=4 st0:-:stagger=0 b6:-:extend_blocks=6 2>&1 | tee $O/r03cq_ab_c2_knobs_after_nt.log
run() { echo -n "$*: "; python bench.py --no-cpu-baseline --no-extra-legs --reps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], '[%s..%s]' % (d.get('value_min'), d.get('value_max')), 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'))"; }
(for r in 1 2 3; do run --steps 16; run --steps 16 --sample-groups 4; run --steps 16 --sample-groups 16; done) 2>&1 | tee -a $O/r03cq_ab_c2_knobs_after_nt.log
