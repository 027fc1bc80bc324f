#!/bin/bash
# round 3, session bt: C2 -- 128 M against 256 M (and more) live paths, three processes each, interleaved; AUTO at K = 1, 2, 16 beside them
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { echo -n "$*: "; python bench.py --no-cpu-baseline --no-extra-legs --reps 5 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], '[%s..%s]' % (d.get('value_min'), d.get('value_max')), 'ms/frame', d['ms_per_step'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'pipes', c.get('pipelines'), 'rounds', d.get('rounds'), 'ws GB', round(d.get('workspace_bytes',0)/2**30,1))"; }
for r in 1 2 3; do
run --steps 16
run --steps 16 --sample-groups 8
run --steps 16 --sample-groups 16
run --steps 8
run --steps 8 --sample-groups 16
run --steps 8 --sample-groups 32
run --steps 2
run --steps 1
run --steps 32 --sample-groups 4
run --config c4 --steps 8
done 2>&1 | tee $O/r03bt_c2_128m_vs_256m.log
