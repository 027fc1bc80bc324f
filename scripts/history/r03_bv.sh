#!/bin/bash
# round 3, session bv: AUTO with 256 M live paths for single-level LDS scenes -- C2 over K = 1 ... 32 in three processes each, C4 / C5 unchanged?, then the default command twice
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { echo -n "$*: "; python bench.py --no-cpu-baseline --no-extra-legs --reps 5 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], '[%s..%s]' % (d.get('value_min'), d.get('value_max')), 'ms/frame', d['ms_per_step'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'pipes', c.get('pipelines'), 'rounds', d.get('rounds'), 'ws GB', round(d.get('workspace_bytes',0)/2**30,1))"; }
(for r in 1 2 3; do
for k in 1 2 4 8 16 32; do run --steps $k; done
run --config c4 --steps 8
run --config c5 --steps 4
done
for r in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default:', d['value'], d['value_min'], d['value_max'], 'c2_exact', d['c2_exact']['mrays_per_s'], 'lat1', d['latency_ms_1frame'], 'c4', d['roofline_c4']['mrays_per_s'], 'c5', d['roofline_c5']['mrays_per_s'], 'c5x', d['roofline_c5x']['mrays_per_s'], 'ws GB', round(d['workspace_bytes']/2**30,1), 'exact', d.get('frame0_film_bit_exact'))"; done) 2>&1 | tee $O/r03bv_auto_256m.log
