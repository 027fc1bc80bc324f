#!/bin/bash
# round 3, session br: BASELINE config C2 exactly (K = 2 frames = 64 spp) and K = 1 under other sample-group counts than AUTO's
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2; do
for a in "--steps 2" "--steps 2 --sample-groups 16" "--steps 2 --sample-groups 32" "--steps 2 --sample-groups 4" "--steps 1" "--steps 1 --sample-groups 32" "--steps 1 --sample-groups 8" "--steps 4" "--steps 4 --sample-groups 8" "--steps 4 --sample-groups 16"; do
  echo -n "$a: "; python bench.py --no-cpu-baseline --no-extra-legs --reps 5 --warmup 1 $a 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], '[%s..%s]' % (d.get('value_min'), d.get('value_max')), 'ms/frame', d['ms_per_step'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'pipes', c.get('pipelines'), 'rounds', d.get('rounds'))"
done; done 2>&1 | tee $O/r03br_c2_small_k_shapes.log
