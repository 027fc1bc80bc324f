#!/bin/bash
# round 3, session au: C5 / C5x / C4 by batch shape (frames in flight x sample groups)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
one() { python bench.py $1 --warmup 1 --reps 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'rounds', d['rounds'], 'ws GB', round(d['workspace_bytes']/2**30,1), 'ms/frame', d['ms_per_step'], 'frac', r['frac'], 'rays/launch', int(r['rays_per_launch']))"; }
for r in 1 2; do
for g in 0 8 16; do echo -n "c5 K=4 G=$g: "; one "--config c5 --steps 4 --sample-groups $g"; done
for g in 0 8; do echo -n "c5 K=8 G=$g: "; one "--config c5 --steps 8 --sample-groups $g"; done
echo -n "c5 K=16 G=0: "; one "--config c5 --steps 16"
echo -n "c5 K=16 G=8: "; one "--config c5 --steps 16 --sample-groups 8"
for g in 0 8 16; do echo -n "c5x K=2 G=$g: "; one "--config c5x --steps 2 --sample-groups $g"; done
echo -n "c5x K=8 G=0: "; one "--config c5x --steps 8"
for g in 0 8; do echo -n "c4 K=8 G=$g: "; one "--config c4 --steps 8 --sample-groups $g"; done
echo -n "c4 K=16 G=0: "; one "--config c4 --steps 16"
done 2>&1 | tee $O/r03au_shapes_c5_c4.log
