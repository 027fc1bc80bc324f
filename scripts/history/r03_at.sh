#!/bin/bash
# round 3, session at: K = 32 with one sample group vs AUTO, process to process; C4 / C5 at more frames
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
one() { python bench.py $1 --warmup 1 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'pipes', c.get('pipelines'), 'rounds', d['rounds'], 'ws GB', round(d['workspace_bytes']/2**30,1), 'ms/frame', d['ms_per_step'])"; }
for r in 1 2 3 4; do
echo -n "c2 K=32 G=1: "; one "--steps 32 --sample-groups 1"
echo -n "c2 K=32 auto: "; one "--steps 32"
echo -n "c2 K=64 G=1: "; one "--steps 64 --sample-groups 1"
echo -n "c2 K=16 auto: "; one "--steps 16"
echo -n "c2 K=24 G=1: "; one "--steps 24 --sample-groups 1"
done 2>&1 | tee $O/r03at_c2_k32_g1.log
for r in 1 2; do
echo -n "c4 K=8 auto: "; one "--config c4 --steps 8"
echo -n "c4 K=32 G=1: "; one "--config c4 --steps 32 --sample-groups 1"
echo -n "c4 K=32 auto: "; one "--config c4 --steps 32"
echo -n "c5 K=4 auto: "; one "--config c5 --steps 4"
echo -n "c5 K=16 G=1: "; one "--config c5 --steps 16 --sample-groups 1"
echo -n "c5 K=16 auto: "; one "--config c5 --steps 16"
done 2>&1 | tee -a $O/r03at_c2_k32_g1.log
