#!/bin/bash
# round 3, session av: AUTO aims big scenes at 128 M live paths -- full GPU suite, C5 / C5x / C4 / C2 default shapes
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/r03av_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03av_pytest.txt
one() { python bench.py $1 --warmup 1 --reps 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'fif', c.get('frames_in_flight'), 'groups', c.get('sample_groups'), 'rounds', d['rounds'], 'ws GB', round(d['workspace_bytes']/2**30,1), 'ms/frame', d['ms_per_step'], 'frac', r['frac'], 'rays/launch', int(r['rays_per_launch']))"; }
for r in 1 2; do
echo -n "c5 K=4: "; one "--config c5 --steps 4"
echo -n "c5x K=2: "; one "--config c5x --steps 2"
echo -n "c4 K=8: "; one "--config c4 --steps 8"
echo -n "c2 K=16: "; one "--steps 16"
done 2>&1 | tee $O/r03av_auto_shapes.log
