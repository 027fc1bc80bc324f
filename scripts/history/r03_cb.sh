#!/bin/bash
# round 3, session cb: BVH4 against 8-wide kernel on small scenes beyond LDS (2 500 ... 12 000 triangles)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { echo -n "$*: "; python bench.py --no-cpu-baseline --no-extra-legs --reps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print(d['value'], '[%s..%s]' % (d.get('value_min'), d.get('value_max')), 'ms/frame', d['ms_per_step'], d['bvh'].get('extend_variant','')[:40], 'nodes/ray', r['gather'].get('bvh_nodes_per_ray'), 'scene MB', round(r['gather'].get('scene_device_bytes',0)/2**20,1))"; }
(for r in 1 2; do for n in 2500 5000 12000; do
  for e in hbm hbm8; do run --config c5 --steps 4 --soup-tris $n --extend $e; done
done; done) 2>&1 | tee $O/r03cb_bvh4_vs_8wide_small.log
