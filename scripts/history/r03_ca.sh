#!/bin/bash
# round 3, session ca: where does the 8-wide kernel overtake the BVH4 kernel now?  soups of 20 k ... 400 k triangles (AUTO: BVH4 up to 32 MiB of nodes + records), both kernels
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { echo -n "$*: "; python bench.py --no-cpu-baseline --no-extra-legs --reps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print(d['value'], '[%s..%s]' % (d.get('value_min'), d.get('value_max')), 'ms/frame', d['ms_per_step'], d['bvh'].get('extend_variant','')[:40], 'nodes/ray', r['gather'].get('bvh_nodes_per_ray'), 'scene MB', round(r['gather'].get('scene_device_bytes',0)/2**20,1))"; }
(for n in 20000 50000 100000 200000 400000; do
  for e in auto hbm hbm8; do run --config c5 --steps 4 --soup-tris $n --extend $e; done
done) 2>&1 | tee $O/r03ca_bvh4_vs_8wide_midsize.log
