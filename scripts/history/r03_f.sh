#!/bin/bash
# round 3, session f: the 64-B 8-wide node (PT_EXTEND_HBM8) -- parity tests, then same-box A/B against the BVH4 on C5 and C5x
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "bvh8 or hbm8 or full_size or big_scene or instanced_hits_and_1spp" > $O/r03f_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/r03f_pytest.txt
for r in 1 2 3; do
  for v in auto hbm8; do
    for c in c5 c5x; do
      echo -n "$c $v: "
      timeout 600 python bench.py --config $c --steps 4 --warmup 1 --reps 3 --extend $v --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; g=r['gather']
print(d['value'], '[%s..%s]' % (d['value_min'], d['value_max']), 'nodes/ray', g['bvh_nodes_per_ray'], 'tris/ray', g['tris_per_ray'], 'lanes', r['active_lanes'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'], 'avg_us', r['avg_launch_us'])"
    done
  done
done 2>&1 | tee $O/r03f_ab_hbm8c.log
