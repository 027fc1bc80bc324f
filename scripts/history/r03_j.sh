#!/bin/bash
# round 3, session j: top levels of the 8-wide tree staged in LDS (pt_tuning.top8_nodes) -- parity, then same-box A/B on C5 / C5x
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "bvh8 or hbm8 or full_size or big_scene or c5_full" > $O/r03j_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/r03j_pytest.txt
for r in 1 2; do
  for v in top8_nodes=0 top8_nodes=9 top8_nodes=73 top8_nodes=585 "top8_nodes=73,lds_stack=10" "top8_nodes=585,lds_stack=10"; do
    for c in c5 c5x; do
      echo -n "$c $v: "
      PT_TUNE="$v" timeout 600 python bench.py --config $c --steps 4 --warmup 1 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; g=r['gather']
print(d['value'], '[%s..%s]' % (d['value_min'], d['value_max']), 'nodes/ray', g['bvh_nodes_per_ray'], 'lanes', r['active_lanes'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'], 'avg_us', r['avg_launch_us'])"
    done
  done
done 2>&1 | tee $O/r03j_ab_top8.log
