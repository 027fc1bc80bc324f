#!/bin/bash
# round 3, session l: leaf_min as a pt_tuning field -- knob test, finer sweep
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "instance or c4 or inst or two_level" > $O/r03l_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03l_pytest.txt
for r in 1 2; do
  for v in leaf_min=1 leaf_min=4 leaf_min=6 leaf_min=8 leaf_min=12 "leaf_min=8,enter_min=8" "leaf_min=8,enter_min=24" "leaf_min=8,node_yield=4" "leaf_min=8,node_yield=10"; do
      echo -n "c4 $v: "
      PT_TUNE="$v" timeout 600 python bench.py --config c4 --steps 8 --warmup 1 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; g=r['gather']
print(d['value'], '[%s..%s]' % (d['value_min'], d['value_max']), 'lanes', r['active_lanes'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'], 'avg_us', r['avg_launch_us'])"
  done
done 2>&1 | tee $O/r03l_ab_c4_leaf_min_fine.log
