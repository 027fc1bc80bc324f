#!/bin/bash
# round 3, session k: two-level kernel, triangle work waits for LEAF_MIN lanes (experiment)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
PT_TUNE=leaf_min=16 timeout 900 python -m pytest tests -m gpu -x -q -k "instance or c4 or inst" > $O/r03k_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03k_pytest.txt
for r in 1 2; do
  for v in 1 8 16 24 32; do
      echo -n "c4 leaf_min=$v: "
      PT_TUNE=leaf_min=$v timeout 600 python bench.py --config c4 --steps 8 --warmup 1 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; g=r['gather']
print(d['value'], '[%s..%s]' % (d['value_min'], d['value_max']), 'lanes', r['active_lanes'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'], 'avg_us', r['avg_launch_us'])"
  done
done 2>&1 | tee $O/r03k_ab_c4_leaf_min.log
