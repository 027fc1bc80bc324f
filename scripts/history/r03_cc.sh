#!/bin/bash
# round 3, session cc: full GPU suite after AUTO's new crossover (8-wide tree from ~11 000 triangles), soup sizes around it under AUTO
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee $O/r03cc_pytest_gpu_suite.txt
run() { echo -n "$*: "; python bench.py --no-cpu-baseline --no-extra-legs --reps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['bvh'].get('extend_variant','')[:30], 'build_ms', d['bvh'].get('build_ms'))"; }
(for n in 8000 12000 30000 100000; do run --config c5 --steps 4 --soup-tris $n; done) 2>&1 | tee $O/r03cc_auto_crossover.txt
