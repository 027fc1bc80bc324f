#!/bin/bash
# round 3, session ci: the C2 headline (default shape, --steps 16) over ten processes of one box on the final code -- the distribution behind one driver sample
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2 3 4 5 6 7 8 9 10; do python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['value_min'], d['value_max'], d['ms_per_step'], 'ws GB', round(d['workspace_bytes']/2**30,1))"; done 2>&1 | tee $O/r03ci_c2_ten_processes.log
