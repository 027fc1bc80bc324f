#!/bin/bash
# round 3, session cr: the final code once more -- fuzzers + determinism soak, and the C2 headline over ten processes of one box
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
(for s in "fuzz_render.py 300 17100" "fuzz_trace.py 150 95000" "fuzz_instances.py 20 16100"; do timeout 1500 python scripts/$s 2>&1 | tail -2; done; timeout 900 python scripts/soak_determinism.py 2>&1 | tail -6) | tee $O/r03cr_fuzz_soak_final.txt
for r in 1 2 3 4 5 6 7 8 9 10; do python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['value_min'], d['value_max'], d['ms_per_step'])"; done 2>&1 | tee $O/r03cr_c2_ten_processes.log
