#!/bin/bash
# round 4, GPU session AQ: k_extend8's 7-wave instantiation without the pending-group bound (template parameter): parity, C5x and C5 lines
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bvh8 or soup or hbm8 or 8wide or full_size or big or sort" 2>&1 | grep -E "passed|failed" | tee $O/r04aq_pytest.log
PT_TUNE=extend_blocks=7 timeout 600 python scripts/fuzz_trace.py 60 8900 2>&1 | tail -1 | tee -a $O/r04aq_pytest.log
for c in c5x c5; do
  timeout 900 python bench.py --config $c --steps $( [ $c = c5x ] && echo 2 || echo 4 ) --no-cpu-baseline --no-extra-legs --no-live-pmc 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['value_min'], d['value_max'], d['ms_per_step'])" | tee -a $O/r04aq_bench.log
done
