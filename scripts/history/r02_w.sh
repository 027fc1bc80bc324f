#!/bin/bash
# C2: interleaved repetitions of (extend blocks per CU, pipelines) -- run-to-run noise on one box is ~5 %
O=gpurun_out; mkdir -p $O; L=$O/${1:-r02w}_blocks_pipes.txt; : > $L
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-legs ${CFG} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$*', d['value'], 'extend_ms', r['extend_ms'], 'shade_ms', r['shade_ms'], 'device_ms', d['device_ms_rank0'])
" >> $L; }
for rep in 1 2 3 4; do
  run PT_TUNE_EXTEND_BLOCKS=7 PT_TUNE_PIPES=2
  run PT_TUNE_EXTEND_BLOCKS=5 PT_TUNE_PIPES=2
  run PT_TUNE_EXTEND_BLOCKS=7 PT_TUNE_PIPES=3
  run PT_TUNE_EXTEND_BLOCKS=6 PT_TUNE_PIPES=3
  run PT_TUNE_EXTEND_BLOCKS=5 PT_TUNE_PIPES=3
done
sort $L | awk '{k=$1" "$2; v[k]=v[k]" "$3} END{for(k in v) print k, v[k]}' | sort
