#!/bin/bash
# C2: how many persistent extend blocks per CU (7 fill the register file: k_shade of the other pipeline cannot be co-resident)
O=gpurun_out; mkdir -p $O; L=$O/${1:-r02v}_blocks.txt; : > $L
run() { echo "== $*" >> $L; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-legs ${CFG} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], 'extend_ms', r['extend_ms'], 'shade_ms', r['shade_ms'], 'device_ms', d['device_ms_rank0'])
" >> $L; }
for b in 7 6 5 4 3; do run PT_TUNE_EXTEND_BLOCKS=$b; done
for b in 5 4; do run PT_TUNE_EXTEND_BLOCKS=$b PT_TUNE_PIPES=3; done
run PT_TUNE_EXTEND_BLOCKS=7
cat $L
