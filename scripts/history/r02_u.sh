#!/bin/bash
# C2: one pipeline (kernels run alone, back to back) vs the default two, per-kernel totals
O=gpurun_out; mkdir -p $O; L=$O/${1:-r02u}_pipes.txt; : > $L
run() { echo "== $*" >> $L; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], 'extend_ms', r['extend_ms'], 'shade_ms', r['shade_ms'], 'device_ms', d['device_ms_rank0'], 'rays/launch', r['rays_per_launch'], 'launches', r['launches'])
" >> $L; }
run PT_TUNE_PIPES=2
run PT_TUNE_PIPES=1
run PT_TUNE_PIPES=3
run PT_TUNE_PIPES=4
cat $L
