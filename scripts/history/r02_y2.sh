#!/bin/bash
# default bench (with c2_exact / latency legs) under 2 vs AUTO pipelines, twice each
O=gpurun_out; mkdir -p $O; L=$O/${1:-r02y2}_legs.txt; : > $L
run() { env "$@" timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        print('$*', d['value'], 'c2_exact', d['c2_exact']['mrays_per_s'], 'latency', d['latency_1frame']['median_ms'], 'c5', d['roofline_c5']['mrays_per_s'])
" >> $L; }
for rep in 1 2; do run PT_TUNE_PIPES=2; run X=1; done
cat $L
