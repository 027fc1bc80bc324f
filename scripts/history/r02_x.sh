#!/bin/bash
# two vs three pipelines on C4 / C5 / C5x, interleaved repetitions
O=gpurun_out; mkdir -p $O; L=$O/${1:-r02x}_pipes_cfg.txt; : > $L
run() { cfg=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --config $cfg --steps $( [ $cfg = c4 ] && echo 8 || echo 4 ) 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$cfg $*', d['value'], 'extend_ms', r['extend_ms'], 'shade_ms', r['shade_ms'], 'device_ms', d['device_ms_rank0'])
" >> $L; }
for rep in 1 2 3; do
  for c in c4 c5 c5x; do
    run $c PT_TUNE_PIPES=2
    run $c PT_TUNE_PIPES=3
    run $c PT_TUNE_PIPES=4
  done
done
sort $L | awk '{k=$1" "$2; v[k]=v[k]" "$3} END{for(k in v) print k, v[k]}' | sort
