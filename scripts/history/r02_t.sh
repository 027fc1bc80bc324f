#!/bin/bash
# C2 shape sweep at small slot counts: do the queues of a small shape stay in the 256 MiB Infinity Cache, and does that pay?
O=gpurun_out; mkdir -p $O; L=$O/${1:-r02t}_shapes.txt; : > $L
for sh in "0 0" "1 1" "1 2" "2 1" "2 2" "4 1" "1 4" "4 4" "8 2"; do
  set -- $sh
  echo "== fif=$1 groups=$2" >> $L
  timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --frames-in-flight $1 --sample-groups $2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], 'extend_ms', r['extend_ms'], 'shade_ms', r['shade_ms'], 'rays/launch', r['rays_per_launch'], 'launches', r['launches'])
" >> $L
done
cat $L
