#!/bin/bash
run() { python bench.py "$@" --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['extend_ms'], r['shade_ms'])"; }
for it in 1 2 4; do for b in 4 8 16; do echo -n "shade items=$it bpc=$b : "; PT_TUNE_SHADE_ITEMS=$it PT_TUNE_SHADE_BPC=$b run --steps 8 --warmup 1; done; done
