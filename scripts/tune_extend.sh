#!/bin/bash
# dev tool: sweep the refill threshold / residency of k_extend (env knobs are for tuning only)
run() { python bench.py "$@" --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['extend_ms'], r['shade_ms'])"; }
for r in 4 8 16 24 32 48; do echo -n "c2 lds refill=$r : "; PT_TUNE=refill=$r run --steps 8 --warmup 1 --extend lds; done

for f in 4 8 16; do echo -n "c2 lds fif=$f : "; run --steps 16 --warmup 1 --extend lds --frames-in-flight $f; done
for r in 8 16 32; do echo -n "c5 refill=$r : "; PT_TUNE=refill=$r run --config c5 --steps 4 --warmup 1; done
