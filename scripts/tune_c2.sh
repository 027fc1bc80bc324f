#!/bin/bash
# dev tool (GPU box): sweeps of the two run-time knobs of the Cornell pipeline: PT_TUNE=refill=N (idle lanes before a
# wave refills) and frames in flight.
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  for v in 8 12 16 20 24 32; do echo -n "refill $v: "; PT_TUNE=refill=$v one; done
  for f in 6 8 12 16; do echo -n "frames in flight $f: "; one --frames-in-flight $f; done
  echo -n "steps 32, fif 32: "; one --steps 32 --frames-in-flight 32
done
