#!/bin/bash
# dev tool: A/B two builds of libpt_amd.so on the same box. usage: ab_bench.sh base.bin new.bin [bench args]
A=$1; B=$2; shift 2
L=single-file-vulkan-pathtracing_amd/libpt_amd.so
one() { python bench.py --no-cpu-baseline --no-extra-legs "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for r in 1 2 3; do
  cp $A $L; echo -n "base: "; one "$@"
  cp $B $L; echo -n "new:  "; one "$@"
done
