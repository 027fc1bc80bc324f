#!/bin/bash
# dev tool: compare several builds of libpt_amd.so on the same box, interleaved. usage: ab_many.sh "bench args" a.bin b.bin ...
ARGS=$1; shift
L=single-file-vulkan-pathtracing_amd/libpt_amd.so
cp $L /tmp/keep.so
for r in $(seq 1 ${AB_ROUNDS:-3}); do
  for B in "$@"; do
    cp $B $L; echo -n "$(basename $B): "
    python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('roofline',{}).get('gather',{})
print(d['value'], d['ms_per_step'], 'node_occ', g.get('lane_occupancy_node_steps'), 'tri_occ', g.get('lane_occupancy_triangle_steps'), 'ext_ms', d.get('roofline',{}).get('extend_ms'), 'sh_ms', d.get('roofline',{}).get('shade_ms'))"
  done
done
cp /tmp/keep.so $L
