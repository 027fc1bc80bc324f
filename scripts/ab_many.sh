#!/bin/bash
# dev tool: compare several builds of libpt_amd.so on the same box, interleaved. usage: ab_many.sh "bench args" a.bin b.bin ...
ARGS=$1; shift
L=single-file-vulkan-pathtracing_amd/libpt_amd.so
cp $L /tmp/keep.so
for r in $(seq 1 ${AB_ROUNDS:-3}); do
  for B in "$@"; do
    cp $B $L; echo -n "$(basename $B): "
    python bench.py --no-cpu-baseline --no-extra-legs $ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); g=r.get('gather',{}); a=r.get('active_lanes',{})
print(d['value'], d['ms_per_step'], 'nodes/ray', g.get('bvh_nodes_per_ray'), 'tris/ray', g.get('tris_per_ray'), 'lanes node', a.get('node_steps'), 'tri', a.get('triangle_steps'), 'valu/64', r.get('valu_wave_instr_per_64_rays'), 'ext_ms', r.get('extend_ms'), 'sh_ms', r.get('shade_ms'))"
  done
done
cp /tmp/keep.so $L
