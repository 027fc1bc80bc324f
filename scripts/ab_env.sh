#!/bin/bash
# dev tool (GPU box): interleaved A/B of library builds and tuning environments.
# usage: ab_env.sh "bench args" spec...   with spec = name:lib.so.bin[:VAR=val,VAR=val]   (lib "-" = the tree's own build)
ARGS=$1; shift
L=single-file-vulkan-pathtracing_amd/libpt_amd.so
cp $L /tmp/keep.so
for r in $(seq 1 ${AB_ROUNDS:-3}); do
  for S in "$@"; do
    IFS=: read -r NAME LIB ENVS <<< "$S"
    if [ "$LIB" = "-" ]; then cp /tmp/keep.so $L; else cp $LIB $L; fi
    echo -n "$NAME: "
    env $(echo $ENVS | tr ',' ' ') python bench.py --no-cpu-baseline --no-extra-legs $ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); g=r.get('gather',{}); a=r.get('active_lanes',{})
print(d['value'], d['ms_per_step'], 'nodes/ray', g.get('bvh_nodes_per_ray'), 'tris/ray', g.get('tris_per_ray'), 'lanes node', a.get('node_steps'), 'tri', a.get('triangle_steps'), 'ext_ms', r.get('extend_ms'), 'sh_ms', r.get('shade_ms'), 'bvh4', d['bvh']['bvh4_nodes'])"
  done
done
cp /tmp/keep.so $L
