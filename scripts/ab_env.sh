#!/bin/bash
# dev tool (GPU box): interleaved A/B of library builds and tunings (include/pt_api.h pt_tuning via PT_TUNE).
# usage: ab_env.sh "bench args" spec...   with spec = name:lib.so.bin[:knob=val,knob=val]   (lib "-" = the tree's own build)
ARGS=$1; shift
L=single-file-vulkan-pathtracing_amd/libpt_amd.so
cp $L /tmp/keep.so
for r in $(seq 1 ${AB_ROUNDS:-3}); do
  for S in "$@"; do
    IFS=: read -r NAME LIB TUNE <<< "$S"
    if [ "$LIB" = "-" ]; then cp /tmp/keep.so $L; else cp $LIB $L; fi
    echo -n "$NAME: "
    PT_TUNE="$TUNE" python bench.py --no-cpu-baseline --no-extra-legs --reps ${AB_REPS:-3} $ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); g=r.get('gather',{}); a=r.get('active_lanes',{})
print(d['value'], '[%s..%s]' % (d.get('value_min'), d.get('value_max')), d['ms_per_step'], 'nodes/ray', g.get('bvh_nodes_per_ray'), 'tris/ray', g.get('tris_per_ray'), 'lanes node', a.get('node_steps'), 'leaf', a.get('leaf_steps'), 'valu/64', r.get('valu_wave_instr_per_64_rays'), 'lanes/instr', r.get('valu_active_lanes_per_instr'), 'ext_ms', r.get('extend_ms'), 'sh_ms', r.get('shade_ms'))"
  done
done
cp /tmp/keep.so $L
