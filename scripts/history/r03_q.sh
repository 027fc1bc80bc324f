#!/bin/bash
# round 3, session q: C2 process to process, old build vs new, default HW queues vs GPU_MAX_HW_QUEUES=2/3/8
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --steps 16 --warmup 2 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'])"; }
L=single-file-vulkan-pathtracing_amd/libpt_amd.so; cp $L /tmp/keep.so
for i in 1 2 3 4; do
  for q in 4 2 3 8; do
    echo -n "new  q=$q $i: "; GPU_MAX_HW_QUEUES=$q run
    cp ab/base_before_div.so.bin $L; echo -n "base q=$q $i: "; GPU_MAX_HW_QUEUES=$q run; cp /tmp/keep.so $L
  done
done 2>&1 | tee $O/r03q_c2_hw_queues.log
