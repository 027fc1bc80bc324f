#!/bin/bash
# round 3, session p: is the C2 frame bimodal from process to process on one box?  (r03o: the old build sat at 21.3 Grays/s in
# three of four interleaved rounds while the new one read 23.4-24.0)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --steps 16 --warmup 2 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['values'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'])"; }
L=single-file-vulkan-pathtracing_amd/libpt_amd.so; cp $L /tmp/keep.so
for i in 1 2 3 4 5 6; do
  echo -n "new  $i: "; run
  cp ab/base_before_div.so.bin $L; echo -n "base $i: "; run; cp /tmp/keep.so $L
done 2>&1 | tee $O/r03p_c2_process_to_process.log
for q in 1 2 4; do echo -n "new GPU_MAX_HW_QUEUES=$q: "; GPU_MAX_HW_QUEUES=$q run; done 2>&1 | tee -a $O/r03p_c2_process_to_process.log
