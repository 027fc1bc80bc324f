#!/bin/bash
# round 4, GPU session Q: k_fused_inst -- PT_PIPELINE_FUSED around the two-level walk: parity against the wavefront pipeline, then config C4.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python scripts/probe_fused_inst.py 2>&1 | tail -20 | tee $O/r04q_fused_inst_parity.log
for P in wavefront fused; do
  timeout 600 python bench.py --config c4 --pipeline $P --steps 8 --reps 3 --no-cpu-baseline --no-extra-legs 2>&1 | tail -1 > $O/r04q_bench_c4_$P.json
  python -c "
import json,sys; d=json.loads(open('$O/r04q_bench_c4_$P.json').read().strip().splitlines()[-1]); print('$P', d['value'], d['ms_per_step'], d['config'])"
done
