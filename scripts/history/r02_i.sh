#!/bin/bash
TAG=${1:-r02i}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v -E "RCCL|HIP version|ROCm|Hostname|Librccl|amdgpu.ids" | tail -25 ) > $O/${TAG}_pytest.log
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c2', d['value'], r['extend_ms'], r['shade_ms'], r.get('valu_wave_instr_per_64_rays'))"; done > $O/${TAG}_c2.log 2>&1
for c in c5 c4; do python bench.py --no-cpu-baseline --config $c --steps 4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$c', d['value'], r['extend_ms'], r['shade_ms'])"; done >> $O/${TAG}_c2.log 2>&1
cat $O/${TAG}_pytest.log $O/${TAG}_c2.log
