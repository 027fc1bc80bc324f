#!/bin/bash
TAG=${1:-r02h}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v -E "RCCL|HIP version|ROCm|Hostname|Librccl" | tail -8 ) > $O/${TAG}_pytest.log
for c in c5x c5; do for srt in off on; do for b in 6 4; do [ $srt = off ] && [ $b = 4 ] && continue; echo -n "$c sort $srt bits $b: "; PT_TUNE_SORT_BITS=$b python bench.py --config $c --steps 4 --no-cpu-baseline --sort-rays $srt 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], 'ext', r['extend_ms'], 'sh', r['shade_ms'], 'dev_ms', d['device_ms_rank0'])"; done; done; done > $O/${TAG}_sort.log 2>&1
cat $O/${TAG}_pytest.log $O/${TAG}_sort.log
