#!/bin/bash
# round 4, GPU session T: the tests of the fused pipeline's two-level kernel; the default line's C4 leg with its fused record
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "fused or full_size" 2>&1 | tail -5 | tee $O/r04t_pytest_fused.log
timeout 900 python bench.py --no-cpu-baseline --c5-frames 0 --c5x-frames 0 > $O/r04t_bench_c4leg.json 2> $O/r04t_bench_c4leg.err
python -c "
import json; d=json.loads(open('$O/r04t_bench_c4leg.json').read().strip().splitlines()[-1]); l=d['roofline_c4']; print(d['value'], l['mrays_per_s'], l.get('fused'))"
