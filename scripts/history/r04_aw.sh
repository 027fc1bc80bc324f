#!/bin/bash
# round 4, GPU session AW: long fuzz / soak of the final tree
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python scripts/fuzz_trace.py 1000 12000 2>&1 | tail -1 | tee $O/r04aw_fuzz_long.log
timeout 2400 python scripts/fuzz_render.py 2000 15000 2>&1 | tail -1 | tee -a $O/r04aw_fuzz_long.log
timeout 3000 python scripts/fuzz_instances.py 200 18000 2>&1 | tail -2 | tee -a $O/r04aw_fuzz_long.log
timeout 1800 python scripts/soak_determinism.py 60 2>&1 | tail -2 | tee -a $O/r04aw_fuzz_long.log
