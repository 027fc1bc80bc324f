#!/bin/bash
# round 4, GPU session AK: the final tree through the whole GPU suite, the three fuzzers at length and the determinism soak
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04ak_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/r04ak_pytest.log | tail -1
timeout 1200 python scripts/fuzz_trace.py 300 4100 2>&1 | tail -2 | tee $O/r04ak_fuzz.log
timeout 1500 python scripts/fuzz_instances.py 60 4400 2>&1 | tail -3 | tee -a $O/r04ak_fuzz.log
timeout 1500 python scripts/fuzz_render.py 400 4700 2>&1 | tail -2 | tee -a $O/r04ak_fuzz.log
timeout 1200 python scripts/soak_determinism.py 2>&1 | tail -4 | tee $O/r04ak_soak.log
