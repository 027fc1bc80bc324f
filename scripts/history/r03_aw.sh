#!/bin/bash
# round 3, session aw: fuzzers and the determinism soak on the round's final kernels and schedule
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
(for s in "fuzz_trace.py 200 61000" "fuzz_instances.py 30 8800" "fuzz_render.py 300 9700"; do timeout 1500 python scripts/$s 2>&1 | tail -2; done; timeout 900 python scripts/soak_determinism.py 2>&1 | tail -6) | tee $O/r03aw_fuzz_soak.txt
