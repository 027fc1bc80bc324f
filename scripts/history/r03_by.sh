#!/bin/bash
# round 3, session by: render fuzz (AUTO shapes among its choices), instance fuzz and the determinism soak on the FINAL code
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
(for s in "fuzz_render.py 400 15100" "fuzz_trace.py 150 91000" "fuzz_instances.py 20 12700"; do timeout 1500 python scripts/$s 2>&1 | tail -2; done; timeout 900 python scripts/soak_determinism.py 2>&1 | tail -6) | tee $O/r03by_fuzz_soak_final.txt
