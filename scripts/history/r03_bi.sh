#!/bin/bash
# round 3, session bi: fuzzers and the determinism soak on the final kernels of the last session (8-wide kernel's vote, LDS permutation table)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
(for s in "fuzz_trace.py 300 73000" "fuzz_instances.py 30 9900" "fuzz_render.py 300 12100"; do timeout 1500 python scripts/$s 2>&1 | tail -2; done; timeout 900 python scripts/soak_determinism.py 2>&1 | tail -6) | tee $O/r03bi_fuzz_soak.txt
