#!/bin/bash
# round 4, GPU session N: fuzzers and the determinism soak with the fused pipeline in their variant lists.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python scripts/fuzz_render.py 400 41000; timeout 600 python scripts/fuzz_trace.py 200 42000; timeout 600 python scripts/fuzz_instances.py 30 43000; timeout 900 python scripts/soak_determinism.py 12 ) 2>&1 | tee $O/r04n_fuzz_and_soak.txt | tail -20
