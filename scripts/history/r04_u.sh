#!/bin/bash
# round 4, GPU session U: the whole GPU suite on the tree with k_fused_inst; the render fuzzer with two-level configurations
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04u_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r04u_pytest.log
timeout 900 python scripts/fuzz_render.py 120 7000 2>&1 | tail -5 | tee $O/r04u_fuzz_render.log
