#!/bin/bash
# round 4, GPU session V: instance fuzzer with the render check (both pipelines), 60 scenes
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python scripts/fuzz_instances.py 60 9100 2>&1 | tail -12 | tee $O/r04v_fuzz_instances.log
