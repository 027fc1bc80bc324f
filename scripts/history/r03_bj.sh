#!/bin/bash
# round 3, session bj: fuzzers with PT_EXTEND_HBM8 among the variants (it was missing from both lists), default tuning and two
# other settings of the vote; the knob test on both HBM kernels
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "vote_knobs" 2>&1 | tail -3
 for s in "fuzz_trace.py 300 81000" "fuzz_render.py 300 13300"; do timeout 1500 python scripts/$s 2>&1 | tail -4; done
 PT_TUNE="tri_enter=3,refill=5" timeout 1500 python scripts/fuzz_trace.py 150 82000 2>&1 | tail -3
 PT_TUNE="tri_enter=40,refill=64,lds_stack=2" timeout 1500 python scripts/fuzz_trace.py 150 83000 2>&1 | tail -3) | tee $O/r03bj_fuzz_hbm8.txt
