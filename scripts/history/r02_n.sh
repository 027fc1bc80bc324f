#!/bin/bash
TAG=${1:-r02n}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v -E "RCCL|HIP version|ROCm|Hostname|Librccl|amdgpu.ids" | tail -12 ) > $O/${TAG}_pytest.log
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c4 --steps 8" old:-:PT_TUNE_INST16=0 inst16:- r48:-:PT_TUNE_REFILL=48 r32:-:PT_TUNE_REFILL=32 s8:-:PT_TUNE_LDS_STACK=8 > $O/${TAG}_ab_c4.log 2>&1
( timeout 900 python scripts/fuzz_instances.py 20 5 2>&1 | tail -3 ) > $O/${TAG}_fuzz.log
cat $O/${TAG}_pytest.log $O/${TAG}_ab_c4.log $O/${TAG}_fuzz.log
