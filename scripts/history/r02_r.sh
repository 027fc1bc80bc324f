#!/bin/bash
TAG=${1:-r02r}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v -E "RCCL|HIP version|ROCm|Hostname|Librccl|amdgpu.ids" | tail -8 ) > $O/${TAG}_pytest.log
AB_ROUNDS=1 bash scripts/ab_env.sh "--config c4 --steps 8" def:- e8:-:PT_TUNE_ENTER_MIN=8 e24:-:PT_TUNE_ENTER_MIN=24 e32:-:PT_TUNE_ENTER_MIN=32 e1:-:PT_TUNE_ENTER_MIN=1 old:-:PT_TUNE_INST16=0 > $O/${TAG}_ab_c4.log 2>&1
( timeout 900 python scripts/fuzz_instances.py 20 9 2>&1 | tail -2 ) > $O/${TAG}_fuzz.log
cat $O/${TAG}_pytest.log $O/${TAG}_ab_c4.log $O/${TAG}_fuzz.log
