#!/bin/bash
TAG=${1:-r02k}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v -E "RCCL|HIP version|ROCm|Hostname|Librccl|amdgpu.ids" | tail -8 ) > $O/${TAG}_pytest.log
AB_ROUNDS=3 bash scripts/ab_env.sh "" base:ab/sort.so.bin frame:- > $O/${TAG}_ab_c2.log 2>&1
cat $O/${TAG}_pytest.log $O/${TAG}_ab_c2.log
