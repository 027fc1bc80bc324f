#!/bin/bash
TAG=${1:-r02s}; O=gpurun_out; mkdir -p $O
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c4 --steps 8" prev:ab/inst16_prev.so.bin regs:- > $O/${TAG}_ab_c4.log 2>&1
( timeout 600 python -m pytest tests -q -m gpu -x -k "inst or c4" 2>&1 | grep -v -E "RCCL|HIP version|ROCm|Hostname|Librccl|amdgpu.ids" | tail -4 ) > $O/${TAG}_pytest.log
cat $O/${TAG}_ab_c4.log $O/${TAG}_pytest.log
