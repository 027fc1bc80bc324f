#!/bin/bash
TAG=${1:-r02l}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v -E "RCCL|HIP version|ROCm|Hostname|Librccl|amdgpu.ids" | tail -8 ) > $O/${TAG}_pytest.log
AB_ROUNDS=3 bash scripts/ab_env.sh "" frame:ab/frame.so.bin dense_w7:ab/dense_w7.so.bin dense_w6:ab/dense_w6.so.bin > $O/${TAG}_ab_c2.log 2>&1
AB_ROUNDS=1 bash scripts/ab_env.sh "--config c5 --steps 4" frame:ab/frame.so.bin dense_w7:ab/dense_w7.so.bin > $O/${TAG}_ab_c5.log 2>&1
AB_ROUNDS=1 bash scripts/ab_env.sh "--config c4 --steps 8" frame:ab/frame.so.bin dense_w7:ab/dense_w7.so.bin > $O/${TAG}_ab_c4.log 2>&1
cat $O/${TAG}_pytest.log $O/${TAG}_ab_c2.log $O/${TAG}_ab_c5.log $O/${TAG}_ab_c4.log
