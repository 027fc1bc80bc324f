#!/bin/bash
TAG=${1:-r02e}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 ) > $O/${TAG}_pytest.log
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c5 --steps 4" bvh4:-:PT_TUNE_BVH8=0 bvh8:- bvh8_r16:-:PT_TUNE_REFILL=16 bvh8_r48:-:PT_TUNE_REFILL=48 bvh8_s8:-:PT_TUNE_LDS_STACK=8 > $O/${TAG}_ab_c5.log 2>&1
AB_ROUNDS=1 bash scripts/ab_env.sh "--config c5x --steps 4" bvh4:-:PT_TUNE_BVH8=0 bvh8:- > $O/${TAG}_ab_c5x.log 2>&1
cat $O/${TAG}_pytest.log $O/${TAG}_ab_c5.log $O/${TAG}_ab_c5x.log
