#!/bin/bash
TAG=${1:-r02f}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x -k "bvh8 or soup or c5" 2>&1 | tail -5 ) > $O/${TAG}_pytest.log
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c5 --steps 4" bvh4:-:PT_TUNE_BVH8=0 bvh4vote:-:PT_TUNE_BVH8=0,PT_TUNE_UNIFIED=0 bvh8w6:- bvh8w5:ab/e8w5.so.bin bvh8w4:ab/e8w4.so.bin > $O/${TAG}_ab_c5.log 2>&1
AB_ROUNDS=1 bash scripts/ab_env.sh "--config c5x --steps 4" bvh4vote:-:PT_TUNE_BVH8=0,PT_TUNE_UNIFIED=0 bvh8w6:- bvh8w5:ab/e8w5.so.bin > $O/${TAG}_ab_c5x.log 2>&1
cat $O/${TAG}_pytest.log $O/${TAG}_ab_c5.log $O/${TAG}_ab_c5x.log
