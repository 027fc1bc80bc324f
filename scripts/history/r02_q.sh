#!/bin/bash
TAG=${1:-r02q}; O=gpurun_out; mkdir -p $O
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c5 --steps 4" cnd:ab/inst16_nobfi_hbm.so.bin bfi:- > $O/${TAG}_ab_c5.log 2>&1
AB_ROUNDS=1 bash scripts/ab_env.sh "--config c4 --steps 8" b4r48:-:PT_TUNE_INST16_BLOCKS=4,PT_TUNE_REFILL=48 b5r48:-:PT_TUNE_REFILL=48 b4r40:-:PT_TUNE_INST16_BLOCKS=4,PT_TUNE_REFILL=40 b4r56:-:PT_TUNE_INST16_BLOCKS=4,PT_TUNE_REFILL=56 b4r48s8:-:PT_TUNE_INST16_BLOCKS=4,PT_TUNE_REFILL=48,PT_TUNE_LDS_STACK=8 b4r48s12:-:PT_TUNE_INST16_BLOCKS=4,PT_TUNE_REFILL=48,PT_TUNE_LDS_STACK=12 b4r44:-:PT_TUNE_INST16_BLOCKS=4,PT_TUNE_REFILL=44 b4r52:-:PT_TUNE_INST16_BLOCKS=4,PT_TUNE_REFILL=52 > $O/${TAG}_ab_c4.log 2>&1
cat $O/${TAG}_ab_c5.log $O/${TAG}_ab_c4.log
