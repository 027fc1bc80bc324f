#!/bin/bash
TAG=${1:-r02d}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > $O/${TAG}_pytest.log
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c5 --steps 4" vote:-:PT_TUNE_UNIFIED=0 uni:- uni16:-:PT_TUNE_REFILL=16 uni24:-:PT_TUNE_REFILL=24 uni48:-:PT_TUNE_REFILL=48 > $O/${TAG}_ab_c5.log 2>&1
AB_ROUNDS=1 bash scripts/ab_env.sh "--config c5x --steps 4" vote:-:PT_TUNE_UNIFIED=0 uni:- > $O/${TAG}_ab_c5x.log 2>&1
cat $O/${TAG}_pytest.log $O/${TAG}_ab_c5.log $O/${TAG}_ab_c5x.log
