#!/bin/bash
TAG=${1:-r02g}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 ) > $O/${TAG}_pytest.log
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c5 --steps 4" scan4:-:PT_TUNE_TOPDOWN4=0 top4:- top4uni:-:PT_TUNE_UNIFIED=1 > $O/${TAG}_ab_c5.log 2>&1
AB_ROUNDS=1 bash scripts/ab_env.sh "--config c5x --steps 4" scan4:-:PT_TUNE_TOPDOWN4=0 top4:- top4uni:-:PT_TUNE_UNIFIED=1 > $O/${TAG}_ab_c5x.log 2>&1
cat $O/${TAG}_pytest.log $O/${TAG}_ab_c5.log $O/${TAG}_ab_c5x.log
