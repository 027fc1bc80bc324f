#!/bin/bash
TAG=${1:-r02b}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > $O/${TAG}_pytest.log
bash scripts/ab_env.sh "" old:ab/r02a.so.bin keysort:-:PT_TUNE_PAIR_LEAVES=0 pairtree_trikernel:-:PT_TUNE_PAIR_KERNEL=0 pairs:- > $O/${TAG}_ab_c2.log 2>&1
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c4 --steps 8" old:ab/r02a.so.bin pairs:- keysort:-:PT_TUNE_PAIR_LEAVES=0 > $O/${TAG}_ab_c4.log 2>&1
cat $O/${TAG}_ab_c2.log $O/${TAG}_ab_c4.log
