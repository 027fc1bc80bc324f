#!/bin/bash
TAG=${1:-r02j}; O=gpurun_out; mkdir -p $O
AB_ROUNDS=3 bash scripts/ab_env.sh "" sortcommit:ab/sort.so.bin now:- > $O/${TAG}_ab_c2.log 2>&1
cat $O/${TAG}_ab_c2.log
