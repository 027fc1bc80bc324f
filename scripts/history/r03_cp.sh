#!/bin/bash
# round 3, session cp: `nt` on the term-log stores (T) and on k_generate's queue stores as well (TG), C2 at K = 16 and K = 2, C4
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=6 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" base:ab/base.so.bin T:ab/T.so.bin TG:ab/TG.so.bin 2>&1 | tee $O/r03cp_ab_nt_terms_generate.log
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--steps 2 --warmup 1" base:ab/base.so.bin T:ab/T.so.bin TG:ab/TG.so.bin 2>&1 | tee -a $O/r03cp_ab_nt_terms_generate.log
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" base:ab/base.so.bin TG:ab/TG.so.bin 2>&1 | tee -a $O/r03cp_ab_nt_terms_generate.log
