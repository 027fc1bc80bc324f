#!/bin/bash
# round 3, session n: exact short division x prim-ids-on-ties x v_mbcnt rank in the Cornell kernel -- interleaved A/B on C2
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "cornell or c1 or c2 or spirv or independent or coincident or fuzz or tie or full_size" > $O/r03n_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03n_pytest.txt
AB_ROUNDS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" base:ab/base_before_div.so.bin div8:ab/div8.so.bin div8m:ab/div8m.so.bin div9:ab/div9.so.bin div9m:ab/div9m.so.bin div3:ab/div3.so.bin div3m:ab/div3m.so.bin 2>&1 | tee $O/r03n_ab_c2_exact_div.log
