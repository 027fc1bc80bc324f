#!/bin/bash
# round 4, GPU session AO: k_extend8 with NO bound on pending groups (PT_E8_NO_BOUND: the model says the bound never culls): parity, then C5 / C5x against the tree's build
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=single-file-vulkan-pathtracing_amd/libpt_amd.so; cp $L /tmp/keep5.so; cp build/e8_nb.so.bin $L
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bvh8 or soup or hbm8 or 8wide or full_size or big or sort" 2>&1 | grep -E "passed|failed" | tee $O/r04ao_pytest.log
timeout 600 python scripts/fuzz_trace.py 60 8800 2>&1 | tail -1 | tee -a $O/r04ao_pytest.log
cp /tmp/keep5.so $L
AB_ROUNDS=3 bash scripts/ab_many.sh "--config c5 --steps 4 --reps 3" build/e8_base.so.bin build/e8_nb.so.bin 2>&1 | tee $O/r04ao_ab_e8_nb_c5.log
AB_ROUNDS=3 bash scripts/ab_many.sh "--config c5x --steps 2 --reps 3" build/e8_base.so.bin build/e8_nb.so.bin 2>&1 | tee $O/r04ao_ab_e8_nb_c5x.log
