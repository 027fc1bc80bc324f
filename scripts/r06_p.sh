cd /root/repo
tag=r06x; mkdir -p gpurun_out/$tag
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or full_size or instanced or c4 or reference" 2>&1 | tail -2 ) | tee gpurun_out/$tag/pytest.txt
for r in 1 2 3; do
  for v in "" cont; do
    echo "== lib ${v:-product}" >> gpurun_out/$tag/ab_one_latch.txt
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_ab_env.py >> gpurun_out/$tag/ab_one_latch.txt 2>&1
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_c4_fused.py >> gpurun_out/$tag/ab_one_latch_c4.txt 2>&1
  done
done
cat gpurun_out/$tag/ab_one_latch.txt gpurun_out/$tag/ab_one_latch_c4.txt
