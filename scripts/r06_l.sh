cd /root/repo
tag=r06r; mkdir -p gpurun_out/$tag
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or full_size or c1_render" 2>&1 | tail -2 ) | tee gpurun_out/$tag/pytest.txt
for r in 1 2 3; do
  for v in "" nospecpop; do
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_ab_env.py >> gpurun_out/$tag/ab_specpop.txt 2>&1
  done
done
cat gpurun_out/$tag/ab_specpop.txt
