cd /root/repo
tag=r06s; mkdir -p gpurun_out/$tag
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "instanced or c4 or two_level or full_size or fused" 2>&1 | tail -2 ) | tee gpurun_out/$tag/pytest.txt
for r in 1 2 3; do
  for v in "" nospecpop; do
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_c4_fused.py >> gpurun_out/$tag/ab_c4.txt 2>&1
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_c4_wavefront.py >> gpurun_out/$tag/ab_c4.txt 2>&1
  done
done
cat gpurun_out/$tag/ab_c4.txt
