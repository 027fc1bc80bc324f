cd /root/repo
tag=r06w; mkdir -p gpurun_out/$tag
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or full_size or c1_render or reference_dispatch" 2>&1 | tail -2 ) | tee gpurun_out/$tag/pytest.txt
for r in 1 2 3; do
  for v in "" head_prev; do
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_ab_env.py >> gpurun_out/$tag/ab_dummy_level.txt 2>&1
  done
done
cat gpurun_out/$tag/ab_dummy_level.txt
