cd /root/repo
tag=r06w; mkdir -p gpurun_out/$tag
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or full_size or instanced or c4" 2>&1 | tail -2 ) | tee gpurun_out/$tag/pytest2.txt
for r in 1 2 3; do
  for v in "" instnomerge; do
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_c4_fused.py >> gpurun_out/$tag/ab_mergediv_c4.txt 2>&1
  done
done
cat gpurun_out/$tag/ab_mergediv_c4.txt
