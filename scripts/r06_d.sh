cd /root/repo
tag=${1:-r06d}
mkdir -p gpurun_out/$tag
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or full_size or auto_pipeline or instanced" 2>&1 | tail -3 ) > gpurun_out/$tag/pytest.txt
cat gpurun_out/$tag/pytest.txt
for r in 1 2; do
  for v in "" build/variants/r05base/libpt_amd.so; do
    echo "== ${v:-product}" >> gpurun_out/$tag/c4.txt
    PT_LIB_AMD=$v python scripts/probe_c4_pipelines.py >> gpurun_out/$tag/c4.txt 2>&1
  done
done
cat gpurun_out/$tag/c4.txt
