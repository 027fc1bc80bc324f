cd /root/repo
tag=${1:-r06c}
mkdir -p gpurun_out/$tag
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or full_size or auto_pipeline or c1_render" 2>&1 | tail -3 ) > gpurun_out/$tag/pytest.txt
cat gpurun_out/$tag/pytest.txt
for r in 1 2; do
  for v in "" build/variants/exit3/libpt_amd.so build/variants/exit5/libpt_amd.so build/variants/r05base/libpt_amd.so; do
    PT_LIB_AMD=$v python scripts/probe_ab_env.py >> gpurun_out/$tag/ab.txt 2>&1
  done
  for rf in 28 32 36 44 48; do python scripts/probe_ab_env.py refill=$rf >> gpurun_out/$tag/ab.txt 2>&1; done
done
cat gpurun_out/$tag/ab.txt
python scripts/dump_fused_blocks.py 16 > gpurun_out/$tag/blocks_k16.txt 2>&1
head -30 gpurun_out/$tag/blocks_k16.txt
