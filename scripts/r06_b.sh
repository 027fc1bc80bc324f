# round 6: the shared spawn steps (B1) and the deferred pop (B2) of k_fused -- parity, then A/B on one box, then the block table
cd /root/repo
tag=${1:-r06b}
mkdir -p gpurun_out/$tag
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or full_size or auto_pipeline or c1_render" 2>&1 | tail -5 ) > gpurun_out/$tag/pytest.txt
cat gpurun_out/$tag/pytest.txt
for r in 1 2 3; do
  for v in "" build/variants/nodefer/libpt_amd.so build/variants/r05base/libpt_amd.so; do
    PT_LIB_AMD=$v python scripts/probe_ab_k16.py >> gpurun_out/$tag/ab.txt 2>&1
  done
done
cat gpurun_out/$tag/ab.txt
python scripts/dump_fused_blocks.py 16 > gpurun_out/$tag/blocks_k16.txt 2>&1
head -32 gpurun_out/$tag/blocks_k16.txt
