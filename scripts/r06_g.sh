# round 6: the BVH4 cut by dynamic programming (bvh4_sah_device.hip) -- new fixture rows, parity, A/B against the greedy cut (greedycut = HEAD before it)
cd /root/repo
tag=r06k
mkdir -p gpurun_out/$tag
python tests/golden/make_sah_rows.py && cp tests/golden/sah_rows.npz gpurun_out/$tag/sah_rows.npz
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or full_size or sah or bvh4 or c1_render or instanced or trace" 2>&1 | tail -5 ) > gpurun_out/$tag/pytest.txt
cat gpurun_out/$tag/pytest.txt
for r in 1 2 3; do
  for v in "" build/variants/greedycut/libpt_amd.so; do
    PT_LIB_AMD=$v python scripts/probe_ab_env.py >> gpurun_out/$tag/ab.txt 2>&1
    PT_LIB_AMD=$v python scripts/probe_c4_fused.py >> gpurun_out/$tag/ab_c4.txt 2>&1
  done
done
cat gpurun_out/$tag/ab.txt gpurun_out/$tag/ab_c4.txt
python scripts/dump_fused_blocks.py 16 > gpurun_out/$tag/blocks_k16.txt 2>&1
head -30 gpurun_out/$tag/blocks_k16.txt
