set -x
cd /root/repo
mkdir -p gpurun_out/r06a
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused" 2>&1 | tail -5 ) > gpurun_out/r06a/pytest_fused.txt
for r in 1 2 3; do
  python scripts/probe_ab_k16.py >> gpurun_out/r06a/ab.txt 2>&1
  PT_LIB_AMD=build/variants/r05base/libpt_amd.so python scripts/probe_ab_k16.py >> gpurun_out/r06a/ab.txt 2>&1
done
python scripts/dump_fused_blocks.py 16 > gpurun_out/r06a/blocks_k16.txt 2>&1
python scripts/dump_fused_blocks.py 1 > gpurun_out/r06a/blocks_k1.txt 2>&1
cat gpurun_out/r06a/pytest_fused.txt gpurun_out/r06a/ab.txt gpurun_out/r06a/blocks_k16.txt
