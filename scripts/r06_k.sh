cd /root/repo
tag=r06q; mkdir -p gpurun_out/$tag
for r in 1 2; do
  for v in "" tb640w5 tb768w6 tb384w6 tb256w6; do
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_ab_env.py >> gpurun_out/$tag/ab_block_size.txt 2>&1
  done
done
cat gpurun_out/$tag/ab_block_size.txt
