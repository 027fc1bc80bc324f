cd /root/repo
mkdir -p gpurun_out/r06i
for r in 1 2; do
  for v in "" fi640w5 fi320w5 fi256w5 fi256w4; do
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_c4_fused.py >> gpurun_out/r06i/c4_fused_occupancy.txt 2>&1
  done
done
cat gpurun_out/r06i/c4_fused_occupancy.txt
