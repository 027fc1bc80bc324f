cd /root/repo
tag=r06z; mkdir -p gpurun_out/$tag
V=${VARIANTS}; OUT=${OUT:-ab.txt}
for r in 1 2 3; do
  for v in "" $V; do
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} timeout 300 python scripts/probe_ab_env.py >> gpurun_out/$tag/$OUT 2>&1
  done
done
cat gpurun_out/$tag/$OUT
