cd /root/repo
tag=r06y; mkdir -p gpurun_out/$tag
V=${VARIANTS:-"lds1 lds2"}; OUT=${OUT:-ab_extra_lds.txt}
for r in 1 2 3; do
  for v in "" $V; do
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_ab_env.py >> gpurun_out/$tag/$OUT 2>&1
    [ -n "$C4" ] && PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_c4_fused.py >> gpurun_out/$tag/$OUT 2>&1
  done
done
cat gpurun_out/$tag/$OUT
