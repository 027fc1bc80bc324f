cd /root/repo
tag=r06z; mkdir -p gpurun_out/$tag
V=${VARIANTS}; OUT=${OUT:-ab.txt}
for v in $V; do
  echo "== parity $v" >> gpurun_out/$tag/$OUT
  ( PT_LIB_AMD=build/variants/$v/libpt_amd.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "${KEXPR:-fused and not bench}" 2>&1 | tail -2 ) >> gpurun_out/$tag/$OUT
done
for r in 1 2 3; do
  for v in "" $V; do
    PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_ab_env.py >> gpurun_out/$tag/$OUT 2>&1
    [ -n "$C4" ] && PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} python scripts/probe_c4_fused.py >> gpurun_out/$tag/$OUT 2>&1
  done
done
cat gpurun_out/$tag/$OUT
