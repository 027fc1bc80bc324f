cd /root/repo
tag=r06z; mkdir -p gpurun_out/$tag; OUT=gpurun_out/$tag/${OUTNAME:-ab_c5_scalar_diet.txt}
V=${VARIANT:-c5salu}
( PT_LIB_AMD=build/variants/$V/libpt_amd.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "${KEXPR:-c5 or hbm or soup or wide8 or extend8 or trace}" 2>&1 | tail -2 ) >> $OUT
for r in 1 2 3; do
  for v in "" $V; do
    for cfg in "--config c5 --mem-budget-mb 32768 --steps 4" "--config c5x --mem-budget-mb 32768 --steps 4"; do
      echo "== lib ${v:-product} $cfg" >> $OUT
      PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} timeout 600 python bench.py $cfg --no-extra-legs --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $OUT
    done
  done
done
cat $OUT
