cd /root/repo
tag=r06z; mkdir -p gpurun_out/$tag; OUT=gpurun_out/$tag/ab_shade_own_sgprs.txt
for r in 1 2; do
  for v in "" shade; do
    for cfg in "--pipeline wavefront --mem-budget-mb 0 --steps 10" "--config c4 --pipeline wavefront --mem-budget-mb 32768 --steps 8" "--config c5 --mem-budget-mb 32768 --steps 4"; do
      echo "== lib ${v:-product} $cfg" >> $OUT
      PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} timeout 600 python bench.py $cfg --no-extra-legs --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $OUT
    done
  done
done
cat $OUT
