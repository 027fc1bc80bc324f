cd /root/repo
tag=r06z; mkdir -p gpurun_out/$tag; OUT=gpurun_out/$tag/ab_inst16_loops.txt
( PT_LIB_AMD=build/variants/i16/libpt_amd.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "(instanced or c4 or inst) and not bench" 2>&1 | grep -E "passed|failed|error" | tail -2 ) >> $OUT
for r in 1 2 3; do
  for v in "" i16; do
    for cfg in "--config c4 --pipeline wavefront --mem-budget-mb 32768 --steps 8" "--config c4 --pipeline wavefront --steps 8"; do
      echo "== lib ${v:-product} $cfg" >> $OUT
      PT_LIB_AMD=${v:+build/variants/$v/libpt_amd.so} timeout 600 python bench.py $cfg --no-extra-legs --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $OUT
    done
  done
done
cat $OUT
