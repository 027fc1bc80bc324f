cd /root/repo
tag=r06z; mkdir -p gpurun_out/$tag; OUT=gpurun_out/$tag/c4_knobs_after_scalar_diet.txt
for r in 1 2; do
  for a in "" "node_yield=3" "node_yield=4" "node_yield=8" "leaf_min=10" "leaf_min=18" "enter_min=8" "enter_min=16" "refill=40" "refill=56"; do python scripts/probe_c4_fused.py $a >> $OUT 2>&1; done
done
cat $OUT
