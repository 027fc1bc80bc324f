cd /root/repo
tag=r06m; mkdir -p gpurun_out/$tag
for r in 1 2; do
  python scripts/probe_c4_fused.py >> gpurun_out/$tag/c4_knobs2.txt 2>&1
  for a in "leaf_min=10" "leaf_min=12" "leaf_min=14" "leaf_min=20" "leaf_min=24" "leaf_min=12 enter_min=8" "leaf_min=12 refill=52" "leaf_min=12 enter_min=8 refill=52 tlas_lds_kb=16" "leaf_min=14 enter_min=12"; do
    python scripts/probe_c4_fused.py $a >> gpurun_out/$tag/c4_knobs2.txt 2>&1
  done
done
cat gpurun_out/$tag/c4_knobs2.txt
