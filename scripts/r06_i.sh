cd /root/repo
tag=r06m
mkdir -p gpurun_out/$tag
for r in 1 2; do
  python scripts/probe_c4_fused.py >> gpurun_out/$tag/c4_knobs.txt 2>&1
  for a in refill=36 refill=40 refill=44 refill=52 enter_min=8 enter_min=12 enter_min=24 leaf_min=4 leaf_min=12 leaf_min=16 node_yield=4 node_yield=8 node_yield=0 tlas_lds_kb=0 tlas_lds_kb=16 tlas_lds_kb=32; do
    python scripts/probe_c4_fused.py $a >> gpurun_out/$tag/c4_knobs.txt 2>&1
  done
done
cat gpurun_out/$tag/c4_knobs.txt
