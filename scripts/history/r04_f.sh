#!/bin/bash
# round 4, GPU session F: the allocation levels of the wavefront C2 shape under counters.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
N=${N_ALLOC:-10}; R=3
python scripts/probe_alloc_pmc.py $N $R > $O/r04f_alloc_plain.txt 2>&1; cat $O/r04f_alloc_plain.txt
i=0
for set in "GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_EA0_WRREQ_STALL_sum" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -f csv -d $O/prof_r04f_$i -o p -- python scripts/probe_alloc_pmc.py $N $R > $O/r04f_alloc_pmc$i.txt 2>&1
  f=$(find $O/prof_r04f_$i -name "*counter_collection.csv" | head -1)
  python scripts/probe_alloc_pmc.py analyse $f $R > $O/r04f_alloc_pmc$i.json 2>> $O/r04f_alloc_pmc$i.txt
  rm -rf $O/prof_r04f_$i
  grep "^alloc" $O/r04f_alloc_pmc$i.txt | head -12
  python - $O/r04f_alloc_pmc$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for a, v in d.items():
    s = v.get("shade", {})
    print("  alloc", a, " ".join(f"{k}={x:.4g}" for k, x in sorted(s.items())))
PY
done
# the fused pipeline's shards after the guided batch sizes; C4 with the TLAS rebuilt by PLOC
timeout 900 python scripts/probe_shard_efficiency.py 32 fused > $O/r04_shard_efficiency_fused.json 2> $O/r04_shard_efficiency_fused.err; cat $O/r04_shard_efficiency_fused.err
timeout 300 python -m pytest tests -m gpu -x -q -k "fused or instanc or c4" > $O/r04f_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r04f_pytest.log
for t in "" "tlas_ploc=1" "tlas_ploc=1,ploc_adopt_pct=1000" "tlas_ploc=1,ploc_adopt_pct=1000,ploc_radius=16"; do
  PT_TUNE="$t" timeout 300 python bench.py --config c4 --steps 8 --reps 3 --no-extra-legs --no-cpu-baseline > $O/r04f_c4_$(echo $t | tr '=,' '__').json 2> $O/r04f_c4.err
  python - "$t" $O/r04f_c4_$(echo $t | tr '=,' '__').json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("c4", sys.argv[1] or "default", d["value"], d["value_min"], d["value_max"], "nodes/ray", r["gather"]["bvh_nodes_per_ray"], "tris/ray", r["gather"]["tris_per_ray"], "valu/64", r.get("valu_wave_instr_per_64_rays"), "lanes", r.get("valu_active_lanes_per_instr"), "ext us", r["avg_launch_us"])
except Exception as e:
    print("c4", sys.argv[1], "ERR", e, open("gpurun_out/r04f_c4.err").read()[-500:])
PY
done
