#!/bin/bash
# round 4, GPU session A: the whole GPU suite (incl. the two new C3 full-size tests), the default bench line on the round's
# starting code, and the PMC picture of k_shade on C2 -- overlapped (three pipelines, the default) and alone (one pipeline).
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04a_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r04a_pytest.log
timeout 600 python bench.py > $O/r04a_bench_default.json 2> $O/r04a_bench_default.err; echo "bench rc=$?"
# the fused pipeline: headline shape, refill sweep, blocks per CU
for t in "" "refill=8" "refill=12" "refill=24" "refill=32" "refill=48" "extend_blocks=4" "extend_blocks=3"; do
  PT_TUNE="$t" timeout 300 python bench.py --pipeline fused --no-extra-legs --no-cpu-baseline --reps 3 > $O/r04a_fused_$(echo $t | tr '=' '_').json 2> $O/r04a_fused.err
  python - "$t" $O/r04a_fused_$(echo $t | tr '=' '_').json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("fused", sys.argv[1] or "default", d["value"], d["value_min"], d["value_max"], "ms/frame", d["ms_per_step"], "groups", d["config"]["sample_groups"], "fif", d["config"]["frames_in_flight"], "ws GB", round(d["workspace_bytes"] / 2**30, 2), "kernel us", d.get("roofline", {}).get("avg_launch_us"))
except Exception as e:
    print("fused", sys.argv[1], "ERR", e, open("gpurun_out/r04a_fused.err").read()[-600:])
PY
done
for mode in over alone; do
  [ $mode = alone ] && export PT_TUNE="pipes=1" || unset PT_TUNE
  PMC_EXTRA=1 bash scripts/gpu_profile.sh r04a_shade_$mode --steps 16 --warmup 0 --reps 1 --no-cpu-baseline --no-extra-legs > $O/r04a_prof_shade_$mode.log 2>&1
  python scripts/make_pmc_json.py $O/prof_r04a_shade_$mode $O/r04a_pmc_shade_c2_$mode.json "--steps 16 --no-extra-legs PT_TUNE=$PT_TUNE" --kernel=k_shade > /dev/null || echo "pmc json (shade, $mode) failed"
  python scripts/make_pmc_json.py $O/prof_r04a_shade_$mode $O/r04a_pmc_extend_c2_$mode.json "--steps 16 --no-extra-legs PT_TUNE=$PT_TUNE" > /dev/null || echo "pmc json (extend, $mode) failed"
  cp $O/prof_r04a_shade_$mode/summary.txt $O/r04a_shade_${mode}_rocprofv3_summary.txt
  rm -rf $O/prof_r04a_shade_$mode
done
unset PT_TUNE
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04a_bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["value_min"], d["value_max"], "frac", d["roofline"]["frac"], "c2_exact", d["c2_exact"]["mrays_per_s"], "lat", d["latency_ms_1frame"],
      "c4", d["roofline_c4"]["mrays_per_s"], "c5", d["roofline_c5"]["mrays_per_s"], d["roofline_c5"]["frac"], "c5x", d["roofline_c5x"]["mrays_per_s"],
      "\nfused leg:", json.dumps(d.get("c2_fused")),
      "\ncpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread_mrays"], d["cpu_baseline"]["scaling_efficiency"])
for m in ("over","alone"):
    for k in ("shade","extend"):
        try:
            p=json.load(open(f"gpurun_out/r04a_pmc_{k}_c2_{m}.json"))
            print(m, k, p["kernel"][:28], "us", round(p["rocprof_avg_launch_us"],1), "hbm B/ray", round(p["hbm_bytes_per_ray"],1), "GB/s", round(p["hbm_GBps"],1),
                  "valu/64", round(p["valu_wave_instr_per_64_rays"],1), "busy", round(p["valu_busy_fraction"],3), "wait", round(p["wait_any_fraction_of_wave_cycles"],3),
                  "utcl1 miss", p.get("utcl1_miss_rate"), "rdlat", p.get("l1_to_l2_read_latency_cycles"), "wrlat", p.get("l1_to_l2_write_latency_cycles"))
        except Exception as e: print(m, k, "ERR", e)
PY
