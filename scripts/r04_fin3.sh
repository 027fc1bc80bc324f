#!/bin/bash
# round 4 final measurements, third pass (after k_fused_inst, the tile-major hand-out, the hoisted record loads, the two-launch surface-area builder, the live counter passes)
# round 4 final measurements (GPU box, repo root): the GPU suite, the default bench line (+ cpu baseline), rocprofv3 kernel stats
# of the very same command, bench lines of C3 (one GPU) / C4 / C5 / C5x and of the fused pipeline, PMC passes (one run per counter
# set) for the traversal kernel of every config, for k_shade and for k_fused, the C3 shard probes.
TAG=${1:-r04fin4}; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/${TAG}_pytest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/${TAG}_smoke.log
timeout 900 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof_${TAG}_default -o stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/${TAG}_bench_default_rocprof.json 2>$GRAFT_REPO_ROOT/$O/${TAG}_rocprof.err )
find $O/prof_${TAG}_default -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_default_kernel_stats.csv
rm -rf $O/prof_${TAG}_default
timeout 600 python bench.py --config c3 --no-extra-legs --no-cpu-baseline > $O/${TAG}_bench_c3_1gpu.json 2> $O/${TAG}_bench_c3.err
timeout 600 python bench.py --pipeline fused --no-extra-legs --no-cpu-baseline > $O/${TAG}_bench_fused.json 2> $O/${TAG}_bench_fused.err
timeout 600 python bench.py --pipeline fused --config c3 --no-extra-legs --no-cpu-baseline > $O/${TAG}_bench_fused_c3_1gpu.json 2>> $O/${TAG}_bench_fused.err
timeout 600 python bench.py --pipeline fused --steps 2 --no-extra-legs --no-cpu-baseline > $O/${TAG}_bench_fused_k2.json 2>> $O/${TAG}_bench_fused.err
timeout 600 python bench.py --pipeline fused --config c4 --steps 8 --no-extra-legs --no-cpu-baseline > $O/${TAG}_bench_fused_c4.json 2>> $O/${TAG}_bench_fused.err
for c in c4 c5 c5x; do
  timeout 900 python bench.py --config $c $( [ $c = c4 ] && echo "--steps 8" || echo "--steps 4" ) > $O/${TAG}_bench_$c.json 2> $O/${TAG}_bench_$c.err
done
for c in c2 c4 c5 c5x; do
  extra="--config $c"; steps="--steps 4"; [ $c = c2 ] && steps="--steps 16"; [ $c = c4 ] && steps="--steps 8"
  [ $c = c2 ] && export PMC_EXTRA=1 || unset PMC_EXTRA
  bash scripts/gpu_profile.sh ${TAG}_$c $extra $steps --warmup 0 --reps 1 --no-cpu-baseline --no-extra-legs > $O/${TAG}_prof_$c.log 2>&1
  python scripts/make_pmc_json.py $O/prof_${TAG}_$c $O/${TAG}_pmc_extend_$c.json "$extra $steps --no-extra-legs" > /dev/null || echo "pmc json failed for $c"
  [ $c = c2 ] && ( python scripts/make_pmc_json.py $O/prof_${TAG}_$c $O/${TAG}_pmc_shade_c2.json "$extra $steps --no-extra-legs" --kernel=k_shade > /dev/null || echo "pmc json (shade) failed" )
  cp $O/prof_${TAG}_$c/summary.txt $O/${TAG}_${c}_rocprofv3_summary.txt; cp $O/prof_${TAG}_$c/summary.json $O/${TAG}_${c}_rocprofv3_summary.json
  rm -rf $O/prof_${TAG}_$c
done
unset PMC_EXTRA
bash scripts/gpu_profile.sh ${TAG}_fused --pipeline fused --steps 16 --warmup 0 --reps 1 --no-cpu-baseline --no-extra-legs > $O/${TAG}_prof_fused.log 2>&1
python scripts/make_pmc_json.py $O/prof_${TAG}_fused $O/${TAG}_pmc_fused_c2.json "--pipeline fused --steps 16 --no-extra-legs" --kernel=k_fused > /dev/null || echo "pmc json (fused) failed"
cp $O/prof_${TAG}_fused/summary.txt $O/${TAG}_fused_rocprofv3_summary.txt; rm -rf $O/prof_${TAG}_fused
bash scripts/gpu_profile.sh ${TAG}_fused_c4 --config c4 --pipeline fused --steps 8 --warmup 0 --reps 1 --no-cpu-baseline --no-extra-legs > $O/${TAG}_prof_fused_c4.log 2>&1
python scripts/make_pmc_json.py $O/prof_${TAG}_fused_c4 $O/${TAG}_pmc_fused_c4.json "--config c4 --pipeline fused --steps 8 --no-extra-legs" --kernel=k_fused_inst > /dev/null || echo "pmc json (fused c4) failed"
cp $O/prof_${TAG}_fused_c4/summary.txt $O/${TAG}_fused_c4_rocprofv3_summary.txt; rm -rf $O/prof_${TAG}_fused_c4
timeout 900 python scripts/probe_shard_efficiency.py 32 wavefront > $O/${TAG}_shard_efficiency.json 2> $O/${TAG}_shard_efficiency.err; cat $O/${TAG}_shard_efficiency.err
timeout 900 python scripts/probe_shard_efficiency.py 32 fused > $O/${TAG}_shard_efficiency_fused.json 2> $O/${TAG}_shard_efficiency_fused.err; cat $O/${TAG}_shard_efficiency_fused.err
du -sh $O; ls $O | grep $TAG | wc -l
python - $TAG <<'PY'
import json, sys
tag=sys.argv[1]
def line(f):
    return json.loads(open(f).read().strip().splitlines()[-1])
d=line(f"gpurun_out/{tag}_bench_default.json")
print("default:", d["value"], d["value_min"], d["value_max"], "ms", d["ms_per_step"], "ws GB", round(d["workspace_bytes"]/2**30,1), "frac", d["roofline"]["frac"], "shade", d["roofline_shade"]["frac"], "c2_exact", d["c2_exact"]["mrays_per_s"], "lat", d["latency_ms_1frame"],
      "c4", d["roofline_c4"]["mrays_per_s"], "c5", d["roofline_c5"]["mrays_per_s"], d["roofline_c5"]["frac"], "c5x", d["roofline_c5x"]["mrays_per_s"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["scaling_efficiency"],
      "frame0", d.get("frame0_film_bit_exact"), "\nfused leg:", json.dumps(d.get("c2_fused")))
for n in ("bench_c3_1gpu","bench_fused","bench_fused_k2","bench_fused_c3_1gpu","bench_fused_c4","bench_c4","bench_c5","bench_c5x"):
    try:
        x=line(f"gpurun_out/{tag}_{n}.json"); print(n, x["value"], x["value_min"], x["value_max"], "ms/step", x["ms_per_step"], "ws GB", round(x["workspace_bytes"]/2**30,1))
    except Exception as e: print(n, "ERR", e)
for c in ("extend_c2","shade_c2","fused_c2","extend_c4","fused_c4","extend_c5","extend_c5x"):
    try:
        p=json.load(open(f"gpurun_out/{tag}_pmc_{c}.json"))
        print(c, p["kernel"][:24], "hbm B/ray", round(p["hbm_bytes_per_ray"],1), "GB/s", round(p["hbm_GBps"],1), "valu/64", round(p["valu_wave_instr_per_64_rays"],1), "lanes", round(p["valu_active_lanes_per_instr"],1), "busy", round(p["valu_busy_fraction"],3), "wait", round(p["wait_any_fraction_of_wave_cycles"],3), "l2hit", p["l2_hit_rate"] and round(p["l2_hit_rate"],3), "us", round(p["rocprof_avg_launch_us"],1), round(p["bench_hipext_avg_launch_us_same_run"],1))
    except Exception as e: print(c, "ERR", e)
PY
