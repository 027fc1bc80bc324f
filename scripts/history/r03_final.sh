#!/bin/bash
# round 3 final measurements (GPU box, repo root): default bench line (+ cpu baseline), rocprofv3 kernel stats of the very
# same command, bench lines of C4 / C5 / C5x, and PMC passes (one run per counter set) for the traversal kernel of each.
TAG=${1:-r03fin}; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof_${TAG}_default -o stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/${TAG}_bench_default_rocprof.json 2>$GRAFT_REPO_ROOT/$O/${TAG}_rocprof.err )
find $O/prof_${TAG}_default -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_default_kernel_stats.csv
rm -rf $O/prof_${TAG}_default
for c in c4 c5 c5x; do
  timeout 900 python bench.py --config $c $( [ $c = c4 ] && echo "--steps 8" || echo "--steps 4" ) > $O/${TAG}_bench_$c.json 2> $O/${TAG}_bench_$c.err
done
for c in c2 c4 c5 c5x; do
  extra="--config $c"; steps="--steps 4"; [ $c = c2 ] && steps="--steps 16"; [ $c = c4 ] && steps="--steps 8"
  bash scripts/gpu_profile.sh ${TAG}_$c $extra $steps --warmup 0 --reps 1 --no-cpu-baseline --no-extra-legs > $O/${TAG}_prof_$c.log 2>&1
  python scripts/make_pmc_json.py $O/prof_${TAG}_$c $O/${TAG}_pmc_extend_$c.json "$extra $steps --no-extra-legs" > /dev/null || echo "pmc json failed for $c"
  cp $O/prof_${TAG}_$c/summary.txt $O/${TAG}_${c}_rocprofv3_summary.txt; cp $O/prof_${TAG}_$c/summary.json $O/${TAG}_${c}_rocprofv3_summary.json
  rm -rf $O/prof_${TAG}_$c
done
du -sh $O; ls $O | grep $TAG
python - $TAG <<'PY'
import json, sys
tag=sys.argv[1]
d=json.loads(open(f"gpurun_out/{tag}_bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["value_min"], d["value_max"], "frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for c in ("c2","c4","c5","c5x"):
    try:
        p=json.load(open(f"gpurun_out/{tag}_pmc_extend_{c}.json"))
        print(c, p["kernel"][:24], "hbm B/ray", round(p["hbm_bytes_per_ray"],1), "rec64", round(p["hbm_read_bytes_per_ray_records64"],1), "valu/64", round(p["valu_wave_instr_per_64_rays"],1), "lanes", round(p["valu_active_lanes_per_instr"],1), "busy", round(p["valu_busy_fraction"],3), "wait", round(p["wait_any_fraction_of_wave_cycles"],3), "l2hit", round(p["l2_hit_rate"],3), "us", round(p["rocprof_avg_launch_us"],1), round(p["bench_hipext_avg_launch_us_same_run"],1))
    except Exception as e: print(c, "ERR", e)
PY
