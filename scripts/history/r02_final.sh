#!/bin/bash
# round 2 final measurements (GPU box, repo root): default bench line (+ cpu baseline), rocprofv3 kernel stats of the very
# same command, bench lines of C4 / C5 / C5x, and PMC passes (one run per counter set) for the traversal kernel of each.
TAG=${1:-r02z}; O=gpurun_out; mkdir -p $O
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
  bash scripts/gpu_profile.sh ${TAG}_$c $extra $steps --warmup 0 --no-cpu-baseline --no-extra-legs > $O/${TAG}_prof_$c.log 2>&1
  python scripts/make_pmc_json.py $O/prof_${TAG}_$c $O/${TAG}_pmc_extend_$c.json "$extra $steps --no-extra-legs" > /dev/null || echo "pmc json failed for $c"
  cp $O/prof_${TAG}_$c/summary.txt $O/${TAG}_${c}_rocprofv3_summary.txt; cp $O/prof_${TAG}_$c/summary.json $O/${TAG}_${c}_rocprofv3_summary.json
  rm -rf $O/prof_${TAG}_$c
done
du -sh $O; ls $O | grep $TAG
