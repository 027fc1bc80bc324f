#!/bin/bash
# round 2, first GPU session: the whole test-suite, the default bench line, rocprofv3 kernel stats of the same command,
# and the C4 / C5 / C5x lines.  usage (GPU box, repo root): bash scripts/r02_first.sh <tag>
TAG=${1:-r02a}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > $O/${TAG}_pytest.log
timeout 600 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
tail -c 600 $O/${TAG}_bench_default.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof_${TAG}_default -o stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/${TAG}_bench_default_rocprof.json 2>$GRAFT_REPO_ROOT/$O/${TAG}_rocprof.err )
find $O/prof_${TAG}_default -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_default_kernel_stats.csv
find $O/prof_${TAG}_default -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O/prof_${TAG}_default -name "*.db" -delete
for c in c4 c5 c5x; do
  timeout 900 python bench.py --config $c $( [ $c = c4 ] && echo "--steps 8" || echo "--steps 4" ) > $O/${TAG}_bench_$c.json 2> $O/${TAG}_bench_$c.err
done
du -sh $O
