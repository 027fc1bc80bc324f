#!/bin/bash
# dev tool (GPU box): kernel timeline of one bench command -> gpurun_out/<tag>_timeline.csv + overlap summary
# usage: gpu_timeline.sh <tag> <bench args...>
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && PT_TUNE=$PT_TUNE_TL timeout 900 rocprofv3 --kernel-trace -f csv -d $O/trace_$TAG -o t -- python $GRAFT_REPO_ROOT/bench.py "$@" --reps 1 --no-cpu-baseline --no-extra-legs --no-kernel-events > $O/${TAG}_bench.json 2> $O/${TAG}.err )
f=$(find $O/trace_$TAG -name "*kernel_trace.csv" | head -1)
python - "$f" $O/${TAG}_timeline.csv <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
out=[]
for r in rows:
    n=r["Kernel_Name"]
    short="extend" if "k_extend" in n else "shade" if "k_shade" in n else "generate" if "k_generate" in n else "resolve" if "k_resolve" in n else "sort" if "k_rs_" in n or "sort" in n else "other"
    out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id","?"), r.get("Stream_Id","?")))
out.sort()
t0=out[0][0]
with open(sys.argv[2],"w") as f:
    f.write("start_us,end_us,kernel,queue,stream\n")
    for s,e,k,q,st in out: f.write("%.2f,%.2f,%s,%s,%s\n" % ((s-t0)/1e3,(e-t0)/1e3,k,q,st))
PY
rm -rf $O/trace_$TAG
python scripts/timeline_stats.py $O/${TAG}_timeline.csv | tail -8
