#!/bin/bash
# round 3, session r: kernel timeline of the C2 frame (rocprofv3 --kernel-trace): how the two pipelines' launches overlap
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -f csv -d $O/trace_c2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 1 --reps 1 --no-cpu-baseline --no-extra-legs --no-kernel-events > $O/r03r_bench.json 2> $O/r03r.err )
f=$(find $O/trace_c2 -name "*kernel_trace.csv" | head -1); echo $f; head -2 $f | cut -c1-600
python - "$f" $O/r03r_c2_timeline.csv <<'PY'
import csv, sys, gzip
rows=list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
out=[]
for r in rows:
    n=r["Kernel_Name"]
    short="extend" if "k_extend" in n else "shade" if "k_shade" in n else "generate" if "k_generate" in n else "resolve" if "k_resolve" in n else "other"
    out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id","?"), r.get("Stream_Id","?")))
out.sort()
t0=out[0][0]
with open(sys.argv[2],"w") as f:
    f.write("start_us,end_us,kernel,queue,stream\n")
    for s,e,k,q,st in out: f.write("%.2f,%.2f,%s,%s,%s\n" % ((s-t0)/1e3,(e-t0)/1e3,k,q,st))
PY
rm -rf $O/trace_c2; ls -la $O/r03r*; tail -1 $O/r03r_bench.json | cut -c1-300
