#!/bin/bash
# round 3, session ao: kernel timelines of several film allocations in one process: what distinguishes a 23.1 from a 26.7 Grays/s film?
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace -f csv -d $O/trace_ao -o t -- python $GRAFT_REPO_ROOT/scripts/probe_alloc_modes.py 8 > $O/r03ao_probe.txt 2> $O/r03ao.err )
f=$(find $O/trace_ao -name "*kernel_trace.csv" | head -1)
python - "$f" $O/r03ao_timeline.csv <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
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
rm -rf $O/trace_ao
cat $O/r03ao_probe.txt | cut -c1-300
python scripts/timeline_stats.py $O/r03ao_timeline.csv > $O/r03ao_timeline_stats.txt; grep -c call $O/r03ao_timeline_stats.txt
