#!/bin/bash
# round 3, session t: experiment -- traversal token between the two pipelines (stagger=2) vs free running, with timelines
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2; do for v in stagger=1 stagger=2; do echo -n "$v: "; PT_TUNE=$v python bench.py --steps 16 --warmup 2 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'min', d['value_min'], 'max', d['value_max'], 'ext_ms', r['extend_ms'], 'sh_ms', r['shade_ms'])"; done; done 2>&1 | tee $O/r03t_token.log
for v in 2; do
( cd /tmp && PT_TUNE=stagger=$v timeout 600 rocprofv3 --kernel-trace -f csv -d $O/trace_c2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 1 --reps 1 --no-cpu-baseline --no-extra-legs --no-kernel-events > /dev/null 2> $O/r03t.err )
f=$(find $O/trace_c2 -name "*kernel_trace.csv" | head -1)
python - "$f" $O/r03t_c2_timeline_stagger$v.csv <<'PY'
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
rm -rf $O/trace_c2
python scripts/timeline_stats.py $O/r03t_c2_timeline_stagger$v.csv 2 3
done
