#!/bin/bash
# round 3, session cn: which part of the Cornell kernels' queue traffic dislikes `nt`?  shade loads / shade stores / extend loads / extend stores alone, eight interleaved processes each
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=8 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" plain:ab/base.so.bin shade_ld:ab/sl.so.bin shade_st:ab/ss.so.bin ext_ld:ab/el.so.bin ext_st:ab/es.so.bin 2>&1 | tee $O/r03cn_ab_c2_nt_parts.log
python - <<'PY'
import re,statistics,collections
d=collections.defaultdict(list)
for l in open("gpurun_out/r03cn_ab_c2_nt_parts.log"):
    m=re.match(r"(\w+): ([\d.]+) ",l)
    if m: d[m.group(1)].append(float(m.group(2)))
for k,v in d.items(): print(k, "median %.0f mean %.0f min %.0f max %.0f n %d" % (statistics.median(v), statistics.mean(v), min(v), max(v), len(v)))
PY
