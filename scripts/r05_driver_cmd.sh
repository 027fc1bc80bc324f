#!/bin/bash
# The driver's own command: python bench.py --gpus 1 --steps 20 --warmup 5, N processes, value / extend_ms / shade_ms /
# workspace of each; optionally interleaved with an older build of the library on the same box (PT_LIB_AMD).
#   scripts/r05_driver_cmd.sh TAG N [variant dir name under build/variants] [extra bench flags]
TAG=$1; N=${2:-10}; VAR=$3; shift 3
O=gpurun_out; mkdir -p $O
row() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print(sys.argv[2], d["value"], "min", d["value_min"], "max", d["value_max"], "extend_ms", r.get("extend_ms"), "shade_ms", r.get("shade_ms"),
          "ws_GB", round(d["workspace_bytes"] / 2**30, 1), "fif", d["config"]["frames_in_flight"], "groups", d["config"]["sample_groups"], "pipes", d["config"]["pipelines"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for i in $(seq 1 $N); do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline "$@" > $O/${TAG}_head_$i.json 2>/dev/null; row $O/${TAG}_head_$i.json "head $i"
  if [ -n "$VAR" ] && [ "$VAR" != "-" ]; then
    PT_LIB_AMD=$PWD/build/variants/$VAR/libpt_amd.so timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline "$@" > $O/${TAG}_${VAR}_$i.json 2>/dev/null; row $O/${TAG}_${VAR}_$i.json "$VAR $i"
  fi
done
