#!/bin/bash
# round 4, GPU session AF: shard efficiency at the DEFAULT bench's size (K = 16 frames, what `bench.py --gpus N` times), both pipelines
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python scripts/probe_shard_efficiency.py 16 wavefront > $O/r04af_shard_efficiency_k16_wavefront.json 2> $O/r04af_k16_wavefront.err; cat $O/r04af_k16_wavefront.err
timeout 900 python scripts/probe_shard_efficiency.py 16 fused > $O/r04af_shard_efficiency_k16_fused.json 2> $O/r04af_k16_fused.err; cat $O/r04af_k16_fused.err
python - <<'PY'
import json
for p in ("wavefront", "fused"):
    d = json.load(open(f"gpurun_out/r04af_shard_efficiency_k16_{p}.json"))
    for w, e in d["worlds"].items():
        r = e["ranks"][0]
        print(p, "world", w, "eff", e["efficiency"], "slowest", e["slowest_rank_ms"], "shape", r["frames_in_flight"], r["sample_groups"], r["pipelines"], "ws GB", round(r["workspace_bytes"] / 1e9, 2))
PY
