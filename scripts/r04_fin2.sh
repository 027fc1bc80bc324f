#!/bin/bash
# round 4, second half of the final pass: the fused pipeline on its final code (the wavefront's kernels have not changed since r04fin1):
# suite, default line (carries c2_fused), fused bench lines, PMC of k_fused, fused shards, pt_main with both pipelines.
TAG=${1:-r04fin2}; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/${TAG}_smoke.log
timeout 900 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
timeout 600 python bench.py --pipeline fused --no-extra-legs --no-cpu-baseline > $O/${TAG}_bench_fused.json 2> $O/${TAG}_bench_fused.err
timeout 600 python bench.py --pipeline fused --config c3 --no-extra-legs --no-cpu-baseline > $O/${TAG}_bench_fused_c3_1gpu.json 2>> $O/${TAG}_bench_fused.err
timeout 600 python bench.py --pipeline fused --steps 2 --no-extra-legs --no-cpu-baseline > $O/${TAG}_bench_fused_k2.json 2>> $O/${TAG}_bench_fused.err
bash scripts/gpu_profile.sh ${TAG}_fused --pipeline fused --steps 16 --warmup 0 --reps 1 --no-cpu-baseline --no-extra-legs > $O/${TAG}_prof_fused.log 2>&1
python scripts/make_pmc_json.py $O/prof_${TAG}_fused $O/${TAG}_pmc_fused_c2.json "--pipeline fused --steps 16 --no-extra-legs" --kernel=k_fused > /dev/null || echo "pmc json (fused) failed"
cp $O/prof_${TAG}_fused/summary.txt $O/${TAG}_fused_rocprofv3_summary.txt; rm -rf $O/prof_${TAG}_fused
timeout 900 python scripts/probe_shard_efficiency.py 32 fused > $O/${TAG}_shard_efficiency_fused.json 2> $O/${TAG}_shard_efficiency_fused.err; cat $O/${TAG}_shard_efficiency_fused.err
P=single-file-vulkan-pathtracing_amd/pt_main
$P --width 640 --height 360 --frames 3 --pfm /tmp/wf.pfm > $O/${TAG}_pt_main_wavefront.txt 2>&1; $P --width 640 --height 360 --frames 3 --pipeline fused --pfm /tmp/fu.pfm > $O/${TAG}_pt_main_fused.txt 2>&1
cmp /tmp/wf.pfm /tmp/fu.pfm && echo "pt_main: fused image == wavefront image"; tail -2 $O/${TAG}_pt_main_fused.txt
python - $TAG <<'PY'
import json, sys
tag=sys.argv[1]
def line(f): return json.loads(open(f).read().strip().splitlines()[-1])
d=line(f"gpurun_out/{tag}_bench_default.json")
print("default:", d["value"], d["value_min"], d["value_max"], "ms", d["ms_per_step"], "ws GB", round(d["workspace_bytes"]/2**30,1), "c4", d["roofline_c4"]["mrays_per_s"], "c5", d["roofline_c5"]["mrays_per_s"], "c5x", d["roofline_c5x"]["mrays_per_s"], "frame0", d.get("frame0_film_bit_exact"), "\nfused leg:", json.dumps(d.get("c2_fused")))
for n in ("bench_fused","bench_fused_c3_1gpu","bench_fused_k2"):
    x=line(f"gpurun_out/{tag}_{n}.json"); print(n, x["value"], x["value_min"], x["value_max"], "ms/step", x["ms_per_step"], "groups", x["config"]["sample_groups"], "ws GB", round(x["workspace_bytes"]/2**30,2))
p=json.load(open(f"gpurun_out/{tag}_pmc_fused_c2.json"))
print("fused pmc", "hbm B/ray", round(p["hbm_bytes_per_ray"],2), "valu/64", round(p["valu_wave_instr_per_64_rays"],1), "lanes", round(p["valu_active_lanes_per_instr"],1), "wait", round(p["wait_any_fraction_of_wave_cycles"],3), "us", round(p["rocprof_avg_launch_us"],1))
PY
