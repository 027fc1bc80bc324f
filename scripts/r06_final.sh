#!/bin/bash
# round 6 final measurements (scripts/r05_final.sh adapted to the compact bench line: the driver's command is run as the driver runs it, its full record is
# taken from gpurun_out/bench_last_full.json; every other line with --full-line)
# round 5 final measurements (GPU box, repo root): the GPU suite + smoke; the DRIVER'S OWN command ten times (value = PT_PIPELINE_AUTO = the fused
# kernel for C2; the wavefront leg and every extra leg inside the line); rocprofv3 kernel stats of that command; bench lines of C3 (one GPU) / C4 /
# C5 / C5x and of the explicit pipelines; PMC passes (one run per counter set) for k_fused, k_fused_inst, the wavefront's two kernels and the
# traversal kernels of C4 / C5 / C5x; the C3 shard probes; pt_main.
#   bash scripts/r06_final.sh TAG [N driver-command processes, default 10] [skip: list of parts to skip, e.g. "pmc shard"]
TAG=${1:-r06fin}; N=${2:-6}; SKIP=" ${3:-} "; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
has() { case "$SKIP" in *" $1 "*) return 1;; *) return 0;; esac; }
if has suite; then
  timeout 1500 python -m pytest tests -m gpu -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/${TAG}_pytest.log | tail -1
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/${TAG}_smoke.log
fi
if has driver; then
  for i in $(seq 1 $N); do
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_driver_$i.out 2> $O/${TAG}_driver_$i.err
    cp $O/bench_last_full.json $O/${TAG}_driver_$i.json
    tail -1 $O/${TAG}_driver_$i.out | wc -c
    python - $O/${TAG}_driver_$i.json $i <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    w = d.get("wavefront") or {}
    print("driver", sys.argv[2], "value", d["value"], "min", d["value_min"], "max", d["value_max"], "pipeline", d["config"].get("pipeline"), "ws_GB", round(d["workspace_bytes"] / 2**30, 2),
          "| wavefront leg", w.get("mrays_per_s"), "extend_ms", (d.get("roofline_wavefront") or {}).get("extend_ms"), "shade_ms", (d.get("roofline_shade") or {}).get("shade_ms"), "ws_GB", round((w.get("workspace_bytes") or 0) / 2**30, 1),
          "| c2_exact", (d.get("c2_exact") or {}).get("mrays_per_s"), "lat_ms", d.get("latency_ms_1frame"), "c4", (d.get("roofline_c4") or {}).get("mrays_per_s"), "c4 fused", ((d.get("roofline_c4") or {}).get("fused") or {}).get("mrays_per_s"),
          "c5", (d.get("roofline_c5") or {}).get("mrays_per_s"), "c5x", (d.get("roofline_c5x") or {}).get("mrays_per_s"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "frame0", d.get("frame0_film_bit_exact"))
except Exception as e:
    print("driver", sys.argv[2], "failed", e)
PY
  done
  cp $O/${TAG}_driver_1.json $O/${TAG}_bench_default.json
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof_${TAG}_default -o stats -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --full-line > $GRAFT_REPO_ROOT/$O/${TAG}_bench_default_rocprof.json 2>$GRAFT_REPO_ROOT/$O/${TAG}_rocprof.err )
  find $O/prof_${TAG}_default -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_default_kernel_stats.csv
  rm -rf $O/prof_${TAG}_default
fi
if has lines; then
  timeout 600 python bench.py --pipeline wavefront --mem-budget-mb 0 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --full-line > $O/${TAG}_bench_wavefront.json 2> $O/${TAG}_bench_wavefront.err
  timeout 600 python bench.py --config c3 --no-extra-legs --no-cpu-baseline --full-line > $O/${TAG}_bench_c3_1gpu.json 2> $O/${TAG}_bench_c3.err
  timeout 600 python bench.py --config c3 --pipeline wavefront --mem-budget-mb 0 --no-extra-legs --no-cpu-baseline --full-line > $O/${TAG}_bench_c3_wavefront_1gpu.json 2>> $O/${TAG}_bench_c3.err
  timeout 600 python bench.py --steps 2 --no-extra-legs --no-cpu-baseline --full-line > $O/${TAG}_bench_k2.json 2> $O/${TAG}_bench_k2.err
  timeout 600 python bench.py --steps 1 --no-extra-legs --no-cpu-baseline --full-line > $O/${TAG}_bench_k1.json 2>> $O/${TAG}_bench_k2.err
  timeout 600 python bench.py --config c4 --steps 8 --no-cpu-baseline --full-line > $O/${TAG}_bench_c4.json 2> $O/${TAG}_bench_c4.err
  timeout 600 python bench.py --config c4 --steps 8 --pipeline wavefront --mem-budget-mb 32768 --no-extra-legs --no-cpu-baseline --full-line > $O/${TAG}_bench_c4_wavefront.json 2>> $O/${TAG}_bench_c4.err
  timeout 600 python bench.py --config c4 --steps 8 --pipeline fused --no-extra-legs --no-cpu-baseline --full-line > $O/${TAG}_bench_c4_fused.json 2>> $O/${TAG}_bench_c4.err
  timeout 600 python bench.py --config c4 --steps 16 --no-extra-legs --no-cpu-baseline --full-line > $O/${TAG}_bench_c4_k16.json 2>> $O/${TAG}_bench_c4.err
  for c in c5 c5x; do timeout 900 python bench.py --config $c --steps 4 --mem-budget-mb 32768 --full-line > $O/${TAG}_bench_$c.json 2> $O/${TAG}_bench_$c.err; done
fi
if has pmc; then
  prof() { # tag-suffix, kernel prefix for the record, record name, bench args...
    local sfx=$1 kern=$2 rec=$3; shift 3
    bash scripts/gpu_profile.sh ${TAG}_$sfx "$@" --warmup 0 --reps 1 --no-cpu-baseline --no-extra-legs --full-line > $O/${TAG}_prof_$sfx.log 2>&1
    python scripts/make_pmc_json.py $O/prof_${TAG}_$sfx $O/${TAG}_pmc_$rec.json "$*" --kernel=$kern > /dev/null || echo "pmc json failed for $rec"
    cp $O/prof_${TAG}_$sfx/summary.txt $O/${TAG}_${sfx}_rocprofv3_summary.txt; cp $O/prof_${TAG}_$sfx/summary.json $O/${TAG}_${sfx}_rocprofv3_summary.json
  }
  prof fused k_fused fused_c2 --steps 16
  rm -rf $O/prof_${TAG}_fused
  export PMC_EXTRA=1
  prof c2 k_extend extend_c2 --pipeline wavefront --mem-budget-mb 0 --steps 16
  python scripts/make_pmc_json.py $O/prof_${TAG}_c2 $O/${TAG}_pmc_shade_c2.json "--pipeline wavefront --mem-budget-mb 0 --steps 16" --kernel=k_shade > /dev/null || echo "pmc json (shade) failed"
  rm -rf $O/prof_${TAG}_c2
  unset PMC_EXTRA
  prof fused_c4 k_fused_inst fused_c4 --config c4 --steps 8 --pipeline fused
  rm -rf $O/prof_${TAG}_fused_c4
  prof c4 k_extend extend_c4 --config c4 --pipeline wavefront --mem-budget-mb 32768 --steps 8
  rm -rf $O/prof_${TAG}_c4
  prof c5 k_extend extend_c5 --config c5 --mem-budget-mb 32768 --steps 4
  rm -rf $O/prof_${TAG}_c5
  prof c5x k_extend extend_c5x --config c5x --mem-budget-mb 32768 --steps 4
  rm -rf $O/prof_${TAG}_c5x
fi
if has shard; then
  timeout 900 python scripts/probe_shard_efficiency.py 32 auto > $O/${TAG}_shard_efficiency.json 2> $O/${TAG}_shard_efficiency.err; cat $O/${TAG}_shard_efficiency.err
  timeout 900 python scripts/probe_shard_efficiency.py 16 auto > $O/${TAG}_shard_efficiency_k16.json 2> $O/${TAG}_shard_efficiency_k16.err; cat $O/${TAG}_shard_efficiency_k16.err
  timeout 900 python scripts/probe_shard_efficiency.py 32 wavefront > $O/${TAG}_shard_efficiency_wavefront.json 2> $O/${TAG}_shard_efficiency_wavefront.err; cat $O/${TAG}_shard_efficiency_wavefront.err
fi
if has ptmain; then
  timeout 300 single-file-vulkan-pathtracing_amd/pt_main --width 1920 --height 1080 --frames 16 > $O/${TAG}_pt_main.json 2>&1; tail -1 $O/${TAG}_pt_main.json | cut -c1-400
  timeout 300 single-file-vulkan-pathtracing_amd/pt_main --width 1920 --height 1080 --frames 16 --pipeline wavefront >> $O/${TAG}_pt_main.json 2>&1; tail -1 $O/${TAG}_pt_main.json | cut -c1-400
fi
du -sh $O; ls $O | grep $TAG | wc -l
python - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
def line(f):
    return json.loads([l for l in open(f) if l.startswith("{")][-1])
for n in ("bench_wavefront", "bench_c3_1gpu", "bench_c3_wavefront_1gpu", "bench_k2", "bench_k1", "bench_c4", "bench_c4_wavefront", "bench_c5", "bench_c5x"):
    try:
        x = line(f"gpurun_out/{tag}_{n}.json")
        r = x.get("roofline") or {}
        print(n, x["value"], x["value_min"], x["value_max"], "ms/step", x["ms_per_step"], "ws GB", round(x["workspace_bytes"] / 2**30, 2), x["config"].get("pipeline"), "| roofline", r.get("kernel"), r.get("frac"), r.get("frac_counted"),
              r.get("bound"), r.get("instr_per_64_rays"), r.get("lanes"), r.get("valu_frac"))
    except Exception as e:
        print(n, "ERR", e)
for c in ("fused_c2", "extend_c2", "shade_c2", "fused_c4", "extend_c4", "extend_c5", "extend_c5x"):
    try:
        p = json.load(open(f"gpurun_out/{tag}_pmc_{c}.json"))
        print(c, p["kernel"][:24], "hbm B/ray", round(p["hbm_bytes_per_ray"], 2), "GB/s", round(p["hbm_GBps"], 1), "valu/64", round(p["valu_wave_instr_per_64_rays"], 1), "lanes", round(p["valu_active_lanes_per_instr"], 1),
              "issue", round(p["valu_issue_frac"], 3), "wait", round(p["wait_any_fraction_of_wave_cycles"], 3), "l2hit", p["l2_hit_rate"] and round(p["l2_hit_rate"], 3), "us", round(p["rocprof_avg_launch_us"], 1), round(p["bench_hipext_avg_launch_us_same_run"], 1))
    except Exception as e:
        print(c, "ERR", e)
PY
