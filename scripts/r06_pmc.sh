# round 6: one rocprofv3 pass per counter set for one kernel (scripts/gpu_profile.sh) -> profiles-ready record.  usage: r06_pmc.sh fused_c2|fused_c4|extend_c2|extend_c4|extend_c5|extend_c5x
cd /root/repo
O=gpurun_out
what=$1
case $what in
  fused_c2)  args="--pipeline fused --steps 16"; kern="--kernel=k_fused" ;;
  fused_c4)  args="--config c4 --pipeline fused --steps 8"; kern="--kernel=k_fused_inst" ;;
  extend_c2) args="--pipeline wavefront --steps 16"; kern="" ;;
  extend_c4) args="--config c4 --pipeline wavefront --steps 8"; kern="" ;;
  extend_c5) args="--config c5 --steps 4"; kern="" ;;
  extend_c5x) args="--config c5x --steps 2"; kern="" ;;
esac
bash scripts/gpu_profile.sh r06_$what $args --warmup 0 --reps 1 --no-cpu-baseline --no-extra-legs --full-line > $O/r06_prof_$what.log 2>&1
python scripts/make_pmc_json.py $O/prof_r06_$what $O/r06_pmc_$what.json "$args --no-extra-legs" $kern > /dev/null || echo "pmc json failed for $what"
[ $what = extend_c2 ] && ( python scripts/make_pmc_json.py $O/prof_r06_$what $O/r06_pmc_shade_c2.json "$args --no-extra-legs" --kernel=k_shade > /dev/null || echo "pmc json (shade) failed" )
python - <<PY
import json
p=json.load(open("$O/r06_pmc_$what.json"))
print("$what", p["kernel"][:40], "valu/64", round(p["valu_wave_instr_per_64_rays"],1), "lanes", round(p["valu_active_lanes_per_instr"],1), "issue", round(p["valu_issue_frac"],3), "wait", round(p["wait_any_fraction_of_wave_cycles"],3), "hbm B/ray", round(p["hbm_bytes_per_ray"],2), "us", round(p["rocprof_avg_launch_us"],1), round(p["bench_hipext_avg_launch_us_same_run"],1))
PY
grep -h '"metric"' $O/prof_r06_$what/stats.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench model:', r.get('instr_per_64_rays'), 'lanes', r.get('lanes'), 'frac', r.get('frac'))" 2>/dev/null
