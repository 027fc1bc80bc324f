#!/bin/bash
# round 3, session e: PLOC after the stall-rule fix -- stress probe (radius 8 and 16), GPU tests
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 8 16; do
  PT_TUNE=ploc_radius=$r timeout 900 python scripts/probe_stress_scene.py > $O/r03e_probe_stress_scene_r$r.txt 2>&1; echo "probe r=$r rc=$?"; grep -v soup $O/r03e_probe_stress_scene_r$r.txt | tail -3
done
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03e_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r03e_pytest.txt; grep -n "AssertionError" -A3 $O/r03e_pytest.txt | head
