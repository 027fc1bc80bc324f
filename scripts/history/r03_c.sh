#!/bin/bash
# round 3, session c: PLOC builder -- GPU tests, the stress-scene probe, and the host CPU question of the cpu_baseline
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/probe_cpu_scaling.py > $O/r03c_cpu_scaling.txt 2>&1; tail -14 $O/r03c_cpu_scaling.txt
timeout 900 python scripts/probe_stress_scene.py > $O/r03c_probe_stress_scene.txt 2>&1; echo "probe rc=$?"; cat $O/r03c_probe_stress_scene.txt | tail -8
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03c_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r03c_pytest.txt
