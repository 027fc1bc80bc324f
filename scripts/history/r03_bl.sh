#!/bin/bash
# round 3, session bl: LDS stack entries of the no-spill 8-wide kernel at 6 and 7 waves per SIMD, C5 and C5x
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python - <<'PY' 2>&1 | tee $O/r03bl_levels.txt
import importlib
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
for n in (1000000, 8000000):
    v, i, f = pt.make_soup(n, 1)
    sc = pt.Scene(ctx, v, i, f)
    sc.trace(__import__("numpy").zeros((64, 6), "float32") + 1, extend=pt.EXTEND_HBM8)
    info = sc.info()
    print(n, "wide8_levels", info.wide8_levels, "n_wide8_nodes", info.n_wide8_nodes)
    sc.close()
PY
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" ns:ab/ns.so.bin ns_l11:ab/ns.so.bin:lds_stack=11 ns_l10:ab/ns.so.bin:lds_stack=10 ns_l9:ab/ns.so.bin:lds_stack=9 ns_l8:ab/ns.so.bin:lds_stack=8 ns_l7:ab/ns.so.bin:lds_stack=7 2>&1 | tee $O/r03bl_ab_c5_lds_stack.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" ns:ab/ns.so.bin ns_l10:ab/ns.so.bin:lds_stack=10 ns_l9:ab/ns.so.bin:lds_stack=9 ns7_l10:ab/ns7.so.bin:lds_stack=10 ns7_l9:ab/ns7.so.bin:lds_stack=9 2>&1 | tee -a $O/r03bl_ab_c5_lds_stack.log
