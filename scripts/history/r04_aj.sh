#!/bin/bash
# round 4, GPU session AJ: the surface-area builder's top kernel by ranks + scans: the trees of a set of small scenes equal the two-launch builder's
# (row hashes, diffed), the build times, the tests that read trees
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=single-file-vulkan-pathtracing_amd/libpt_amd.so; cp $L /tmp/keep4.so
for B in sah_old sah_new; do cp build/$B.so.bin $L; python scripts/probe_sah_trees.py > $O/r04aj_trees_$B.txt 2>&1; done
cp /tmp/keep4.so $L
diff <(sed 's/ *#.*//' $O/r04aj_trees_sah_old.txt) <(sed 's/ *#.*//' $O/r04aj_trees_sah_new.txt) && echo "trees equal" | tee $O/r04aj_diff.txt
paste -d'|' $O/r04aj_trees_sah_old.txt $O/r04aj_trees_sah_new.txt | sed 's/pair_leaves \([01]\) \([a-z0-9]*\).*# \(.*\)|.*# \(.*\)/\1 \2 old \3 new \4/' | tee $O/r04aj_times.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "sah or bvh or tree or cornell" 2>&1 | grep -E "passed|failed" | tee $O/r04aj_pytest.log
