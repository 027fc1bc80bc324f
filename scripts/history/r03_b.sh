#!/bin/bash
# round 3, session b (GPU box, repo root): the whole GPU test-suite on the cleaned-up library, the new default bench line,
# and the calibration once more with the 128-B-line pattern added.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/r03b
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03b_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r03b_pytest.txt
timeout 900 python bench.py > $O/r03b_bench_default.json 2> $O/r03b_bench_default.err; echo "bench rc=$?"
cd /tmp
BIN=$GRAFT_REPO_ROOT/scripts/ubench/bin/fetch_calib
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr A-Z a-z | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --kernel-trace -f csv -d $O/r03b/calib_$d -o calib -- $BIN > $O/r03b/calib_stdout_$d.txt 2>&1
done
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -f csv -d $O/r03b/calib_rdreq -o calib -- $BIN > $O/r03b/calib_stdout_rdreq.txt 2>&1
cp $O/r03b/calib_stdout_fetch.txt $O/r03b/calib_stdout.txt
cd $GRAFT_REPO_ROOT
python scripts/calib_fetch.py $O/r03b $O/r03b_fetch_size_calibration.json > $O/r03b_calib_summary.txt 2>&1
find $O/r03b -name "*.db" -delete; find $O/r03b -name "*kernel_trace.csv" -delete
grep gather_line $O/r03b_calib_summary.txt
