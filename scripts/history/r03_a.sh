#!/bin/bash
# round 3, session a (GPU box, repo root): FETCH_SIZE / WRITE_SIZE calibration on known byte counts,
# then the round-2 code's default bench line as this round's same-box baseline.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/r03a
export TMPDIR=/tmp
cd /tmp
BIN=$GRAFT_REPO_ROOT/scripts/ubench/bin/fetch_calib
timeout 300 $BIN > $O/r03a/calib_plain.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr A-Z a-z | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --kernel-trace -f csv -d $O/r03a/calib_$d -o calib -- $BIN > $O/r03a/calib_stdout_$d.txt 2>&1
  echo "pmc $c rc=$?"
done
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -f csv -d $O/r03a/calib_rdreq -o calib -- $BIN > $O/r03a/calib_stdout_rdreq.txt 2>&1
echo "pmc rdreq rc=$?"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -f csv -d $O/r03a/calib_tcc -o calib -- $BIN > $O/r03a/calib_stdout_tcc.txt 2>&1
echo "pmc tcc rc=$?"
cp $O/r03a/calib_stdout_fetch.txt $O/r03a/calib_stdout.txt
cd $GRAFT_REPO_ROOT
python scripts/calib_fetch.py $O/r03a $O/r03a_fetch_size_calibration.json > $O/r03a_calib_summary.txt 2>&1
cat $O/r03a_calib_summary.txt
find $O/r03a -name "*.db" -delete; find $O/r03a -name "*kernel_trace.csv" -delete
timeout 900 python bench.py --no-cpu-baseline > $O/r03a_bench_default.json 2> $O/r03a_bench_default.err
tail -c 600 $O/r03a_bench_default.json
du -sh $O/r03a
