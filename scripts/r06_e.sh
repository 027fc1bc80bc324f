cd /root/repo
tag=${1:-r06f}
mkdir -p gpurun_out/$tag
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/$tag/pytest.txt
cat gpurun_out/$tag/pytest.txt
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > gpurun_out/$tag/smoke.txt
cat gpurun_out/$tag/smoke.txt
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/bench.out 2> gpurun_out/$tag/bench.err ); tail -c 2500 gpurun_out/$tag/bench.out; tail -5 gpurun_out/$tag/bench.err
