cd /root/repo
tag=r06z; mkdir -p gpurun_out/$tag; OUT=gpurun_out/$tag/refill_after_scalar_diet.txt
for r in 1 2; do for x in 28 32 36 40 44; do python scripts/probe_ab_env.py refill=$x >> $OUT 2>&1; done; done
cat $OUT
