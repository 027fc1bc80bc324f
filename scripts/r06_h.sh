cd /root/repo
tag=r06l
mkdir -p gpurun_out/$tag
for r in 1 2; do
  for v in "" build/variants/exit3/libpt_amd.so build/variants/exit5/libpt_amd.so build/variants/exit6/libpt_amd.so; do
    PT_LIB_AMD=$v python scripts/probe_ab_env.py >> gpurun_out/$tag/ab.txt 2>&1
  done
  for rf in 32 36 44 48; do python scripts/probe_ab_env.py refill=$rf >> gpurun_out/$tag/ab.txt 2>&1; done
done
cat gpurun_out/$tag/ab.txt
