#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/gpu_profile.sh <tag> [bench args...]
# Collects, with rocprofv3: (1) kernel-trace stats, (2..) PMC passes (each in its own run, with
# --kernel-trace only, as the pool's gpurun requires).  Raw output -> gpurun_out/prof_<tag>/,
# summary -> gpurun_out/prof_<tag>/summary.json (copy into profiles/ to commit it).
TAG=${1:-r01}; shift
ARGS=${@:-"--steps 2 --warmup 1 --no-cpu-baseline"}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, extra rocprof flags...
  local name=$1; shift
  rocprofv3 "$@" --kernel-trace -f csv -d $OUT/$name -o $name -- python bench.py $ARGS > $OUT/$name.log 2>&1
  echo "$name rc=$? : $(tail -c 300 $OUT/$name.log | tr '\n' ' ' | cut -c1-200)"
}
run stats --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run pmc_sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run pmc_sq3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
if [ -n "$PMC_EXTRA" ]; then  # the memory side in detail (round 4: k_shade's roofline): address translation, L1 stalls, L1<->L2 latency, L2<->fabric queues
  run pmc_utcl --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum
  run pmc_tcp1 --pmc TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum
  run pmc_tcp2 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
  run pmc_ea1 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
  run pmc_ea2 --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum
fi
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merge small: drop the big per-dispatch CSVs except stats
# keep the merge small (gpurun merges <= 64 MiB back): only the kernel-stats CSV and the summaries stay
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" -delete
find $OUT -name "*.db" -delete
du -sh $OUT
