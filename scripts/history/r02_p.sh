#!/bin/bash
TAG=${1:-r02p}; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for v in old new; do
  [ $v = old ] && export PT_TUNE_INST16=0 || export PT_TUNE_INST16=1
  export PT_TUNE_PIPES=1
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
    n=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --kernel-trace -f csv -d $O/pp_${v}_$n -o x -- python bench.py --config c4 --steps 4 --warmup 0 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
  done
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); nd=collections.defaultdict(set)
for f in glob.glob("$O/pp_${v}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_extend_inst" not in k or "<true" in k.split("k_extend_inst")[1][:8]: continue
        acc[k[:60]][r["Counter_Name"]]+=float(r["Counter_Value"]); nd[k[:60]].add(r["Dispatch_Id"])
for k,c in acc.items():
    print("$v", k, "dispatches", len(nd[k]))
    for n_,v_ in sorted(c.items()): print("   %-28s %.4g" % (n_, v_))
PY
  rm -rf $O/pp_${v}_*
done > $O/${TAG}_pmc_inst.log 2>&1
cat $O/${TAG}_pmc_inst.log
