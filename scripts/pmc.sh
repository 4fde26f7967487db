#!/bin/bash
# PMC passes on the rollout kernel (profiling aid): instruction and wait counters, each set in its own rocprofv3 run
# (--pmc with --kernel-trace only).  Usage: bash scripts/pmc.sh <tag> [time_rollout.py arguments, e.g. --n-veh 64 --f16]
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python scripts/time_rollout.py --iters 40 "$@" > $OUT/pmc_$name.log 2>&1
done
python scripts/pmc_summary.py $OUT | tee $OUT/summary.txt
