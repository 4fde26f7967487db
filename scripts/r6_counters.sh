#!/bin/bash
# Round 6: instruction / wait counters of the per-step rollout kernel at the headline and at the 8-GPU shard size (scripts/pmc.sh), and the
# HBM-resident case (8 env sets round-robin) under the launch schedules
TAG=${1:-r6cnt}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
bash scripts/pmc.sh ${TAG}_65536 > $OUT/pmc_65536.txt 2>&1; tail -20 $OUT/pmc_65536.txt
bash scripts/pmc.sh ${TAG}_32768 --n-env 32768 > $OUT/pmc_32768.txt 2>&1; tail -20 $OUT/pmc_32768.txt
{
for rep in 1 2; do
  for sched in "-1,-1" "0,0" "1,1" "1,0"; do
    echo -n "rep $rep: "; python scripts/time_rollout.py --lanes 8 --sched=$sched --iters 2000 2>&1 | grep "us/step  "
  done
done
} 2>&1 | tee $OUT/lanes8_sched.txt
