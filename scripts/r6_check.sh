#!/bin/bash
# Round 6: the GPU suite on the product build, then the per-step rollout kernel's launch schedules (eb_debug_set_rollout_sched) at the
# sizes the policy switches between.   usage: bash scripts/r6_check.sh <tag> [notest]
TAG=${1:-r6chk}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
if [ "$2" != notest ]; then
  python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/gpu_suite.txt
fi
{
for rep in 1 2; do
  for cfg in "4096 16" "16384 32" "32768 32" "49152 32" "65536 32" "98304 32" "131072 32" "262144 32" "65536 64 --f16" "32768 64 --f16"; do
    set -- $cfg
    for sched in "-1,-1" "0,0" "1,1" "0,1" "1,0"; do
      echo -n "rep $rep: "; python scripts/time_rollout.py --n-env $1 --n-veh $2 $3 --sched=$sched --iters 2000 2>&1 | grep "us/step  "
    done
  done
done
} 2>&1 | tee $OUT/sched.txt
