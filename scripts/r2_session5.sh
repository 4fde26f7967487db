#!/bin/bash
OUT=gpurun_out/${1:-r2f}
mkdir -p $OUT
export TMPDIR=/tmp
T="python scripts/time_rollout.py"
{
for ab in 0 512; do
  export EB_ABLATE=$ab
  $T --iters 500 2>&1 | tail -1
  $T --iters 500 --n-veh 64 --f16 2>&1 | tail -1
  $T --iters 500 --n-veh 64 2>&1 | tail -1
  $T --iters 400 --lanes 8 2>&1 | tail -1
done
} > $OUT/defer.txt 2>&1
cat $OUT/defer.txt
