#!/bin/bash
OUT=gpurun_out/${1:-r2h}
mkdir -p $OUT
export TMPDIR=/tmp
T="python scripts/time_rollout.py"
{
for m in 0 1 0 1; do
  export EB_MULTI_TILE=$m
  echo "== multi=$m"
  $T --iters 500 2>&1 | tail -1
  $T --iters 500 --n-veh 64 --f16 2>&1 | tail -1
  $T --iters 500 --n-veh 64 2>&1 | tail -1
  $T --iters 200 --n-env 131072 2>&1 | tail -1
  $T --iters 200 --n-env 262144 2>&1 | tail -1
  $T --iters 100 --n-env 524288 2>&1 | tail -1
done
} > $OUT/multi.txt 2>&1
cat $OUT/multi.txt
