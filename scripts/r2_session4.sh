#!/bin/bash
OUT=gpurun_out/${1:-r2e}
mkdir -p $OUT
export TMPDIR=/tmp
T="python scripts/time_rollout.py"
{
for i in 1 2; do
  $T --iters 500 2>&1 | tail -1
  $T --iters 500 --n-veh 64 --f16 2>&1 | tail -1
  $T --iters 500 --n-veh 64 2>&1 | tail -1
  $T --iters 400 --lanes 8 2>&1 | tail -1
  $T --iters 100 --n-env 524288 2>&1 | tail -1
  $T --iters 1000 --n-env 4096 --n-veh 16 2>&1 | tail -1
done
} > $OUT/timing.txt 2>&1
cat $OUT/timing.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
