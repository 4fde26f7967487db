#!/bin/bash
# Round-2 GPU session 3: where the step's time goes — ablations (EB_ABLATE bits: 1 no closest point, 4 no ego step,
# 8 no near test, 16 no queue pass, 256 no prediction arithmetic) at the headline size and at configs[4].
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
export TMPDIR=/tmp
T="python scripts/time_rollout.py"
{
for ab in 0 256 264 280 285 8 16 5; do
  export EB_ABLATE=$ab
  $T --iters 500 2>&1 | tail -1
  $T --iters 500 --n-veh 64 --f16 2>&1 | tail -1
  $T --iters 100 --n-env 524288 2>&1 | tail -1
done
} > $OUT/ablate.txt 2>&1
cat $OUT/ablate.txt
