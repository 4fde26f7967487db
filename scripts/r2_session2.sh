#!/bin/bash
# Round-2 GPU session 2: nt record loads and tile shape at HBM-resident sizes.
OUT=gpurun_out/${1:-r2c}
mkdir -p $OUT
export TMPDIR=/tmp
T="python scripts/time_rollout.py"
{
for ab in 0 128; do
  for tile in -1 1; do
    export EB_ABLATE=$ab EB_ROLLOUT=$tile
    echo "== ablate=$ab tile=$tile"
    $T --iters 500 2>&1 | tail -1
    $T --iters 400 --lanes 8 2>&1 | tail -1
    $T --iters 200 --n-env 262144 2>&1 | tail -1
    $T --iters 100 --n-env 524288 2>&1 | tail -1
    $T --iters 500 --n-veh 64 --f16 2>&1 | tail -1
    $T --iters 400 --n-veh 64 --f16 --lanes 8 2>&1 | tail -1
  done
done
} > $OUT/nt_tile.txt 2>&1
cat $OUT/nt_tile.txt
