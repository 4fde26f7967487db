#!/bin/bash
# same-box A/B of two builds of the library: scripts/ab/libA_*.so against the in-tree one, interleaved
OUT=gpurun_out/${1:-r2ab}
mkdir -p $OUT
export TMPDIR=/tmp
T="python scripts/time_rollout.py"
A=$(ls scripts/ab/libA_*.so | head -1)
{
for i in 1 2 3; do
  for lib in "$A" ""; do
    L=""; [ -n "$lib" ] && L="--lib $lib"
    echo "== ${lib:-in-tree}"
    $T --iters 500 $L 2>&1 | tail -1
    $T --iters 500 --n-veh 64 --f16 $L 2>&1 | tail -1
    $T --iters 500 --n-veh 64 $L 2>&1 | tail -1
    $T --iters 400 --lanes 8 $L 2>&1 | tail -1
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
