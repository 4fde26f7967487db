#!/bin/bash
# Round 6: which tile shape at the batch sizes between the small-tile and the 2048-record-tile regimes, now that the latter has rolling loads
TAG=${1:-r6tile}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
{
for rep in 1 2; do
  for cfg in "4096 16" "8192 32" "12288 32" "16384 32" "24576 32" "32768 32" "16384 64" "32768 16"; do
    set -- $cfg
    for tile in -1 0 1 2; do
      echo -n "rep $rep: "; python scripts/time_rollout.py --n-env $1 --n-veh $2 --tile $tile --iters 3000 2>&1 | grep "us/step  "
    done
  done
done
} 2>&1 | tee $OUT/tiles.txt
