#!/bin/bash
# Round 6: slot counts that leave the 2048-record tile half empty (N <= 16: a tile is capped at 64 envs) — the 1024-record tile instead?
TAG=${1:-r6tile2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
{
for rep in 1 2; do
  for cfg in "32768 16" "65536 16" "131072 16" "65536 8" "131072 8" "65536 9" "65536 5" "65536 24"; do
    set -- $cfg
    for tile in 0 1; do
      echo -n "rep $rep: "; python scripts/time_rollout.py --n-env $1 --n-veh $2 --tile $tile --iters 3000 2>&1 | grep "us/step  "
    done
  done
done
} 2>&1 | tee $OUT/tiles.txt
