#!/bin/bash
# Round 6: the native slot counts (8 / 9 / 5) by tile shape and batch size — is the 256-record tile ever ahead of the 1024-record one?
TAG=${1:-r6tile3}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
{
for rep in 1 2; do
  for cfg in "65536 8 left" "65536 9 straight" "65536 5 right" "16384 8 left" "262144 8 left" "65536 4 left" "65536 2 left"; do
    set -- $cfg
    for tile in 1 2; do
      echo -n "rep $rep: "; python scripts/time_rollout.py --task $3 --n-env $1 --n-veh $2 --tile $tile --iters 3000 2>&1 | grep "us/step  "
    done
  done
done
} 2>&1 | tee $OUT/tiles.txt
