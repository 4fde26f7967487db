#!/bin/bash
# Same-box A/B of the closest-point range scan: first groups of table entries in one round trip (round 5) against one group per
# loop trip (rounds 1-4), through eb_debug_set_scan_prefetch — rollout kernel at 4 096 x 16, 32 768 x 32, 65 536 x 32; env step.
TAG=${1:-r5scan}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
{
for rep in 1 2 3; do
  for cfg in "--n-env 4096 --n-veh 16" "--n-env 16384 --n-veh 32" "--n-env 32768 --n-veh 32" "--n-env 65536 --n-veh 32"; do
    for sp in 0 1; do echo -n "scan_prefetch=$sp $cfg: "; python scripts/time_rollout.py $cfg --scan-prefetch $sp --iters 600 2>/dev/null | tail -1; done
  done
done
} | tee $OUT/ab_scan.txt
