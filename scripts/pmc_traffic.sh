#!/bin/bash
# HBM traffic of the rollout kernel from the L2 memory-side counters, per MI355X_MICROARCH.md §HBM:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (with --kernel-trace only), calibrated on a plain float4
# copy of an obs-sized buffer (known bytes, same 16-byte-per-lane access width), gfx950 FETCH_SIZE x2.
# Usage (GPU box, repo root): bash scripts/pmc_traffic.sh <tag>   -> gpurun_out/<tag>/pmc_traffic.json
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -x scripts/micro/copybench ] || make -s -C scripts/micro copybench
# the hash of the kernel sources these passes measure, stamped HERE, at measurement time (pmc_traffic.py carries it into the JSON;
# bench.py refuses the bytes of other code)
python -c "import json; from env_build_amd import build as b; json.dump({k: b.kernel_hash(k) for k in b.KERNEL_SOURCES}, open('$OUT/kernel_hash.json', 'w'))"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/bench_$c -o p -- python bench.py --steps 50 --warmup 25 --no-cpu-baseline --no-side > $OUT/bench_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/copy_$c -o p -- scripts/micro/copybench calib > $OUT/copy_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/envstep_$c -o p -- python bench.py --env-step > $OUT/envstep_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/f16_$c -o p -- python scripts/time_rollout.py --n-veh 64 --f16 --iters 60 > $OUT/f16_$c.log 2>&1
done
python scripts/pmc_traffic.py $OUT | tee $OUT/pmc_traffic.txt
