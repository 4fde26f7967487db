#!/bin/bash
# Round-2 GPU session 1: parity, bench under the driver's flags and by default, copy floors, store-policy A/B.
TAG=${1:-r2a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver rc=$?"
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 3000 $OUT/bench.err
make -s -C scripts/micro copybench
scripts/micro/copybench > $OUT/copy_65536.txt 2>&1
scripts/micro/copybench 524288 > $OUT/copy_524288.txt 2>&1
for ab in 0 32 64 96; do
  EB_ABLATE=$ab python scripts/time_rollout.py --iters 500 2>&1 | tail -1
  EB_ABLATE=$ab python scripts/time_rollout.py --iters 400 --lanes 8 2>&1 | tail -1
  EB_ABLATE=$ab python scripts/time_rollout.py --iters 500 --n-veh 64 --f16 2>&1 | tail -1
done > $OUT/store_policy.txt
cat $OUT/store_policy.txt
python scripts/time_rollout.py --iters 500 --n-env 4096 --n-veh 16 2>&1 | tail -1 | tee $OUT/small.txt
python scripts/time_rollout.py --iters 100 --n-env 524288 2>&1 | tail -1 | tee -a $OUT/small.txt
