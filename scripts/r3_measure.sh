#!/bin/bash
# Round-3 measurement pass on a GPU box (through gpurun): bench.py defaults, rocprofv3 kernel stats of the headline and of
# the env step, the shield line, the flows-traffic env step, HBM counters.  Usage: bash scripts/r3_measure.sh <tag>
TAG=${1:-r3m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-side > $OUT/prof_bench.json 2> $OUT/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_env -o p -- python bench.py --env-step > $OUT/env_step.json 2> $OUT/prof_env.err
python bench.py --shield > $OUT/bench_shield.json 2>> $OUT/bench.err
python scripts/time_env_step.py --sizes 4096,65536 --traffic pool > $OUT/facade_pool.txt 2>&1
python scripts/time_env_step.py --sizes 65536 --traffic flows > $OUT/facade_flows.txt 2>&1
python scripts/time_reset_pool.py --sizes 4096,16384,65536 > $OUT/reset_pool.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_flows -o p -- python scripts/time_env_step.py --sizes 65536 --traffic flows > /dev/null 2> $OUT/prof_flows.err
bash scripts/pmc_traffic.sh ${TAG}_pmc > $OUT/pmc_traffic.log 2>&1
tail -n 5 $OUT/facade_pool.txt $OUT/facade_flows.txt; head -8 $OUT/prof_flows/*kernel_stats.csv
