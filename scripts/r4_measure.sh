#!/bin/bash
# Round-4 measurement pass on a GPU box (through gpurun): bench.py defaults, rocprofv3 kernel stats of the headline, of the env step
# (incl. the auto-reset variant) and of the fp16 x 64 instantiation, the shield line, facade timings (pool: auto reset / masked reset /
# none; flows: in the step launch / as a launch of its own), the env step's instruction counters, HBM counters.
# Usage: bash scripts/r4_measure.sh <tag>   -> gpurun_out/<tag>/ ; scripts/install_profiles.sh <tag> copies the summaries to profiles/
TAG=${1:-r4m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2>> $OUT/bench.err      # the driver's short form
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-side > $OUT/prof_bench.json 2> $OUT/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_env -o p -- python bench.py --env-step > $OUT/env_step.json 2> $OUT/prof_env.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f16 -o p -- python scripts/time_rollout.py --n-veh 64 --f16 --iters 400 > $OUT/f16.txt 2> $OUT/prof_f16.err
python bench.py --shield > $OUT/bench_shield.json 2>> $OUT/bench.err
python scripts/time_env_step.py --sizes 4096,65536 --traffic pool --steps 2000 > $OUT/facade_pool.txt 2>&1
python scripts/time_env_step.py --sizes 65536 --traffic flows --steps 50 > $OUT/facade_flows.txt 2>&1
python scripts/time_env_step.py --sizes 65536 --traffic flows --steps 50 --separate-flow 2>&1 | sed 's/^n_env/flow step as a launch of its own: n_env/' >> $OUT/facade_flows.txt
python scripts/time_reset_pool.py --sizes 4096,16384,65536 > $OUT/reset_pool.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_flows -o p -- python scripts/time_env_step.py --sizes 65536 --traffic flows --steps 50 > /dev/null 2> $OUT/prof_flows.err
python scripts/trace_env_step.py --n-env 4096 --auto > $OUT/trace_env_step_4096_auto.txt 2>&1
python scripts/trace_env_step.py --n-env 65536 --auto > $OUT/trace_env_step_65536_auto.txt 2>&1
bash scripts/pmc_env_step.sh ${TAG}_pmc_env > $OUT/pmc_env_step.log 2>&1
bash scripts/pmc_traffic.sh ${TAG}_pmc > $OUT/pmc_traffic.log 2>&1
tail -n 8 $OUT/facade_pool.txt $OUT/facade_flows.txt; head -5 $OUT/prof_env/*kernel_stats.csv | cut -c1-150
