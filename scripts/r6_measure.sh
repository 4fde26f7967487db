#!/bin/bash
# Round-6 measurement pass on a GPU box (through gpurun): the GPU suite, bench.py (default, the driver's short form), rocprofv3 kernel stats of the
# headline, of the env step (incl. the flow-source entries) and of the fp16 x 64 instantiation, the shield line, the facade (rollout_out;
# env-step loops), phase timelines (rollout at both sizes, env step pool / flows), the no-resets profile, HBM and instruction counters, fuzz sweeps.
# Usage: bash scripts/r6_measure.sh <tag>   -> gpurun_out/<tag>/ ; scripts/install_profiles.sh <tag> copies the summaries to profiles/
TAG=${1:-r6m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2>> $OUT/bench.err      # the driver's short form
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-side > $OUT/prof_bench.json 2> $OUT/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_env -o p -- python bench.py --env-step > $OUT/env_step.json 2> $OUT/prof_env.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f16 -o p -- python scripts/time_rollout.py --n-veh 64 --f16 --iters 400 > $OUT/f16.txt 2> $OUT/prof_f16.err
python bench.py --shield > $OUT/bench_shield.json 2>> $OUT/bench.err
python bench.py --facade > $OUT/facade_rollout_out.jsonl 2>> $OUT/bench.err
python scripts/time_env_step.py --sizes 4096,65536 --traffic pool --steps 2000 > $OUT/facade_pool.txt 2>&1
python scripts/time_env_step.py --sizes 65536 --traffic flows --steps 50 > $OUT/facade_flows.txt 2>&1
python scripts/time_reset_pool.py --sizes 4096,16384,65536 > $OUT/reset_pool.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_flows -o p -- python scripts/time_env_step.py --sizes 65536 --traffic flows --steps 50 > /dev/null 2> $OUT/prof_flows.err
python scripts/trace_rollout.py --n-env 65536 > $OUT/trace_rollout_65536.txt 2>&1
python scripts/trace_rollout.py --n-env 32768 > $OUT/trace_rollout_32768.txt 2>&1
python scripts/trace_env_step.py --n-env 4096 --auto > $OUT/trace_env_step_4096_auto.txt 2>&1
python scripts/trace_env_step.py --n-env 65536 --auto > $OUT/trace_env_step_65536_auto.txt 2>&1
python scripts/trace_env_step.py --flows > $OUT/trace_env_step_flows.txt 2>&1
python scripts/trace_env_step.py --flows --auto > $OUT/trace_env_step_flows_auto.txt 2>&1
(echo "# no resets"; python scripts/offgrid_profile.py 65536; echo "# auto reset in the step launch"; python scripts/offgrid_profile.py 65536 auto) 2>&1 | grep -v amdgpu.ids > $OUT/noreset_profile.txt
bash scripts/pmc_env_step.sh ${TAG}_pmc_env > $OUT/pmc_env_step.log 2>&1
bash scripts/pmc_flows.sh ${TAG}_pmc_flows > $OUT/pmc_flows.log 2>&1
bash scripts/pmc_traffic.sh ${TAG}_pmc > $OUT/pmc_traffic.log 2>&1
timeout 900 python scripts/fuzz_env_auto.py > $OUT/fuzz_env_auto.txt 2>&1; tail -3 $OUT/fuzz_env_auto.txt
timeout 600 python scripts/fuzz_env_auto.py --waves 4 > $OUT/fuzz_env_auto_w4.txt 2>&1; tail -2 $OUT/fuzz_env_auto_w4.txt
timeout 400 python scripts/fuzz_rollout.py > $OUT/fuzz_rollout.txt 2>&1; tail -2 $OUT/fuzz_rollout.txt
timeout 400 python scripts/fuzz_env_step.py > $OUT/fuzz_env_step.txt 2>&1; tail -2 $OUT/fuzz_env_step.txt
tail -n 8 $OUT/facade_pool.txt $OUT/facade_flows.txt; head -5 $OUT/prof_env/*kernel_stats.csv | cut -c1-150
