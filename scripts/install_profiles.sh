#!/bin/bash
# Copy the summaries of one scripts/r3_measure.sh / r4_measure.sh pass from gpurun_out/<tag>/ (scratch) into profiles/ (tracked) as <tag>_*.
# Usage: bash scripts/install_profiles.sh <tag>
set -e
T=$1; G=gpurun_out/$T
cp $G/bench.json profiles/${T}_bench.json
[ -f $G/bench_steps20.json ] && cp $G/bench_steps20.json profiles/${T}_bench_steps20.json
cp $G/env_step.json profiles/${T}_bench_env_step.json
cp $G/bench_shield.json profiles/${T}_bench_shield.json
cp $G/prof_bench.json profiles/${T}_bench_under_rocprof.json
cp $G/prof/p_kernel_stats.csv profiles/${T}_kernel_stats.csv
cp $G/prof_env/p_kernel_stats.csv profiles/${T}_env_step_kernel_stats.csv
cp $G/prof_flows/p_kernel_stats.csv profiles/${T}_flows_kernel_stats.csv
(grep n_env $G/facade_pool.txt; grep n_env $G/facade_flows.txt) > profiles/${T}_facade_env_step_timing.txt
grep n_env $G/reset_pool.txt > profiles/${T}_reset_pool_timing.txt
R=${T:0:2}                                        # r3 / r4: the round's traffic file (bench.py takes the newest whose kernel hash matches)
cp gpurun_out/${T}_pmc/pmc_traffic.json profiles/${R}_pmc_traffic.json
[ -f $G/prof_f16/p_kernel_stats.csv ] && cp $G/prof_f16/p_kernel_stats.csv profiles/${T}_f16x64_kernel_stats.csv
[ -f $G/trace_env_step_4096_auto.txt ] && cp $G/trace_env_step_4096_auto.txt $G/trace_env_step_65536_auto.txt profiles/ && for f in trace_env_step_4096_auto trace_env_step_65536_auto; do mv profiles/$f.txt profiles/${T}_$f.txt; done
[ -f $G/pmc_env_step.log ] && cp $G/pmc_env_step.log profiles/${T}_env_step_pmc.txt
python scripts/pmc_traffic.py gpurun_out/${T}_pmc > profiles/${T}_pmc_traffic.txt
ls profiles | grep "^${T}_"
# round 5 additions (scripts/r5_measure.sh)
[ -f $G/facade_rollout_out.jsonl ] && cp $G/facade_rollout_out.jsonl profiles/${T}_facade_rollout_out.jsonl
for f in trace_rollout_65536 trace_rollout_32768 trace_env_step_flows trace_env_step_flows_auto noreset_profile; do [ -f $G/$f.txt ] && cp $G/$f.txt profiles/${T}_$f.txt; done
[ -f $G/pmc_flows.log ] && cp $G/pmc_flows.log profiles/${T}_flows_pmc.txt
if ls $G/fuzz_*.txt > /dev/null 2>&1; then (for f in $G/fuzz_*.txt; do echo "== $(basename $f)"; grep -v amdgpu.ids $f | tail -3; done) > profiles/${T}_fuzz.txt; fi
[ -f $G/pytest_gpu.log ] && tail -1 $G/pytest_gpu.log > profiles/${T}_gpu_suite.txt
ls profiles | grep "^${T}_" | wc -l
