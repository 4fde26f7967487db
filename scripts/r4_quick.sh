#!/bin/bash
# round-4 quick pass on a GPU box: env-step parity subset, phase trace (wave 2 / wave 0 at 4 096 envs), bench.py --env-step summary
# usage: scripts/r4_quick.sh <tag> [notest]
tag=${1:-r4x}
mkdir -p gpurun_out
if [ "$2" != "notest" ]; then
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reset_frames.py tests/test_gpu_golden.py -m gpu -x -q -k "env_step or auto_reset or get_obs or g6 or g7 or reset_pool or masked or facade" 2>&1 | tail -3
fi
timeout 120 python scripts/trace_env_step.py --n-env 4096 > gpurun_out/${tag}_trace_4096.txt 2>&1
timeout 120 python scripts/trace_env_step.py --n-env 65536 > gpurun_out/${tag}_trace_65536.txt 2>&1
grep -A13 "^wave 0" gpurun_out/${tag}_trace_4096.txt | cut -c1-135
grep -A13 "^wave 4" gpurun_out/${tag}_trace_4096.txt | cut -c1-135
grep -A11 "^wave 2" gpurun_out/${tag}_trace_65536.txt | cut -c1-135
timeout 300 python bench.py --env-step > gpurun_out/${tag}_bench_env_step.json 2>/dev/null
python - <<PY
import json
for l in open("gpurun_out/${tag}_bench_env_step.json"):
    d = json.loads(l)
    a = d["step_with_auto_reset"]
    print("n_env %6d: step %.2f us (frac %.3f)  masked reset %.2f us  step+auto-reset %.2f us (finished/step %.4f)" % (d["n_env_per_gpu"], d["avg_launch_us"], d["frac"], d["masked_reset"]["us_per_call"], a["us_per_step"], a["finished_per_step_fraction"]))
PY
