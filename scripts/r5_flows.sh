#!/bin/bash
# Round-5: the env step over the flow source — bench entries (step, step + auto reset in one launch) and the per-wave timeline.
TAG=${1:-r5fl}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py --env-step > $OUT/env_step.jsonl 2> $OUT/err.log; python - <<PY
import json
for l in open('$OUT/env_step.jsonl'):
    d = json.loads(l)
    a = d['step_with_auto_reset']
    print('%s...: step %.2f us (frac %.3f)  step+auto reset %.2f us (frac %.3f)' % (d['workload'][:40], d['avg_launch_us'], d['frac'], a['us_per_step'], a['frac']))
PY
timeout 300 python scripts/trace_env_step.py --flows > $OUT/trace_flows.txt 2>&1; tail -60 $OUT/trace_flows.txt
