#!/bin/bash
# Round-5 quick pass for the env step: its parity tests, bench.py --env-step (step / step + auto reset at 65 536 x 16, 4 096 x 16, flows 65 536 x 60), timelines.
# usage: bash scripts/r5_envq.sh <tag> [full]
TAG=${1:-r5e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
if [ "$2" == "full" ]; then
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
else
timeout 1200 python -m pytest tests -m gpu -x -q -k "env or flow or reset or facade or binding or G6 or G7 or golden" > $OUT/pytest_env.log 2>&1; echo "pytest(env) rc=$?"; tail -3 $OUT/pytest_env.log
fi
for rep in 1 2; do
timeout 900 python bench.py --env-step > $OUT/env_step_$rep.jsonl 2> $OUT/err.log; python - <<PY
import json
for l in open('$OUT/env_step_$rep.jsonl'):
    d = json.loads(l)
    a = d['step_with_auto_reset']
    print('%s...: step %.2f us (frac %.3f)  step+auto reset %.2f us (frac %.3f)' % (d['workload'][:40], d['avg_launch_us'], d['frac'], a['us_per_step'], a['frac']))
PY
done
timeout 300 python scripts/trace_env_step.py --flows > $OUT/trace_flows.txt 2>&1; grep -A4 "^wave\|^launch" $OUT/trace_flows.txt | head -40
timeout 300 python scripts/trace_env_step.py > $OUT/trace_env_step.txt 2>&1; grep -A4 "^wave\|^launch" $OUT/trace_env_step.txt | head -40
