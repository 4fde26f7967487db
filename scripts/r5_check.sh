#!/bin/bash
# Round-5 check pass: GPU suite, env-step bench entries, rollout timelines at the headline and the 8-GPU shard size, bench.py in the driver's short form.
# usage: bash scripts/r5_check.sh <tag> [notest]
TAG=${1:-r5h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
if [ "$2" != "notest" ]; then
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
fi
timeout 900 python bench.py --env-step > $OUT/env_step.jsonl 2> $OUT/err.log; python - <<PY
import json
for l in open('$OUT/env_step.jsonl'):
    d = json.loads(l)
    a = d['step_with_auto_reset']
    print('%s...: step %.2f us (frac %.3f)  step+auto reset %.2f us (frac %.3f)' % (d['workload'][:40], d['avg_launch_us'], d['frac'], a['us_per_step'], a['frac']))
PY
timeout 300 python scripts/trace_rollout.py --n-env 65536 > $OUT/trace_rollout_65536.txt 2>&1; head -24 $OUT/trace_rollout_65536.txt
timeout 300 python scripts/trace_rollout.py --n-env 32768 > $OUT/trace_rollout_32768.txt 2>&1; head -24 $OUT/trace_rollout_32768.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; python - <<PY
import json
try:
    d = json.load(open('$OUT/bench_steps20.json'))
    print('steps20: value %.3f G  ms_per_step %.5f  launch_us %.3f frac %.3f' % (d['value'] / 1e9, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))
    p = d['strong']['projection']
    print('projection:', p['one_gpu_ms_per_step'], {k: (v['ms_per_step'], v['projected_speedup']) for k, v in p['by_n_gpus'].items()})
except Exception as e:
    print('bench steps20 failed', e); print(open('$OUT/bench_steps20.err').read()[-2000:])
PY
