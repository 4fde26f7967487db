#!/bin/bash
# Round-5 iteration pass on a GPU box: the accumulating-rollout tests first, then the whole GPU suite, the chain
# micro-benchmark, and bench.py in the driver's short form.   usage: bash scripts/r5_quick.sh <tag>
TAG=${1:-r5q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "accumulating or plan_graph or episode_summary" > $OUT/pytest_acc.log 2>&1; echo "acc tests rc=$?"; tail -5 $OUT/pytest_acc.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 300 ./scripts/micro/chainbench > $OUT/chainbench.txt 2>&1; cat $OUT/chainbench.txt
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
