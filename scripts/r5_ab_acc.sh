#!/bin/bash
# Same-box A/B: accumulating launches + fold against plain launches + the two-pass summary, the driver's short form and a
# long region; the 32 768-env shard likewise.   usage: bash scripts/r5_ab_acc.sh <tag>
TAG=${1:-r5ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "accumulating or plan_graph or episode_summary" > $OUT/pytest_acc.log 2>&1; echo "acc tests rc=$?"; tail -3 $OUT/pytest_acc.log
for rep in 1 2; do
for mode in "--acc-summary" ""; do
  for cfg in "--steps 20 --warmup 5" "--steps 500 --warmup 50" "--n-env 32768 --steps 500 --warmup 50"; do
    python bench.py $cfg $mode --no-side --no-cpu-baseline 2>>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s %-40s ms/step %.5f  launch_us %.3f  frac %.3f  %s' % ('$mode' or 'two-pass (default)', '$cfg', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['workload'][-40:]))
"
  done
done
done | tee $OUT/ab.txt
