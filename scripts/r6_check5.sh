#!/bin/bash
# Round 6: the shield's accumulation folded into the rollout step's launch — GPU suite, the shield line, the headline / shard kernel loops
TAG=${1:-r6chk5}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/gpu_suite.txt
for i in 1 2 3; do python bench.py --shield 2>/dev/null | tail -1 > $OUT/shield$i.json; python -c "
import json; d=json.load(open('$OUT/shield$i.json')); print('shield: %.2f M checks/s, %.1f us per pass' % (d['value']/1e6, d['ms_per_step']*1e3)); print({k: v for k, v in d.items() if 'us' in k or 'policy' in k})" 2>&1 | cut -c1-400; done
for n in 65536 32768; do for i in 1 2; do python scripts/time_rollout.py --n-env $n --iters 3000 2>&1 | grep "us/step  "; done; done
