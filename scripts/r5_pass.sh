#!/bin/bash
# Round-5 iteration pass: whole GPU suite, then the same-box A/B of the accumulating launches.   usage: bash scripts/r5_pass.sh <tag>
TAG=${1:-r5p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
bash scripts/r5_ab_acc.sh $TAG 2>&1 | grep -v "^build\|^acc tests\|passed\|amdgpu.ids\|^\.\.\."
timeout 600 python bench.py --facade > $OUT/facade.jsonl 2>> $OUT/err.log; python - <<PY
import json
for l in open('$OUT/facade.jsonl'):
    d = json.loads(l)
    print(d['workload'][:60], {k: (round(v['host_us'], 2), round(v['with_drain_us'], 2)) for k, v in d.items() if isinstance(v, dict) and 'host_us' in v}, d.get('facade_over_c_abi'))
PY
bash scripts/r5_ab_scan.sh $TAG 2>&1 | grep -v "^build"
