#!/bin/bash
# Round-5 iteration pass: whole GPU suite, then the same-box A/B of the accumulating launches.   usage: bash scripts/r5_pass.sh <tag>
TAG=${1:-r5p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
bash scripts/r5_ab_acc.sh $TAG 2>&1 | grep -v "^build\|^acc tests\|passed\|amdgpu.ids\|^\.\.\."
