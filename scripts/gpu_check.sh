#!/bin/bash
# Standard GPU-box pass: parity tests, smoke, bench (graph + eager), rocprofv3 kernel stats.
# Usage (from the repo root, through gpurun): bash scripts/gpu_check.sh <tag>
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
python bench.py --eager --no-cpu-baseline > $OUT/bench_eager.json 2>> $OUT/bench.err; cat $OUT/bench_eager.json
# headline line only (--no-side), so that the per-kernel average below is the headline kernel's alone
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-side > $OUT/prof_bench.json 2> $OUT/prof.err
cat $OUT/prof_bench.json
ls $OUT/prof | head; find $OUT/prof -name '*stats*' | head
