#!/bin/bash
# Round 6, last pass on the committed kernel sources: HBM counters (profiles/r6_pmc_traffic.json must carry the hash of what is committed),
# then bench.py in both forms (their `traffic` fields filled from it), a soak of the vectorised env loop.   usage: bash scripts/r6_final.sh <tag>
TAG=${1:-r6f}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
bash scripts/pmc_traffic.sh ${TAG}_pmc > $OUT/pmc_traffic.log 2>&1; tail -4 $OUT/pmc_traffic.log
cp gpurun_out/${TAG}_pmc/pmc_traffic.json profiles/r6_pmc_traffic.json     # (on the box: the bench runs below read it; install by hand from gpurun_out/<tag>_pmc afterwards)
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2>> $OUT/bench.err; echo "bench20 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-side > $OUT/prof_bench.json 2> $OUT/prof.err
timeout 600 python scripts/soak_env.py --auto-reset > $OUT/soak_auto.txt 2>&1; tail -3 $OUT/soak_auto.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/gpu_suite.txt
