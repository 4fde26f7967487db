#!/bin/bash
# the driver's command on a fresh box (box-to-box spread of the line)
TAG=${1:-r6box}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench.err; echo "rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench_steps20.json'))
print('value %.3f G  ms_per_step %.5f  launch %.3f us  frac %.4f  proj8 %.3f (shard %.2f us, one gpu %.2f us)' % (d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['strong']['projection']['projected_speedup_at_8'], d['strong']['projection']['by_n_gpus']['8']['ms_per_step']*1e3, d['strong']['projection']['one_gpu_ms_per_step']*1e3))
for e in d['extra']:
    if 'workload' in e and 'env_step' in e['workload'][:9]: print('   ', e['workload'][:40], round(e['avg_launch_us'],2), round(e['frac'],3))
"
