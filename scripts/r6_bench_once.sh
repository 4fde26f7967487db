#!/bin/bash
# the driver's command once, with its wall time and the new `extra` entry printed
TAG=${1:-r6g}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
t0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench.err; echo "rc=$?"
t1=$(date +%s.%N); python -c "print(\"bench wall: %.1f s\" % ($t1 - $t0))"
tail -3 $OUT/bench.err
python - <<PY
import json
d = json.load(open('$OUT/bench_steps20.json'))
print('value %.3f G  ms_per_step %.5f  frac %.4f' % (d['value'] / 1e9, d['ms_per_step'], d['roofline']['frac']))
for e in d['extra']:
    for o in e.get('other_tasks_and_native_shapes', []):
        print('  ', o['task'], o['n_veh'], round(o['ms_per_step'] * 1e3, 2), 'us', round(o['frac'], 3), o['launch_form'])
PY
