#!/bin/bash
# Round 6: the env step's issue priority by phase A/B, then bench.py (long and the driver's short form) on the product build
TAG=${1:-r6chk3}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
python scripts/r6_ab_envprio.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_envprio.txt
python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 2> $OUT/bench20.err | tail -1 > $OUT/bench_steps20.json; echo "bench20 rc=$?"
EB_TAG=$TAG python - <<'PY'
import json, os
for f in ('bench.json', 'bench_steps20.json'):
    d = json.load(open('gpurun_out/%s/%s' % (os.environ['EB_TAG'], f)))
    print(f, 'value %.3f G  ms_per_step %.5f  frac %.4f  launch_us %.3f' % (d['value'] / 1e9, d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us']))
    st = d.get('strong') or {}
    pr = st.get('projection') or {}
    if pr:
        print('  projection at 8: %.3f  one-gpu %.2f us  by n: %s' % (pr['projected_speedup_at_8'], pr['one_gpu_ms_per_step'] * 1e3, {k: round(v['ms_per_step'] * 1e3, 2) for k, v in pr['by_n_gpus'].items()}))
    for e in d.get('extra') or []:
        if 'workload' in e: print('  ', e['workload'][:70], '| us', round(e.get('ms_per_step', 0) * 1e3, 2) if 'ms_per_step' in e else e.get('avg_launch_us'), '| frac', e.get('frac'))
PY
