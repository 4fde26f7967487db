#!/usr/bin/env python3
"""Reduce the rocprofv3 counter CSVs of scripts/pmc_traffic.sh to HBM bytes per rollout launch.

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Calibration: the copy kernel moves exactly
`bytes` in and `bytes` out per launch (scripts/micro/copybench.hip, 65 536 x 137 floats); the ratio
known / counted gives one factor per counter in this access pattern (16 bytes per lane, coalesced) — on
gfx950 the read factor comes out at 2 (MI355X_MICROARCH.md §HBM), the write factor near 1."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from env_build_amd import build as _build

out = sys.argv[1]
COPY_BYTES = 65536 * 137 * 4


HEADLINE_GRID = 1024 * 320   # the 65 536 x 32 launch: 1024 tiles of 64 envs, 5 waves each (bench.py also runs larger batches)


def mean_counter(sub, counter, kernel_substr, grid=None):
    vals = []
    for f in glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r['Kernel_Name'] for k in ([kernel_substr] if isinstance(kernel_substr, str) else kernel_substr)) and r['Counter_Name'] == counter and (grid is None or int(r['Grid_Size']) == grid):
                vals.append(float(r['Counter_Value']))
    if not vals:
        raise SystemExit('no %s rows for %s in %s' % (counter, kernel_substr, sub))
    return sum(vals) / len(vals), len(vals)


res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    k_kib, n_k = mean_counter('bench_' + c, c, 'rollout_fused_4x8<0, true, 8, float>', HEADLINE_GRID)
    c_kib, n_c = mean_counter('copy_' + c, c, 'copy_one')
    factor = COPY_BYTES / (c_kib * 1024.0)
    res[c] = {'rollout_kib_per_launch': k_kib, 'rollout_launches': n_k, 'copy_kib_per_launch': c_kib,
              'copy_launches': n_c, 'copy_known_bytes': COPY_BYTES, 'calibration_factor': factor,
              'rollout_bytes_per_launch': k_kib * 1024.0 * factor}
read_b, write_b = res['FETCH_SIZE']['rollout_bytes_per_launch'], res['WRITE_SIZE']['rollout_bytes_per_launch']
alg = (104 + 32 * 32) * 65536
# the sources that were MEASURED (stamped by pmc_traffic.sh next to the counter files), never the tree this summary happens to run in
try:
    measured_hash = json.load(open(os.path.join(out, 'kernel_hash.json')))
except OSError:
    measured_hash = {k: None for k in _build.KERNEL_SOURCES}
    print('(no kernel_hash.json next to the counter files: the summary carries no source hash and bench.py will not use it)')
summary = {'kernel_source_hash': measured_hash,   # bench.py checks these against the sources it runs
           'hbm_bytes_per_launch': read_b + write_b, 'read_bytes_per_launch': read_b, 'write_bytes_per_launch': write_b,
           'algorithmic_bytes_per_launch': alg, 'traffic_over_algorithmic': (read_b + write_b) / alg,
           'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace; KiB per dispatch x 1024 x '
                     'calibration factor measured on a float4 copy of a known 35.9 MB buffer in the same run '
                     '(gfx950: FETCH_SIZE counts half of a wide coalesced read)',
           'counters': res}
# the one-launch env step (bench.py --env-step: 65 536 envs x 16 candidates, 64-env tiles -> grid 1024 x 256), same calibration
try:
    es = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        k_kib, n_k = mean_counter('envstep_' + c, c, 'env_step_kernel<0, 64, false, false, 4>', 1024 * 256)
        es[c] = {'kib_per_launch': k_kib, 'launches': n_k, 'bytes_per_launch': k_kib * 1024.0 * res[c]['calibration_factor']}
    es_alg = (8 * 41 + 33 * 16 + 105) * 65536
    es_total = es['FETCH_SIZE']['bytes_per_launch'] + es['WRITE_SIZE']['bytes_per_launch']
    summary['env_step'] = {'hbm_bytes_per_launch': es_total, 'read_bytes_per_launch': es['FETCH_SIZE']['bytes_per_launch'],
                           'write_bytes_per_launch': es['WRITE_SIZE']['bytes_per_launch'], 'algorithmic_bytes_per_launch': es_alg,
                           'traffic_over_algorithmic': es_total / es_alg, 'kernel': 'eb::env_step_kernel<0, 64, false, false, 4>', 'counters': es}
except SystemExit as e:
    summary['env_step'] = None
    print('(no env-step passes: %s)' % e)
# configs[4]: 65 536 envs x 64 slots, binary16 state (scripts/time_rollout.py --n-veh 64 --f16: 2048 tiles of 32 envs x 5 waves)
try:
    f16 = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        k_kib, n_k = mean_counter('f16_' + c, c, ('rollout_fused_4x8<0, true, 3, _Float16>', 'rollout_fused_4x8ILi0ELb1ELi3EDF16_', 'rollout_fused_4x8<0, true, 8, _Float16>', 'rollout_fused_4x8ILi0ELb1ELi8EDF16_'), 2048 * 320)
        f16[c] = {'kib_per_launch': k_kib, 'launches': n_k, 'bytes_per_launch': k_kib * 1024.0 * res[c]['calibration_factor']}
    f_alg = (68 + 16 * 64) * 65536
    f_total = f16['FETCH_SIZE']['bytes_per_launch'] + f16['WRITE_SIZE']['bytes_per_launch']
    summary['fp16_x64'] = {'hbm_bytes_per_launch': f_total, 'read_bytes_per_launch': f16['FETCH_SIZE']['bytes_per_launch'],
                           'write_bytes_per_launch': f16['WRITE_SIZE']['bytes_per_launch'], 'algorithmic_bytes_per_launch': f_alg,
                           'traffic_over_algorithmic': f_total / f_alg, 'kernel': 'eb::rollout_fused_4x8<0, true, 3, _Float16>', 'counters': f16}
except SystemExit as e:
    summary['fp16_x64'] = None
    print('(no fp16 passes: %s)' % e)
json.dump(summary, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k not in ('counters', 'env_step', 'fp16_x64', 'kernel_source_hash')}, indent=1))
if summary.get('fp16_x64'):
    print('fp16 x 64:', json.dumps({k: v for k, v in summary['fp16_x64'].items() if k != 'counters'}))
if summary.get('env_step'):
    print('env step:', json.dumps({k: v for k, v in summary['env_step'].items() if k != 'counters'}))
for c, r in res.items():
    print('%s: rollout %.0f KiB x factor %.3f (copy: %.0f KiB for %d known bytes)' % (c, r['rollout_kib_per_launch'], r['calibration_factor'], r['copy_kib_per_launch'], COPY_BYTES))
