#!/usr/bin/env python3
"""Print the figures DESIGN.md §6 quotes from one measurement pass: python scripts/summarize_pass.py <tag> (reads profiles/<tag>_*)."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = sys.argv[1]
P = lambda name: os.path.join(ROOT, 'profiles', '%s_%s' % (T, name))
d = json.load(open(P('bench.json')))
r = d['roofline']
print('headline: %.3f G env-steps/s, %.2f us per step wall (%.2f-%.2f), launch %.2f us = %.1f %%, traffic %s' % (
    d['value'] / 1e9, d['ms_per_step'] * 1e3, d['repeats']['ms_per_step_min'] * 1e3, d['repeats']['ms_per_step_max'] * 1e3,
    r['avg_launch_us'], 100 * r['frac'], None if r['traffic'] is None else '%.2f MB = %.3f x' % (r['traffic'] / 1e6, r['traffic'] / r['alg_bytes_per_launch'])))
for k, v in r['hbm_resident'].items():
    if isinstance(v, dict) and 'frac' in v:
        print('  hbm_resident %-30s %.1f %%  (%.2f us per step)' % (k, 100 * v['frac'], v['ms_per_step'] * 1e3))
s = d.get('strong') or {}
if s:
    print('strong N=1: %.2f G, %.1f %%, %.2f us per step' % (s['value'] / 1e9, 100 * s['frac'], s['ms_per_step'] * 1e3))
    for n, v in (s.get('projection') or {}).get('by_n_gpus', {}).items():
        print('  projection N=%s: shard %d envs %.2f us per step -> %.2f x' % (n, v['n_env_per_gpu'], v['ms_per_step'] * 1e3, v['projected_speedup']))
for e in d.get('extra', []):
    print('%-62s %.2f us = %.1f %%, %.3g env-steps/s, traffic %s' % (e.get('workload', '')[:62], e.get('avg_launch_us', 0), 100 * e.get('frac', 0), e.get('value', 0),
          None if e.get('traffic') is None else '%.2f MB' % (e['traffic'] / 1e6)))
    for o in e.get('other_tasks_and_native_shapes', []):
        print('    %-9s N_veh=%-3d %.2f us per step = %.1f %%, %.3g env-steps/s' % (o['task'], o['n_veh'], o['ms_per_step'] * 1e3, 100 * o['frac'], o['value']))
    if 'one_launch_forms' in e:
        for k, v in e['one_launch_forms'].items():
            if isinstance(v, dict):
                print('    %-46s %.2f us per step (min %.2f), %.3g env-steps/s' % (k, v['us_per_step'], v['us_per_step_min'], v['value']))
    if 'step_with_auto_reset' in e:
        a = e['step_with_auto_reset']
        print('    step + auto reset %.2f us (frac %.3f%s)%s' % (a['us_per_step'], a['frac'],
              ', finished per step %.4f' % a['finished_per_step_fraction'] if 'finished_per_step_fraction' in a else '',
              '; masked reset %.2f us' % e['masked_reset']['us_per_call'] if 'masked_reset' in e else ''))
c = d.get('cpu_baseline') or {}
if c:
    print('cpu baseline: %.2f M on %d threads, %.3f M on one' % (c['value'] / 1e6, c['cores'], (c.get('value_1core') or 0) / 1e6))
if os.path.exists(P('bench_steps20.json')):
    k = json.load(open(P('bench_steps20.json')))
    print('--steps 20 --warmup 5: %.2f us per step, launch %.2f us = %.1f %% (regions %.2f-%.2f), %.3f G' % (k['ms_per_step'] * 1e3, k['roofline']['avg_launch_us'], 100 * k['roofline']['frac'],
          k['roofline']['avg_launch_us_by_region']['min'], k['roofline']['avg_launch_us_by_region']['max'], k['value'] / 1e9))
if os.path.exists(P('bench_shield.json')):
    k = json.load(open(P('bench_shield.json')))
    print('shield: %.1f M checks/s, policy kernel %.1f us = %.1f %%' % (k['value'] / 1e6, k['roofline'].get('avg_launch_us', 0), 100 * k['roofline']['frac']))
for name in ('kernel_stats.csv', 'env_step_kernel_stats.csv', 'f16x64_kernel_stats.csv', 'flows_kernel_stats.csv'):
    if os.path.exists(P(name)):
        print(name)
        for row in list(csv.DictReader(open(P(name))))[:5]:
            if 'eb' in row['Name']:
                print('    %-70s %6s calls  %.2f us' % (row['Name'].split('(')[0][:70], row['Calls'], float(row['AverageNs']) / 1e3))
for name in ('facade_env_step_timing.txt', 'reset_pool_timing.txt'):
    if os.path.exists(P(name)):
        print(name); print('    ' + open(P(name)).read().replace('\n', '\n    ').rstrip())
if os.path.exists(P('pmc_traffic.txt')):
    print('pmc_traffic.txt'); print('    ' + '\n    '.join(l[:200] for l in open(P('pmc_traffic.txt')).read().splitlines()[-5:]))
