#!/usr/bin/env python3
"""Average the rocprofv3 counter_collection CSVs of scripts/pmc.sh per kernel (profiling aid)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'rollout_' not in k: continue
        acc['rollout'][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print('  %-24s n=%-4d mean=%.4g' % (c, len(v), sum(v) / len(v)))
