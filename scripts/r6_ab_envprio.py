#!/usr/bin/env python3
"""A/B aid (round 6): the one-launch env step with its issue priority by phase off / on (eb_debug_set_rollout_sched), interleaved —
bench.py's env-step entries at 65 536 x 16, 4 096 x 16 and the flow source's 65 536 x 60."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device('cuda', 0)
for rep in range(3):
    for bp in (0, 1):
        for b in (65536, 4096):
            d = bench.env_step_bench(torch, dev, b, by_progress=bp)
            print('rep %d by_progress=%d  %6d x 16: step %.2f us  step + auto reset %.2f us' % (rep, bp, b, d['avg_launch_us'], d['step_with_auto_reset']['us_per_step']), flush=True)
        d = bench.env_step_flows_bench(torch, dev, 65536, by_progress=bp)
        print('rep %d by_progress=%d  flows 65536 x 60: %s' % (rep, bp, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k in ('avg_launch_us',) or k.startswith('us_')} ), flush=True)
        a = d.get('step_with_auto_reset') or {}
        print('      flows + auto reset: %s' % {k: round(v, 2) for k, v in a.items() if isinstance(v, float)}, flush=True)
