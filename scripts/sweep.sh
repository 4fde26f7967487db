#!/bin/bash
# Profiling aid: rollout-kernel timings per tile shape (EB_ROLLOUT = 0: 4 x 8 records per lane, 1: 4 x 4, 2: 1 x 4).
run() { echo -n "[$*] "; env "$@" python scripts/time_rollout.py --iters 1000 | tail -1; }
for v in 0 1 2; do run EB_ROLLOUT=$v; done
