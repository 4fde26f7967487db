#!/bin/bash
# Profiling aid: rollout-kernel timings under EB_ABLATE / EB_ROLLOUT settings (see eb_rollout.hip).
run() { echo -n "[$*] "; env "$@" python scripts/time_rollout.py --iters 1000 | tail -1; }
for v in 0 1 3 4 5; do run EB_ROLLOUT=$v; done
