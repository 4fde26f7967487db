#!/bin/bash
# Profiling aid: rollout-kernel timings per tile shape (--tile 0: 4 x 8 records per lane, 1: 4 x 4, 2: 1 x 4; eb_debug_set_tile).
for v in 0 1 2; do echo -n "[tile $v] "; python scripts/time_rollout.py --iters 1000 --tile $v | tail -1; done
